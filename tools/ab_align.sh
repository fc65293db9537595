#!/bin/bash
# A/B of experiment builds of the library on the headline kernel (same box, same process order):
#   tools/ab_align.sh out.log build/libexp_A.so build/libexp_B.so ...     ("-" = the in-tree library)
# Per library: tools/perf_align_fixed_cost.py at 250 and 2000 rows, score only and with traceback, twice.
set -uo pipefail
out=$1; shift
mkdir -p gpurun_out
: > "$out"
for rep in 1 2; do
for L in "$@"; do
    if [ "$L" = "-" ]; then unset SARLACC_LIB_PATH; else export SARLACC_LIB_PATH=$PWD/$L; fi
    echo "== $L (run $rep)" >> "$out"
    timeout -k 10 200 python tools/perf_align_fixed_cost.py 1000000 250,2000 2>&1 | grep "^L=\|^trace" >> "$out" || exit 1
done
done
