#!/bin/bash
# A/B of experiment builds of the library on the two merge workloads (pure groups / pipeline clusters):
#   tools/ab_libs.sh out.log build/libexp_A.so build/libexp_B.so ...     ("-" = the in-tree library)
# One line per library and workload: stage times and the merge kernel's phase cycles.
set -uo pipefail
out=$1; shift
mkdir -p gpurun_out
: > "$out"
for L in "$@"; do
    if [ "$L" = "-" ]; then unset SARLACC_LIB_PATH; else export SARLACC_LIB_PATH=$PWD/$L; fi
    echo "== $L" >> "$out"
    timeout -k 10 200 python tools/perf_pipeline_resident.py ${AB_MOLECULES:-100000} 2 2 pure 2>&1 | grep -A1 'rep 1' | cut -c1-900 >> "$out" || exit 1
    timeout -k 10 200 python tools/perf_pipeline_resident.py ${AB_MOLECULES:-100000} 2 2 2>&1 | grep -A3 '^rep 1' | grep -v 'stage s\|msa host' | cut -c1-1200 >> "$out" || exit 1
done
