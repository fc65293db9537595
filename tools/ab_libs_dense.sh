#!/bin/bash
# like ab_libs.sh, on the dense clusters of the giant pre-group rehearsal (50 000 molecules, 10-base UMIs) and the pipeline clusters:
#   tools/ab_libs_dense.sh out.log build/libexp_A.so ...     ("-" = the in-tree library)
set -uo pipefail
out=$1; shift
mkdir -p gpurun_out
: > "$out"
for L in "$@"; do
    if [ "$L" = "-" ]; then unset SARLACC_LIB_PATH; else export SARLACC_LIB_PATH=$PWD/$L; fi
    echo "== $L" >> "$out"
    timeout -k 10 300 python tools/perf_pipeline_resident.py 50000 2 2 clusters 10 2>&1 | grep -A3 '^rep 1' | grep -v 'stage s\|msa host' | cut -c1-1200 >> "$out" || exit 1
    timeout -k 10 200 python tools/perf_pipeline_resident.py ${AB_MOLECULES:-100000} 2 2 2>&1 | grep -A3 '^rep 1' | grep -v 'stage s\|msa host' | cut -c1-1200 >> "$out" || exit 1
done
