#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats and the three PMC
# passes (FETCH_SIZE, WRITE_SIZE, SQ/GRBM in separate runs, MI355X_MICROARCH.md "rocprofv3
# PMC slots"), all into gpurun_out/<tag>/.  tools/pmc_summary.py turns the counter CSVs
# into the JSON bench.py reads from profiles/.
#   usage: tools/profile_round.sh <tag> [extra bench.py args]
set -e
TAG=${1:-prof}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu --no-pipeline $*"
python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"; tail -c 600 "$OUT/bench.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats done"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "fetch done"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "write done"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_sq" -o pmc -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "sq done"
cd - > /dev/null
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
cat "$OUT/pmc_summary.json"
# keep the merge-back small: drop everything but the stats and counter CSVs
find "$OUT" -type f ! -name '*.csv' ! -name '*.json' ! -name '*.log' ! -name '*.err' -delete
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
