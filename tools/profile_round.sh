#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, then rocprofv3 kernel stats and separate
# PMC passes (FETCH_SIZE, WRITE_SIZE, two SQ/GRBM sets -- MI355X_MICROARCH.md "rocprofv3 PMC
# slots") of the SAME bench command (DP steps + the 10^6-read pipeline), all into gpurun_out/<tag>/.
# tools/pmc_summary.py turns the counter CSVs into one JSON per dominant kernel
# (k_align, k_msa_pairwise, k_consensus_code, k_m2_group, k_align_wide) -- the files bench.py reads from profiles/.
#   usage: tools/profile_round.sh <tag> [extra bench.py args]
set -euo pipefail
TAG=${1:-prof}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# which machine this is: timings and counters of different boxes of the pool differ by 7-9 %, so every summary carries it
python3 - "$OUT/box.json" <<'PY'
import json, os, socket, subprocess, sys
def run(cmd):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout
    except Exception as e:   # noqa: BLE001
        return "unavailable: %s" % e
info = {"hostname": socket.gethostname(), "cpus": len(os.sched_getaffinity(0))}
smi = run(["rocm-smi", "--showuniqueid", "--showserial", "--showproductname", "--showclocks", "--json"])
try:
    info["rocm_smi"] = json.loads(smi)
except ValueError:
    info["rocm_smi"] = smi[-2000:]
rinfo = run(["rocminfo"])
info["gpu_agents"] = [l.strip() for l in rinfo.splitlines() if "Marketing Name" in l or "Uuid" in l or "Max Clock Freq" in l][:24]
json.dump(info, open(sys.argv[1], "w"), indent=1)
print("box:", info["hostname"], [a for a in info["gpu_agents"] if "Uuid" in a and "GPU" in a][:1])
PY
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu --no-host-pointer $*"
python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"; tail -c 1500 "$OUT/bench.json"
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
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_sq2" -o pmc -- $BENCH > "$OUT/pmc_sq2.log" 2>&1 || echo "sq2 pass failed (counter names?)"
echo "sq2 done"
cd - > /dev/null
for K in k_align k_msa_pairwise k_consensus_code k_m2_group k_align_wide; do
    python3 tools/pmc_summary.py "$OUT" $K > "$OUT/pmc_$K.json"
done
cat "$OUT"/pmc_k_*.json
# keep the merge-back small: drop everything but the stats and counter CSVs
find "$OUT" -type f ! -name '*.csv' ! -name '*.json' ! -name '*.log' ! -name '*.err' -delete
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
find "$OUT" -name '*counter_collection.csv' -size +30M -delete
