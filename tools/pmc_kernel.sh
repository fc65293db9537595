# SQ / LDS / cache counters of the kernels whose name contains <substr>, for an arbitrary python command (three passes).
#   tools/pmc_kernel.sh <tag> <kernel substr> <script> [args]
set -euo pipefail
TAG=$1; SUB=$2; SCRIPT=$3; shift; shift; shift
pass() {   # pass <dir> <counters...> -- : one rocprofv3 counter pass of the script; a failed pass (a counter this arch does not have) is an error, not an empty table
    local d=$1; shift
    if ! rocprofv3 --output-format csv --pmc "$@" -d "$OUT/$d" -o pmc -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "${ARGS[@]}" > "$OUT/$d.log" 2>&1; then
        echo "pass $d FAILED:"; tail -n 15 "$OUT/$d.log"; exit 1
    fi
    echo "pass $d done"
}
ARGS=("$@")
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp
pass p1 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass p2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE
pass p3 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
for p in ("p1", "p2", "p3"):
    tot, calls = {}, {}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "$SUB" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
                tot[key] = tot.get(key, 0) + float(r["Counter_Value"])
                calls[key] = calls.get(key, 0) + 1
    for k in sorted(tot):
        print("%-42s %-24s %14.5g  (%d launches)" % (k[0], k[1], tot[k], calls[k]))
PY
find $OUT -name "*counter_collection.csv" -size +20M -delete
