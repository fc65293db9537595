# scalar-cache / instruction-cache counters of the kernels whose name contains <substr>: tools/pmc_sqc.sh <tag> <substr> <script> [args]
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
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
pass p4 SQ_INSTS_SMEM SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES GRBM_GUI_ACTIVE
pass p5 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
for p in ("p4", "p5"):
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
grep -i "SQC_\|SMEM" $OUT/counters.txt | head -40
find $OUT -name "*counter_collection.csv" -size +20M -delete
