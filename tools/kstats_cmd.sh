# rocprofv3 kernel stats of an arbitrary python command: tools/kstats_cmd.sh <tag> <script> [args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o st -- python3 $GRAFT_REPO_ROOT/$@ > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/$TAG.log
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/$TAG/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:10]:
        print("%-70s calls %5s total %9.1f ms avg %8.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
