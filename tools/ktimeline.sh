# kernel timeline of the LAST repetition of a python command (gaps between kernels show host waits): tools/ktimeline.sh <tag> <script> [args]
set -euo pipefail
TAG=$1; SCRIPT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$TAG" -o tl -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "$@" > "$GRAFT_REPO_ROOT/gpurun_out/$TAG.log" 2>&1
cd "$GRAFT_REPO_ROOT"
tail -n 3 "gpurun_out/$TAG.log" | cut -c1-300
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("gpurun_out/$TAG/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last pass: from the last k_sk_* / umi kernel block backwards -- simply the last 1.2 s of the trace
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - 1_200_000_000]
def short(n):
    n = n.replace("void ", "").replace("sarlacc::", "")
    return n.split("(")[0][:46]
if not rows:
    raise SystemExit("no kernel in the trace (did the command run on the GPU?)")
out, prev_end, acc = [], None, {}
for s, e, n in rows:
    gap = 0 if prev_end is None else (s - prev_end) / 1e6
    d = (e - s) / 1e6
    if d >= 0.3 or gap >= 0.15:
        out.append("%9.3f ms  +gap %7.3f  %8.3f ms  %s" % ((s - rows[0][0]) / 1e6, gap, d, short(n)))
    prev_end = max(prev_end or e, e)
print("\n".join(out))
busy = sum(e - s for s, e, n in rows) / 1e6
print("kernel time summed %.1f ms over a window of %.1f ms (overlapping streams count twice)" % (busy, (rows[-1][1] - rows[0][0]) / 1e6))
PY
find "gpurun_out/$TAG" -name "*kernel_trace.csv" -delete
