#!/bin/bash
# Memory-path counters (TA / TCP / TCC / EA) of the kernels whose name contains <substr>, one rocprofv3 pass per counter set
# (two counters per TA / TCP pass, four per TCC pass: more exceeds the blocks' slots):
#   tools/pmc_mem.sh <tag> <kernel substr> <script> [args]
# Prints per kernel instantiation the sums over its launches and a few ratios: TA busy share, L1 -> L2 and L2 -> fabric read
# latencies (level / requests), bytes requested from the fabric by request size, credit stalls.
set -euo pipefail
TAG=$1; SUB=$2; SCRIPT=$3; shift; shift; shift
ARGS=("$@")
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
pass() {   # a pass that asks for more than a block's counter slots aborts inside the profiler and then hangs: bounded, and the next pass still runs
    local d=$1; shift
    if ! timeout -k 10 240 rocprofv3 --output-format csv --pmc "$@" -d "$OUT/$d" -o pmc -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "${ARGS[@]}" > "$OUT/$d.log" 2>&1; then
        echo "pass $d FAILED:"; grep -m3 -i "error\|exceeds" "$OUT/$d.log" || tail -n 5 "$OUT/$d.log"
    else
        echo "pass $d done"
    fi
}
cd /tmp
pass m1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass m1b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass m2 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
pass m2b TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass m3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass m4 TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass m4b TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE
pass m5 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE
pass m6 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE
cd "$GRAFT_REPO_ROOT"
python3 - "$OUT" "$SUB" <<'PY'
import csv, glob, sys
out, sub = sys.argv[1], sys.argv[2]
tot, calls = {}, {}
for p in ("m1", "m1b", "m2", "m2b", "m3", "m4", "m4b", "m5", "m6"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, p), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0][-40:]
                c = r["Counter_Name"] if r["Counter_Name"] != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE@" + p
                tot.setdefault(k, {})
                tot[k][c] = tot[k].get(c, 0.0) + float(r["Counter_Value"])
                calls[(k, c)] = calls.get((k, c), 0) + 1
for k in sorted(tot):
    t = tot[k]
    print("== %s" % k)
    for c in sorted(t):
        print("   %-44s %16.6g  (%d launches)" % (c, t[c], calls[(k, c)]))
    g = lambda n: t.get(n, float("nan"))
    def ratio(a, b):   # (a failed pass gives NaN on purpose; a counter that read zero -- no flat reads, no L2 requests -- must not end the printout)
        return a / b if b == b and b != 0 else float("nan")
    cyc = g("GRBM_GUI_ACTIVE@m1") / 8.0   # (the counter comes summed over the 8 XCDs)
    print("   kernel cycles (per XCD)                     %.4g" % cyc)
    print("   TA busy per TA and cycle (256 TAs)          %.3f" % ratio(g("TA_TA_BUSY_sum"), cyc * 256))
    print("   L1 stalled on pending data, per L1 and cycle %.3f" % ratio(g("TCP_PENDING_STALL_CYCLES_sum"), g("GRBM_GUI_ACTIVE@m2b") / 8.0 * 256))
    print("   cycles per wave-wide read in the TAs        %.1f" % ratio(g("TA_TA_BUSY_sum"), g("TA_FLAT_READ_WAVEFRONTS_sum")))
    print("   L1 -> L2 read latency (cycles)              %.0f" % ratio(g("TCP_TCC_READ_REQ_LATENCY_sum"), g("TCP_TCC_READ_REQ_sum")))
    print("   L1 accesses per L1 -> L2 read request       %.2f" % ratio(g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCP_TCC_READ_REQ_sum")))
    print("   L2 hit share of its requests                %.3f" % ratio(g("TCC_HIT_sum"), g("TCC_HIT_sum") + g("TCC_MISS_sum")))
    rd = g("TCC_EA0_RDREQ_sum"); r32 = g("TCC_EA0_RDREQ_32B_sum"); r128 = g("TCC_EA0_RDREQ_128B_sum")
    rbytes = 32 * r32 + 128 * r128 + 64 * (rd - r32 - r128)
    print("   fabric reads: %.4g requests (32 B: %.3g, 128 B: %.3g) = %.4g bytes" % (rd, r32, r128, rbytes))
    print("   L2 -> fabric read latency (cycles)          %.0f" % ratio(g("TCC_EA0_RDREQ_LEVEL_sum"), rd))
    wr = g("TCC_EA0_WRREQ_sum"); w64 = g("TCC_EA0_WRREQ_64B_sum")
    print("   fabric writes: %.4g requests (64 B: %.3g) = %.4g bytes" % (wr, w64, 64 * w64 + 32 * (wr - w64)))
    print("   DRAM read credit stall cycles / (cycles x 16 channels x 8 XCDs) %.3f" % ratio(g("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"), g("GRBM_GUI_ACTIVE@m6") / 8.0 * 128))
PY
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
find "$OUT" -type f ! -name '*.csv' ! -name '*.log' -delete
