"""What an alignment costs beyond its DP rows: sarlacc_dev_align on 10^6 reads of several lengths, score only and with traceback
+ one section; a line through (length, ms) gives the per-row rate and the fixed part (set-up, traceback, result)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarlacc_amd
from sarlacc_amd import device as sdev, devsynth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
lens = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "125,250,500,1000,2000").split(",")]
dev = torch.device("cuda", 0)
sarlacc_amd.set_device(0)
enc = sarlacc_amd.phred_encoding()
st = torch.cuda.current_stream().cuda_stream
res = {False: [], True: []}
for L in lens:
    seq, qual, off, max_len = devsynth.make_reads(n, L, bench.ADAPTOR1, bench.ADAPTOR2, seed=1, device=dev)
    scores = torch.empty(n, dtype=torch.float64, device=dev)
    starts = torch.empty(n, dtype=torch.int32, device=dev); ends = torch.empty_like(starts)
    sso = torch.empty_like(starts); swo = torch.empty_like(starts)
    for trace in (False, True):
        ms = []
        for _ in range(3):
            if trace:
                sdev.dev_align(seq, qual, off, n, max_len, enc, 5.0, 1.0, bench.ADAPTOR1, True, [9], [21], scores, starts, ends, sso, swo, st)
            else:
                sdev.dev_align(seq, qual, off, n, max_len, enc, 5.0, 1.0, bench.ADAPTOR1, True, (), (), scores, None, None, None, None, st)
            ms.append(sarlacc_amd.last_kernel_ms())
        t = min(ms)
        mean_len = int(off[-1].item()) / n
        res[trace].append((mean_len, t))
        print("L=%d (mean %.0f) trace=%d  %.3f ms  %.0f GCUPS" % (L, mean_len, trace, t, mean_len * n * 30 / t / 1e6), flush=True)
    del seq, qual, off, scores, starts, ends, sso, swo
for trace in (False, True):
    (l0, t0), (l1, t1) = res[trace][0], res[trace][-1]
    b = (t1 - t0) / (l1 - l0)
    print("trace=%d: %.4f ms per row per %d reads, fixed part %.2f ms = %.0f rows" % (trace, b, n, t0 - b * l0, (t0 - b * l0) / b))
