"""Tuning probe for the DP kernel: times sarlacc_dev_align on a resident batch for
score-only vs traceback and a few launch shapes (options align_k, align_waves_per_cu)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarlacc_amd
from sarlacc_amd import calls, device as sdev, devsynth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dev = torch.device("cuda", 0)
sarlacc_amd.set_device(0)
enc = sarlacc_amd.phred_encoding()
seq, qual, off, max_len = devsynth.make_reads(n, 2000, bench.ADAPTOR1, bench.ADAPTOR2, seed=1, device=dev)
cells = int(off[-1].item()) * 30
scores = torch.empty(n, dtype=torch.float64, device=dev)
starts = torch.empty(n, dtype=torch.int32, device=dev); ends = torch.empty_like(starts)
sso = torch.empty_like(starts); swo = torch.empty_like(starts)
st = torch.cuda.current_stream().cuda_stream

def run(trace, reps=3):
    ms = []
    for _ in range(reps):
        if trace:
            sdev.dev_align(seq, qual, off, n, max_len, enc, 5.0, 1.0, bench.ADAPTOR1, True, [9], [21], scores, starts, ends, sso, swo, st)
        else:
            sdev.dev_align(seq, qual, off, n, max_len, enc, 5.0, 1.0, bench.ADAPTOR1, True, (), (), scores, None, None, None, None, st)
        ms.append(sarlacc_amd.last_kernel_ms())
    return min(ms)

for K in (os.environ.get("KS", "1,2,4").split(",")):
    for w in (os.environ.get("WS", "4,8,12,16").split(",")):
        calls.set_option("align_k", int(K))
        calls.set_option("align_waves_per_cu", int(w))
        for trace in (False, True):
            t = run(trace)
            print("K=%s waves/CU=%s trace=%d  %.2f ms  %.1f GCUPS" % (K, w, trace, t, cells / t / 1e6), flush=True)
