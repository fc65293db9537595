"""Scale check of the umiGroup stage (BASELINE config 3 shape): n 12-bp UMIs, 10 reads
per molecule, mockReads error process, one pre-group.  Compares with the oracle at a
size the oracle can finish, then times the GPU path at full size."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls
from sarlacc_amd.strset import StringSet


def make_umis(nmol, reads_per, seed, sub=0.05, indel=0.01):
    rng = np.random.default_rng(seed)
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    truth = nuc[rng.integers(0, 4, (nmol, 12))]
    r = np.repeat(truth, reads_per, axis=0)
    s = rng.random(r.shape) < sub
    r[s] = nuc[rng.integers(0, 4, int(s.sum()))]
    counts = np.ones(r.shape, np.int64)
    ind = rng.random(r.shape) < indel
    ch = np.array([0, 2, 3, 4, 5])
    counts[ind] = ch[rng.integers(0, 5, int(ind.sum()))]
    flat = np.repeat(r.reshape(-1), counts.reshape(-1))
    lens = counts.sum(1)
    off = np.zeros(len(lens) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    perm = rng.permutation(len(lens))
    ss = StringSet(flat, off).subset(perm)
    return ss


if __name__ == "__main__":
    n_check = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n_full = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    from oracle import oracle as O
    for thr in (1, 2, 3):   # 3 = umiGroup's default threshold1 (R/umiGroup.R:3)
        ss = make_umis(n_check // 10, 10, 1000)
        strs = ss.to_strings()
        g = [np.arange(1, len(strs) + 1, dtype=np.int32)]
        t0 = time.perf_counter(); got = calls.umi_group(ss, thr, None, thr, g); t1 = time.perf_counter()
        want = O.umi_group(strs, thr, None, thr, g, fast=True); t2 = time.perf_counter()
        same = len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
        print("thr=%d n=%d gpu %.3fs oracle %.3fs clusters %d identical=%s" % (thr, len(strs), t1 - t0, t2 - t1, len(got), same), flush=True)
    ss = make_umis(n_full // 10, 10, 1001)
    g = [np.arange(1, len(ss) + 1, dtype=np.int32)]
    for thr in (1, 2, 3):
        for rep in range(2):
            t0 = time.perf_counter(); got = calls.umi_group(ss, thr, None, thr, g); t1 = time.perf_counter()
            tot = sum(len(c) for c in got)
            print("thr=%d n=%d gpu %.3fs clusters %d covered %d pair-kernel %.1f ms" % (thr, len(ss), t1 - t0, len(got), tot, sarlacc_amd.last_kernel_ms()), flush=True)
