"""umi_group on strings beyond one code word: n strings of `length` bases in families of 8 (1 % substitutions), one pre-group.
python tools/perf_umi_long.py [n] [length] [threshold]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sarlacc_amd import calls
from sarlacc_amd.strset import StringSet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(5)
nuc = np.frombuffer(b"ACGT", np.uint8)
fam = nuc[rng.integers(0, 4, (n // 8 + 1, L))]
rows = np.repeat(fam, 8, axis=0)[:n].copy()
m = rng.random(rows.shape) < 0.01
rows[m] = nuc[rng.integers(0, 4, int(m.sum()))]
umis = StringSet(rows.reshape(-1), np.arange(0, (n + 1) * L, L, dtype=np.int64))
for rep in range(2):
    t0 = time.perf_counter()
    coff, cmem = calls.umi_group_flat(umis, thr, None, thr, np.array([0, n], np.int64), np.arange(1, n + 1, dtype=np.int32))
    print("n %d length %d threshold %d: %.3f s, %d clusters" % (n, L, thr, time.perf_counter() - t0, coff.size - 1), flush=True)
