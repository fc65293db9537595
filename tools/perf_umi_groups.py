"""umi_group with many small pre-groups (the vignette's usage): G groups of ~m reads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sarlacc_amd import calls
from tools.perf_umi import make_umis
from oracle import oracle as O

for G, m in ((2000, 20), (50000, 20)):
    ss = make_umis(G * m // 10, 10, 7)
    n = len(ss)
    rng = np.random.default_rng(1)
    labels = rng.integers(0, G, n)
    order = np.argsort(labels, kind="stable")
    bounds = np.searchsorted(labels[order], np.arange(G + 1))
    groups = [(order[bounds[g]:bounds[g + 1]] + 1).astype(np.int32) for g in range(G)]
    t0 = time.perf_counter(); got = calls.umi_group(ss, 1, None, 1, groups); t1 = time.perf_counter()
    msg = "G=%d n=%d gpu %.3fs clusters %d" % (G, n, t1 - t0, len(got))
    if G <= 2000:
        want = O.umi_group(ss.to_strings(), 1, None, 1, groups, fast=True)
        msg += " identical=%s" % (len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want)))
    print(msg, flush=True)
