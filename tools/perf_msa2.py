"""Spec v2 on pure clusters (G groups x 10 reads x L bp): stage timers.  python tools/perf_msa2.py [G] [L] [copies]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls, _lib
from perf_pipeline import NUC, noisy_copies

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(1000)
reads, quals = noisy_copies(NUC[rng.integers(0, 4, (G, L))], K, rng)
n = len(reads)
goff = np.arange(0, n + 1, K, dtype=np.int64)
gflat = np.arange(1, n + 1, dtype=np.int32)
enc = sarlacc_amd.phred_encoding()
for rep in range(2):
    t0 = time.perf_counter()
    cons, _ = calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
    dt = time.perf_counter() - t0
    print("rep %d: %.3f s wall | pairwise %.1f ms (%d pairs) | merge %.1f ms | consensus %.1f ms" % (
        rep, dt, _lib.stage_ms("msa_pairwise"), _lib.stage_count("msa_pairs"), _lib.stage_ms("msa_merge"), _lib.stage_ms("consensus")), flush=True)
