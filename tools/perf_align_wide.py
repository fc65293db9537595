"""qualityAlign-shaped calls with references beyond 1 024 columns (k_align_wide): n reads of about R bases against an
R-base reference, global alignment.  python tools/perf_align_wide.py [n] [R]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(7)
nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
ref = nuc[rng.integers(0, 4, R)]
reads, quals = [], []
for i in range(n):
    r = ref.copy()
    m = rng.random(R) < 0.08
    r[m] = nuc[rng.integers(0, 4, int(m.sum()))]
    r = np.delete(r, np.nonzero(rng.random(R) < 0.02)[0])
    reads.append(r.tobytes().decode())
    quals.append(rng.integers(45, 80, r.size).astype(np.uint8).tobytes().decode())
enc = sarlacc_amd.phred_encoding()
ref_s = ref.tobytes().decode()
cells = sum(len(r) for r in reads) * R
for mode in ("scores (barcode_align)", "edit only", "strings"):
    for rep in range(2):
        t0 = time.perf_counter()
        if mode.startswith("scores"):
            calls.barcode_align(reads, quals, enc, 5, 1, ref_s)
        else:
            calls.general_align(reads, quals, enc, 5, 1, ref_s, mode == "edit only")
        dt = time.perf_counter() - t0
    k = _lib.last_kernel_ms()
    print("%-24s n %d R %d: %.3f s host call, kernel %.2f ms = %.1f GCUPS" % (mode, n, R, dt, k, cells / max(k, 1e-9) / 1e6), flush=True)
