cd tools
for PAD in 0 4000 10000 20000 40000 80000; do
  echo "== LDS pad $PAD"
  SARLACC_MSA_LDSPAD=$PAD SARLACC_MSA_DBG=7 python - <<'PY'
import os, sys
sys.path.insert(0, "..")
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls, _lib
from perf_pipeline import NUC, noisy_copies
rng = np.random.default_rng(1000)
reads, quals = noisy_copies(NUC[rng.integers(0, 4, (10000, 2000))], 10, rng)
n = len(reads)
goff = np.arange(0, n + 1, 10, dtype=np.int64); gflat = np.arange(1, n + 1, dtype=np.int32)
enc = sarlacc_amd.phred_encoding()
for rep in range(3):
    try:
        calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
    except Exception as e:
        pass
    if rep: print("  fill only: %.2f ms" % _lib.stage_ms("msa_pairwise"), flush=True)
PY
done
