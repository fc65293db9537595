"""A/B timing of the pairwise MSA kernel on the C4 job shape (G groups x 10 reads x L bp), several
variants in one process (SARLACC_MSA_DBG / SARLACC_MSA_INT32 are read at every call):
    python tools/perf_msa.py [G] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls, _lib
from perf_pipeline import NUC, noisy_copies


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = np.random.default_rng(1000)
    reads, quals = noisy_copies(NUC[rng.integers(0, 4, (G, L))], 10, rng)
    n = len(reads)
    goff = np.arange(0, n + 1, 10, dtype=np.int64)
    gflat = np.arange(1, n + 1, dtype=np.int32)
    enc = sarlacc_amd.phred_encoding()
    variants = [("packed", {}), ("int32", {"SARLACC_MSA_INT32": "1"}), ("packed no-walk", {"SARLACC_MSA_DBG": "1"}),
                ("packed no-guard", {"SARLACC_MSA_DBG": "2"}), ("packed no-store", {"SARLACC_MSA_DBG": "4"}),
                ("packed no-walk no-guard no-store", {"SARLACC_MSA_DBG": "7"})]
    for rep in range(3):
        for name, env in variants:
            for k in ("SARLACC_MSA_INT32", "SARLACC_MSA_DBG"):
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
            except Exception as e:  # debug variants produce garbage alignments
                pass
            if rep:
                print("rep %d %-36s pairwise %.2f ms (%d pairs, %.3g cells)" % (rep, name, _lib.stage_ms("msa_pairwise"),
                      _lib.stage_count("msa_pairs"), _lib.stage_count("msa_cells")), flush=True)


if __name__ == "__main__":
    main()
