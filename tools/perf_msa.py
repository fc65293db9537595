"""A/B timing of the pairwise MSA kernel on the C4 job shape (G groups x 10 reads x L bp), several
variants in one process (packed 16-bit kernel, and option msa_int32 = the 32-bit kernel):
    python tools/perf_msa.py [G] [L]"""
import os, sys
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
    variants = [("packed", 0), ("int32", 1)]
    for rep in range(3):
        for name, int32 in variants:
            calls.set_option("msa_int32", int32)
            calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
            if rep:
                print("rep %d %-36s pairwise %.2f ms (%d pairs, %.3g cells)" % (rep, name, _lib.stage_ms("msa_pairwise"),
                      _lib.stage_count("msa_pairs"), _lib.stage_count("msa_cells")), flush=True)


if __name__ == "__main__":
    main()
