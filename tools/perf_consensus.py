"""Consensus vote at BASELINE C4 shape on resident reads: the three kernels of the quality vote -- generic characters
(k_consensus_q4), fast characters (k_consensus_qf), vote codes written by the MSA stage (k_consensus_code, the default of
the fused calls) -- same strings required, kernel ms from HIP events; "merge" = the MSA stage's row writer and merge.
Groups are the molecules themselves (known, in read order), MSA under spec v1 so the run is short.
    python tools/perf_consensus.py [molecules] [copies] [read_len]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sarlacc_amd
from sarlacc_amd import _lib, calls, device, devsynth


def main():
    nocheck = "--nocheck" in sys.argv      # timing / profiling only: no string comparison
    if nocheck:
        sys.argv.remove("--nocheck")
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    dev = torch.device("cuda:0")
    mol = devsynth.make_molecule_reads(G, copies, L, seed=5, device=dev)
    off = mol["off"].cpu().numpy()
    n = off.size - 1
    goff = np.arange(0, n + 1, copies, dtype=np.int64)
    gflat = np.arange(1, n + 1, dtype=np.int32)
    enc = sarlacc_amd.phred_encoding()
    calls.set_msa_spec(1)
    res = {}
    # generic: characters, k_consensus_q4; fast: characters, k_consensus_qf (+ q4 on flagged groups);
    # codes: the MSA stage writes 16-bit vote codes, k_consensus_code (what the fused calls do by default)
    for mode in ("generic", "fast", "codes", "generic", "fast", "codes"):
        calls.set_option("consensus_chars", 0 if mode == "codes" else 1)
        calls.set_option("consensus_generic", 1 if mode == "generic" else 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cons, phred = device.dev_msa_consensus(goff, gflat, mol["seq"], mol["qual"], off, 0, -1, -5, -1, 100, 0.6, encoding=enc)
        dt = time.perf_counter() - t0
        ms = _lib.stage_ms("consensus")
        cells = _lib.stage_count("consensus_cells")
        print("%-8s consensus %.3f ms  %.0f cells  %.1f GB/s algorithmic (2 B/cell)  merge %.2f ms  call %.2f s" % (
            mode, ms, cells, 2 * cells / ms / 1e6, _lib.stage_ms("msa_merge"), dt), flush=True)
        if not nocheck:
            res[mode] = (cons.to_strings(), phred.to_strings())
    if nocheck:
        return
    same = res["generic"] == res["fast"] == res["codes"]
    print("identical strings:", same, "groups", len(res["fast"][0]))
    if not same:
        a, b = res["generic"], res["fast"]
        for k in range(len(a[0])):
            if a[0][k] != b[0][k] or a[1][k] != b[1][k]:
                print("first difference in group", k, len(a[0][k]), len(b[0][k]))
                for i, (x, y) in enumerate(zip(a[0][k], b[0][k])):
                    if x != y:
                        print(" base", i, x, y); break
                for i, (x, y) in enumerate(zip(a[1][k], b[1][k])):
                    if x != y:
                        print(" phred", i, x, y); break
                break
        sys.exit(1)


if __name__ == "__main__":
    main()
