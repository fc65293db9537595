"""What do the own rules of MSA spec v2 change?  CPU only (oracle/msa2.c with its
rule switches): rows and consensus error against the simulated molecule with the rules as specified, without the row cap
(16 partner columns), without the noise filter (entries lighter than half the row's heaviest) and without both, with a narrower and a wider library (1 + 1, 1 + 2, 1 + 63 partner positions per (a, p, b)) and with the unbounded library and
row-major enumeration of rounds 2-4, on
  pure       same-molecule clusters, 10 reads x 2 kb, mockReads error process (BASELINE config 4 shape)
  mixed      clusters of two molecules, 9 + 3 reads x 1 kb (a UMI collision: the consensus should be the majority's)
  hard       3-5 reads x 400 bases, 10 % substitutions, 3 % indel events, every third cluster with a chimeric read
over several seeds; counters of how often each rule acts.
    python tools/msa2_rules.py [seeds] [pure clusters per seed] > profiles/r03_msa2_rules_v1.txt"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as O
from sarlacc_amd.mock import NUC, mutate

PARAMS = (0, -1, -5, -1, 100)
# (name, row cap off, noise filter off, partner positions per (a, p, b) beside the direct one; -1: the unbounded library of rounds 2-4)
MODES = [("spec v2 (library 1+3, cap 16, filter)", False, False, 3), ("no cap", True, False, 3), ("no filter", False, True, 3),
         ("neither rule", True, True, 3), ("library 1+1", False, False, 1), ("library 1+2", False, False, 2), ("library 1+63", False, False, 63),
         ("library 1+63, neither rule", True, True, 63), ("rounds 2-4 (unbounded library, cap 16, filter)", False, False, -1),
         ("rounds 2-4, neither rule", True, True, -1)]


def make(kind, rng, n):
    reads, groups, truths = [], [], []
    for k in range(n):
        if kind == "pure":
            t = [NUC[rng.integers(0, 4, 2000)]]
            src = [0] * 10
            rate = (0.05, 0.01)
        elif kind == "mixed":
            t = [NUC[rng.integers(0, 4, 1000)], NUC[rng.integers(0, 4, 1000)]]
            src = [0] * 9 + [1] * 3
            rate = (0.05, 0.01)
        else:
            t = [NUC[rng.integers(0, 4, 400)]]
            src = [0] * int(rng.integers(3, 6))
            rate = (0.10, 0.03)
        idx = []
        for r, m in enumerate(rng.permutation(src)):
            x = mutate(t[m], rng, *rate)
            if kind == "hard" and r == 0 and k % 3 == 0:   # chimera: the second half is unrelated sequence
                x = np.concatenate([x[:len(x) // 2], NUC[rng.integers(0, 4, int(rng.integers(100, 300)))]])
            reads.append(x.tobytes().decode())
            idx.append(len(reads))
        groups.append(idx)
        truths.append(t[0].tobytes().decode())
    return reads, groups, truths


def run(reads, groups, cores):
    with ThreadPoolExecutor(cores) as ex:
        return list(ex.map(lambda g: O.quick_msa([g], reads, *PARAMS)[0], groups))


def errors(aln, truths, cores):
    cons, _ = O.create_consensus_basic_loop(aln, 0.6, 1)
    with ThreadPoolExecutor(cores) as ex:
        d = list(ex.map(lambda ct: float(O.compute_lev_masked([ct[0], ct[1]])[0]) if ct[0] else float(len(ct[1])), zip(cons, truths)))
    return np.array(d), np.array([len(t) for t in truths], float)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    npure = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cores = min(8, os.cpu_count() or 1)
    O.build()
    print("MSA spec v2: effect of its two own rules (oracle/msa2.c, CPU), %d seeds" % seeds)
    for kind, n in (("pure", npure), ("mixed", 12), ("hard", 24)):
        tot = {m[0]: [0.0, 0.0] for m in MODES}
        same = {m[0]: 0 for m in MODES}
        cols_diff = {m[0]: 0.0 for m in MODES}
        ngroups = 0
        stats = None
        t0 = time.time()
        for seed in range(seeds):
            rng = np.random.default_rng(9000 + 17 * seed + len(kind))
            reads, groups, truths = make(kind, rng, n)
            base = None
            for name, nocap, nofilter, library in MODES:
                O.msa2_set_rules(nocap, nofilter)
                O.msa2_set_library(library)
                O.msa2_stats()
                try:
                    aln = run(reads, groups, cores)
                finally:
                    O.msa2_set_rules(False, False)
                    O.msa2_set_library(3)
                st = O.msa2_stats()
                if base is None:
                    base = aln
                    stats = st if stats is None else {k: (max(stats[k], v) if k == "max_row_entries" else stats[k] + v) for k, v in st.items()}
                d, ln = errors(aln, truths, cores)
                tot[name][0] += d.sum()
                tot[name][1] += ln.sum()
                same[name] += sum(a == b for a, b in zip(aln, base))
                cols_diff[name] += sum(abs(len(a[0]) - len(b[0])) for a, b in zip(aln, base) if a and b)
            ngroups += len(groups)
        print("\n%s clusters: %d (%.0f s)" % (kind, ngroups, time.time() - t0))
        print("  rule activity under the spec: rows %d, rows that met a 17th partner column %d (%.3f %%), candidates ignored by the cap %d,"
              % (stats["rows"], stats["rows_capped"], 100.0 * stats["rows_capped"] / max(stats["rows"], 1), stats["candidates_ignored_by_cap"]))
        print("    entries before the filter %d, dropped by it %d (%.1f %%) in %d rows (%.1f %% of the rows); most entries in a row %d"
              % (stats["entries_before_filter"], stats["entries_filtered"], 100.0 * stats["entries_filtered"] / max(stats["entries_before_filter"], 1),
                 stats["rows_filtered"], 100.0 * stats["rows_filtered"] / max(stats["rows"], 1), stats["max_row_entries"]))
        print("    (a, p, b) triples %d, naming two positions of b %d, three or more %d; partner positions the bounded library ignored %d"
              % (stats["triples"], stats["triples_2_positions"], stats["triples_3_positions"], stats["library_positions_ignored"]))
        for name, _, _, _ in MODES:
            e, l = tot[name]
            print("  %-48s consensus error %.3e per base (%g edits / %g bases)   clusters with rows identical to the spec's: %d of %d   "
                  "width difference summed: %g columns" % (name, e / l, e, l, same[name], ngroups, cols_diff[name]))


if __name__ == "__main__":
    main()
