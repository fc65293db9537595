"""The own rules of MSA spec v2, step 5 -- the bounded library (the direct partner and three further
positions per pair and base), the row cap of 16 partner columns and the noise filter -- bounded on the CPU statement of the spec (oracle/msa2.c carries switches for them and counters of how often
they act).  PARITY WITH THE REFERENCE IS UNPINNED for this stage (SeqAn is absent, the reference never tests quick_msa);
what can be pinned is what our own departures from SeqAn's pipeline change.  tools/msa2_rules.py is the long version
(C4-shaped clusters, five seeds; profiles/r05_msa2_rules_v2.txt)."""
import numpy as np
import pytest

PARAMS = (0, -1, -5, -1, 100)


def clusters(kind, seed, n):
    from sarlacc_amd.mock import NUC, mutate
    rng = np.random.default_rng(seed)
    reads, groups, truths = [], [], []
    for _ in range(n):
        if kind == "pure":
            t, src, rate = [NUC[rng.integers(0, 4, 500)]], [0] * 8, (0.05, 0.01)
        elif kind == "mixed":
            t, src, rate = [NUC[rng.integers(0, 4, 400)], NUC[rng.integers(0, 4, 400)]], [0] * 9 + [1] * 3, (0.05, 0.01)
        else:
            t, src, rate = [NUC[rng.integers(0, 4, 300)]], [0] * int(rng.integers(3, 6)), (0.10, 0.03)
        idx = []
        for m in rng.permutation(src):
            reads.append(mutate(t[m], rng, *rate).tobytes().decode())
            idx.append(len(reads))
        groups.append(idx)
        truths.append(t[0].tobytes().decode())
    return reads, groups, truths


def run(oracle, reads, groups, nocap, nofilter):
    oracle.msa2_set_rules(nocap, nofilter)
    oracle.msa2_stats()
    try:
        aln = oracle.quick_msa(groups, reads, *PARAMS)
    finally:
        oracle.msa2_set_rules(False, False)
    return aln, oracle.msa2_stats()


def error(oracle, aln, truths):
    cons, _ = oracle.create_consensus_basic_loop(aln, 0.6, 1)
    return sum(float(oracle.compute_lev_masked([c, t])[0]) for c, t in zip(cons, truths)) / sum(len(t) for t in truths)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_row_cap_is_inert_on_same_molecule_clusters(oracle, seed):
    reads, groups, truths = clusters("pure", 100 + seed, 8)
    spec, st = run(oracle, reads, groups, False, False)
    assert st["rows_capped"] == 0 and st["candidates_ignored_by_cap"] == 0 and st["max_row_entries"] <= 16
    assert st["entries_kept"] == st["entries_before_filter"] - st["entries_filtered"]
    nocap, _ = run(oracle, reads, groups, True, False)
    assert nocap == spec
    # the filter moves gaps in some clusters; the consensus does not get worse with it
    nofilter, st2 = run(oracle, reads, groups, False, True)
    assert st2["entries_filtered"] == 0
    assert error(oracle, spec, truths) <= error(oracle, nofilter, truths) + 1e-3


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_rules_on_clusters_of_two_molecules_and_on_hard_clusters(oracle, seed):
    """Clusters of two molecules: both rules act (that is what they are for); the consensus against the majority molecule
    stays within a few edits of what the unbounded library gives.  Hard clusters: the cap is inert, the filter harmless."""
    reads, groups, truths = clusters("mixed", 200 + seed, 4)
    spec, st = run(oracle, reads, groups, False, False)
    assert st["rows_capped"] > 0 and st["entries_filtered"] > 0
    free, stf = run(oracle, reads, groups, True, True)
    assert stf["rows_capped"] == 0 and stf["entries_filtered"] == 0 and stf["max_row_entries"] > 16
    assert abs(error(oracle, spec, truths) - error(oracle, free, truths)) < 5e-3
    for rows, g in zip(spec, groups):
        assert len({len(r) for r in rows}) == 1 and [r.replace("-", "") for r in rows] == [reads[i - 1] for i in g]
    reads, groups, truths = clusters("hard", 300 + seed, 10)
    spec, st = run(oracle, reads, groups, False, False)
    assert st["rows_capped"] == 0
    free, _ = run(oracle, reads, groups, True, True)
    assert error(oracle, spec, truths) <= error(oracle, free, truths) + 5e-3


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bounded_library_against_the_unbounded_one(oracle, seed):
    """The library of round 5 keeps the direct partner and the first three other positions per (a, p, b); rounds 2-4 kept every
    position (oracle.msa2_set_library(-1) restores their enumeration).  Same-molecule clusters: the bound ignores next to nothing
    and the consensus is the same; clusters of two molecules: it ignores most of the noise the third reads of the other
    molecule name, and the consensus against the majority molecule stays where the unbounded library put it."""
    def with_library(others, reads, groups):
        oracle.msa2_set_library(others)
        oracle.msa2_stats()
        try:
            return oracle.quick_msa(groups, reads, *PARAMS), oracle.msa2_stats()
        finally:
            oracle.msa2_set_library(3)
    reads, groups, truths = clusters("pure", 400 + seed, 6)
    spec, st = with_library(3, reads, groups)
    free, stf = with_library(-1, reads, groups)
    assert st["library_positions_ignored"] < 0.001 * st["triples"] and stf["library_positions_ignored"] == 0
    assert abs(error(oracle, spec, truths) - error(oracle, free, truths)) < 1e-3
    wide, stw = with_library(63, reads, groups)
    assert stw["library_positions_ignored"] == 0 and abs(error(oracle, wide, truths) - error(oracle, free, truths)) < 1e-3
    reads, groups, truths = clusters("mixed", 500 + seed, 4)
    spec, st = with_library(3, reads, groups)
    free, _ = with_library(-1, reads, groups)
    assert st["library_positions_ignored"] > 0
    assert abs(error(oracle, spec, truths) - error(oracle, free, truths)) < 5e-3
    for rows, g in zip(spec, groups):
        assert len({len(r) for r in rows}) == 1 and [r.replace("-", "") for r in rows] == [reads[i - 1] for i in g]
    assert oracle.quick_msa(groups, reads, *PARAMS) == spec   # (the default is back)
