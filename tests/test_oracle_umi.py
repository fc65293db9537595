"""Pins the masked-Levenshtein / clustering / umi_group oracle without a GPU.

Follows the reference's own tests (tests/testthat/test-levenshtein.R,
test-umicluster.R): brute-force Levenshtein stands in for Biostrings::stringDist, the
R function REF (test-umicluster.R:4-29) is restated in Python, umi_group must equal
the composition fast_levdist -> intersect -> cluster (REFVERSION, :106-120), and the
literal Rd examples (man/umiGroup.Rd:49-65) must give the outputs recorded in
SURVEY.md section 8c.
"""
import numpy as np
import pytest

RANK = {c: i for i, c in enumerate("ACGTN")}


def lev2(a, b):
    """weighted distance x2: indel 2, substitution 2, anything-vs-N 1 (incl. N-vs-N)."""
    prev = [2 * i for i in range(len(a) + 1)]
    for ch in b:
        cur = [prev[0] + 2]
        for i, c in enumerate(a):
            s = 1 if (c == "N" or ch == "N") else (0 if c == ch else 2)
            cur.append(min(prev[i + 1] + 2, cur[i] + 2, prev[i] + s))
        prev = cur
    return prev[-1]


def trie_key(s):
    return [RANK[c] for c in s]


def brute_neighbours(seqs, limit):
    out = []
    for i, a in enumerate(seqs):
        hits = [j for j, b in enumerate(seqs) if lev2(a, b) <= 2 * limit]
        hits.sort(key=lambda j: (trie_key(seqs[j]), j))   # prefix first, ties by index
        out.append(np.array(hits, dtype=np.int32) + 1)
    return out


def seqsim(rng, n, lo, hi, alphabet="ACGT"):
    return ["".join(rng.choice(list(alphabet), int(rng.integers(lo, hi + 1)))) for _ in range(n)]


def test_lev_masked_equals_levenshtein(oracle):
    rng = np.random.default_rng(1000)
    for lo, hi in ((1, 20), (5, 10)):
        seqs = seqsim(rng, 60, lo, hi)
        out = oracle.compute_lev_masked(seqs)
        want = [lev2(seqs[i], seqs[j]) / 2 for i in range(len(seqs)) for j in range(i + 1, len(seqs))]
        assert out.tolist() == want


def test_lev_masked_known_answers(oracle):
    # tests/testthat/test-levenshtein.R:31-46
    ref = "ACAGCTAGC"
    for i in range(len(ref)):
        masked = ref[:i] + "N" + ref[i + 1:]
        assert oracle.compute_lev_masked([ref, masked]).tolist() == [0.5]
        assert oracle.compute_lev_masked([ref, ref]).tolist() == [0.0]
        assert oracle.compute_lev_masked([masked, masked]).tolist() == [0.5]
    assert oracle.compute_lev_masked(["ACGT"]).size == 0


@pytest.mark.parametrize("lo,hi,dup", [(1, 20, False), (5, 10, False), (5, 10, True), (0, 5, False)])
def test_fast_levdist_equals_brute_force(oracle, lo, hi, dup):
    rng = np.random.default_rng(lo * 100 + hi + dup)
    seqs = seqsim(rng, 50 if dup else 100, lo, hi)
    if dup:
        seqs = [seqs[i] for i in rng.integers(0, 50, 100)]
    for limit in ((1, 2) if hi == 5 else (5, 2, 1)):
        got = oracle.fast_levdist_test(seqs, limit)
        want = brute_neighbours(seqs, limit)
        for g, w in zip(got, want):
            assert g.tolist() == w.tolist()


def test_fast_levdist_masked(oracle):
    # tests/testthat/test-levenshtein.R:122-138
    ref = "ACAGCTAGC"
    for i in range(len(ref)):
        masked = ref[:i] + "N" + ref[i + 1:]
        out = oracle.fast_levdist_test([ref, masked], 1)
        assert sorted(out[0].tolist()) == [1, 2] and sorted(out[1].tolist()) == [1, 2]
        out = oracle.fast_levdist_test([ref, masked], 0)
        assert out[0].tolist() == [1] and out[1].tolist() == []
    # N-containing random sets against brute force
    rng = np.random.default_rng(5)
    seqs = seqsim(rng, 80, 4, 9, "ACGTN")
    for limit in (0, 1, 2, 3):
        for g, w in zip(oracle.fast_levdist_test(seqs, limit), brute_neighbours(seqs, limit)):
            assert g.tolist() == w.tolist()


# ---------------------------------------------------------------------------
def ref_cluster(groups):
    """Python restatement of REF (tests/testthat/test-umicluster.R:4-29), 1-based lists."""
    groups = [list(g) for g in groups]
    collected = []
    for _ in range(len(groups)):
        sizes = [len(g) for g in groups]
        mx = max(sizes)
        if mx == 0:
            break
        chosen = max(i for i, s in enumerate(sizes) if s == mx)   # last, if ties
        cur = groups[chosen]
        collected.append(list(cur))
        for j in cur:
            groups[j - 1] = []
        cs = set(cur)
        groups = [[x for x in g if x not in cs] for g in groups]
    return collected


def mockup(rng, n, density):
    a = rng.random((n, n)) < density
    a = np.triu(a, 1)
    a = a | a.T | np.eye(n, dtype=bool)
    return [(np.flatnonzero(a[:, j]) + 1).tolist() for j in range(n)]


@pytest.mark.parametrize("n,density", [(20, 0.05), (20, 0.1), (20, 0.2), (50, 0.2), (50, 0.4), (50, 0.1), (50, 0.0)])
def test_cluster_matches_R_reference(oracle, n, density):
    rng = np.random.default_rng(int(n * 1000 + density * 100))
    for _ in range(5):
        links = mockup(rng, n, density)
        ref = ref_cluster(links)
        obs = [c.tolist() for c in oracle.cluster_umis_test(links)]
        # COMPARE (test-umicluster.R:43-49): same clusters as sets of lists
        assert len(ref) == len(obs)
        assert sorted(map(tuple, ref)) == sorted(map(tuple, obs))
        # solos first in index order, then picks by (count desc, index desc) (App.B Q12)
        solos = [c for c in obs if len(links[c[0] - 1]) == 1]
        assert obs[:len(solos)] == solos and [c[0] for c in solos] == sorted(c[0] for c in solos)
        fast = [c.tolist() for c in oracle.cluster_umis_test(links, fast=True)]
        assert fast == obs


def test_cluster_errors(oracle):
    with pytest.raises(oracle.OracleError, match="zero length read group"):
        oracle.cluster_umis_test([[1], []])
    with pytest.raises(oracle.OracleError, match="single-read groups should contain only the read itself"):
        oracle.cluster_umis_test([[2], [1, 2]])


def umisim(rng, n, length):
    ref = rng.choice(list("ACGT"), length)
    out = []
    for _ in range(n):
        t = ref.copy()
        ch = rng.random(length) < 0.1
        t[ch] = rng.choice(list("ACGT"), int(ch.sum()))
        out.append("".join(t))
    return out


def refversion(oracle, u1, t1, u2=None, t2=None):
    out1 = oracle.fast_levdist_test(u1, t1)
    if u2 is not None:
        out2 = oracle.fast_levdist_test(u2, t2)
        out1 = [np.array([x for x in b if x in set(a.tolist())], dtype=np.int32) for a, b in zip(out1, out2)]
    return oracle.cluster_umis_test([x.tolist() for x in out1])


def test_umi_group_composition(oracle):
    rng = np.random.default_rng(77)
    seqs1, seqs2, pre = [], [], []
    for x in range(10):
        N = int(rng.integers(10, 21))
        seqs1 += umisim(rng, N, 10)
        seqs2 += umisim(rng, N, 5)
        pre += [x] * N
    o = rng.permutation(len(pre))
    seqs1 = [seqs1[i] for i in o]
    seqs2 = [seqs2[i] for i in o]
    pre = np.array(pre)[o]
    allidx = [list(range(1, len(seqs1) + 1))]
    for t in (1, 3):
        obs = oracle.umi_group(seqs1, t, None, t, allidx)
        ref = refversion(oracle, seqs1, t)
        assert [a.tolist() for a in obs] == [b.tolist() for b in ref]
        assert [a.tolist() for a in oracle.umi_group(seqs1, t, None, t, allidx, fast=True)] == [a.tolist() for a in obs]
    for t1, t2 in ((1, 1), (3, 1)):
        obs = oracle.umi_group(seqs1, t1, seqs2, t2, allidx)
        ref = refversion(oracle, seqs1, t1, seqs2, t2)
        assert [a.tolist() for a in obs] == [b.tolist() for b in ref]
    # with pre-groups (test-umicluster.R:140-157)
    by_group = [(np.flatnonzero(pre == g) + 1).tolist() for g in range(10)]
    obs = oracle.umi_group(seqs1, 1, None, 1, by_group)
    want = []
    for idx in by_group:
        sub = refversion(oracle, [seqs1[i - 1] for i in idx], 1)
        want += [[idx[v - 1] for v in c] for c in sub]
    assert [a.tolist() for a in obs] == want
    # every read appears exactly once
    assert sorted(x for c in obs for x in c.tolist()) == list(range(1, len(seqs1) + 1))
    # solo pre-groups pass straight through (test-umicluster.R:159-165)
    out = oracle.umi_group(seqs1[:10], 1, None, 1, [[i] for i in range(1, 11)])
    assert [c.tolist() for c in out] == [[i] for i in range(1, 11)]


def test_umi_group_rd_examples(oracle):
    # man/umiGroup.Rd:49-65 with the outputs recorded in SURVEY.md section 8c
    u1 = ["AACCGGTT", "AACGGTT", "ACCCGGTT", "AACCGGTTT"]
    u2 = ["AACCGGTT", "CACCGGTT", "AACCCGGTA", "AACGGGTT"]
    g = [[1, 2, 3, 4]]
    assert [c.tolist() for c in oracle.umi_group(u1, 3, None, 3, g)] == [[1, 4, 2, 3]]
    assert [c.tolist() for c in oracle.umi_group(u1, 0, None, 0, g)] == [[1], [2], [3], [4]]
    assert [c.tolist() for c in oracle.umi_group(u1, 3, u2, 3, g)] == [[3, 1, 4, 2]]
