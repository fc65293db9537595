"""GPU parity of the alignment-profiling routines through the C ABI: the literal alignment strings of
the reference's tests (tests/testthat/test-homopolymer.R, test-error.R; tests/profile_cases.py) and random gapped
alignments against the CPU oracle -- integer lists, identical including their order."""
import numpy as np
import pytest

from tests.profile_cases import ERROR_CASES, FIND_SEQS, MATCH_CASES, checkfun, findcheck, matchcheck

pytestmark = pytest.mark.gpu


def same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (x if isinstance(x, (list, str)) else np.asarray(x).tolist()) == (y if isinstance(y, (list, str)) else np.asarray(y).tolist())


def gapped_pairs(rng, n, L, p_gap=0.08):
    """n pairwise alignments of noisy reads against one reference (gaps on either side, never both)"""
    ref = rng.choice(list("ACGT"), L)
    # homopolymer-rich reference
    for _ in range(L // 12):
        k = int(rng.integers(0, L - 6))
        ref[k:k + int(rng.integers(2, 7))] = ref[k]
    refs, reads = [], []
    for _ in range(n):
        a, b = [], []
        for c in ref:
            u = rng.random()
            if u < p_gap / 2:                     # deletion in the read
                a.append(c); b.append("-")
            elif u < p_gap:                        # insertion in the read
                k = int(rng.integers(1, 4))
                a += ["-"] * k; b += list(rng.choice(list("ACGT"), k))
                a.append(c); b.append(c)
            else:
                a.append(c); b.append(c if rng.random() > 0.06 else "ACGT"[int(rng.integers(0, 4))])
        if rng.random() < 0.3:
            k = int(rng.integers(1, 5)); a += ["-"] * k; b += list(rng.choice(list("ACGT"), k))
        if rng.random() < 0.3:
            k = int(rng.integers(1, 5)); a = ["-"] * k + a; b = list(rng.choice(list("ACGT"), k)) + b
        refs.append("".join(a)); reads.append("".join(b))
    return refs, reads


def test_find_homopolymers(oracle):
    from sarlacc_amd import calls, generics
    got = calls.find_homopolymers(FIND_SEQS)
    want = findcheck(FIND_SEQS)
    assert [np.asarray(x).tolist() for x in got[:3]] == [want[0], want[1], want[2]] and got[3] == want[3]
    same(got, oracle.find_homopolymers(FIND_SEQS))
    rng = np.random.default_rng(5)
    refs, reads = gapped_pairs(rng, 300, 400)
    for seqs in (refs, reads, refs + [""] + ["----"] + ["A"] + ["-A-A-"], []):
        same(calls.find_homopolymers(seqs), oracle.find_homopolymers(seqs))
    runs = generics.homopolymerFinder(FIND_SEQS[:1])[0]
    assert runs == [(3, 6, "G"), (7, 8, "C"), (10, 12, "T"), (13, 14, "A")]


def test_match_homopolymers(oracle):
    from sarlacc_amd import SarlaccError, calls, generics
    for reads, refs in MATCH_CASES:
        for rd, rf in zip(reads, refs):
            got = calls.match_homopolymers([rf], [rd])
            pos, rlen = matchcheck(rf, rd)
            assert np.asarray(got[1]).tolist() == pos and np.asarray(got[2]).tolist() == rlen, (rd, rf)
        same(calls.match_homopolymers(refs, reads), oracle.match_homopolymers(refs, reads))
    rng = np.random.default_rng(6)
    refs, reads = gapped_pairs(rng, 500, 300)
    same(calls.match_homopolymers(refs, reads), oracle.match_homopolymers(refs, reads))
    same(calls.match_homopolymers(refs + [""], reads + [""]), oracle.match_homopolymers(refs + [""], reads + [""]))
    out = generics.homopolymerMatcher(refs, reads)
    assert len(out) == len(generics.homopolymerFinder([refs[0]])[0]) and all(len(r["observed"]) == len(refs) for r in out)
    with pytest.raises(SarlaccError, match="lengths of alignment vectors should match up"):
        calls.match_homopolymers(["AACC"], [])
    with pytest.raises(SarlaccError, match="equal length"):
        calls.match_homopolymers(["AACC", "AA"], ["AACC", "A"])


def test_find_errors(oracle):
    from sarlacc_amd import SarlaccError, calls, generics
    for reads, refs in ERROR_CASES:
        for rd, rf in zip(reads, refs):
            got = calls.find_errors([rf], [rd])
            want = checkfun([rf], [rd])
            assert got[0] == want[0] and [np.asarray(x).tolist() for x in got[1:6]] == list(want[1:6]), (rd, rf)
            same(got, oracle.find_errors([rf], [rd]))
    rng = np.random.default_rng(7)
    refs, reads = gapped_pairs(rng, 2000, 250)
    got = calls.find_errors(refs, reads)
    same(got, oracle.find_errors(refs, reads))
    want = checkfun(refs, reads)
    assert got[0] == want[0] and [np.asarray(x).tolist() for x in got[1:6]] == list(want[1:6])
    ef = generics.errorFinder(refs, reads)
    assert ef["transition"].sum() == sum(int(np.asarray(got[k]).sum()) for k in (1, 2, 3, 4))
    assert all(len(v) == len(refs) for v in ef["full"]["insertion"])
    same(calls.find_errors([], []), oracle.find_errors([], []))
    # errors in the order the reference's loop meets them
    for rf, rd, msg in ((["ACGT"], ["ACG"], "equal length"), (["ACGT", "ACGTA"], ["ACGT", "ACGTA"], "same for all alignments"),
                        (["ACGT", "ACGT"], ["ACGT", "ACNT"], "unknown character 'N'"), (["ACGT"], [], "should match up"),
                        (["ACGT", "ACGT", "AC"], ["ACGT", "ANGT", "A"], "unknown character 'N'"),
                        (["ACGT", "AC", "ACGT"], ["ACGT", "A", "ANGT"], "equal length"),
                        (["ACGT", "ACGTAA"], ["ACGT", "ACGTNA"], "same for all alignments")):
        with pytest.raises(SarlaccError, match=msg):
            calls.find_errors(rf, rd)
        with pytest.raises(oracle.OracleError, match=msg):
            oracle.find_errors(rf, rd)
