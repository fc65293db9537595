"""Oracle for the alignment-profiling routines (find_homopolymers, match_homopolymers, find_errors)
against independent restatements of the checkers the reference's tests carry -- FINDCHECK and MATCHCHECK of
tests/testthat/test-homopolymer.R:4-126 and CHECKFUN of tests/testthat/test-error.R:4-36 -- on the literal
alignment strings of those tests (DNAString upper-cases its input, so the literals are upper-cased here)."""
import numpy as np
import pytest

from tests.profile_cases import ERROR_CASES, FIND_SEQS, MATCH_CASES, checkfun, findcheck, matchcheck


def test_find_homopolymers_literals(oracle):
    got = oracle.find_homopolymers(FIND_SEQS)
    want = findcheck(FIND_SEQS)
    assert [np.asarray(x).tolist() for x in got[:3]] == [want[0], want[1], want[2]] and got[3] == want[3]
    assert [np.asarray(x).tolist() if not isinstance(x, list) else x for x in oracle.find_homopolymers([])] == [[], [], [], []]
    # all test sequences spell the same molecule: same homopolymers whatever the gaps
    by_seq = {}
    for i, p, s, b in zip(*[np.asarray(x).tolist() if not isinstance(x, list) else x for x in got]):
        by_seq.setdefault(i, []).append((p, s, b))
    assert by_seq[0] == by_seq[1] == by_seq[2] == by_seq[4] == by_seq[5] == [(3, 4, "G"), (7, 2, "C"), (10, 3, "T"), (13, 2, "A")]
    assert by_seq[3] == [(3, 4, "G"), (7, 2, "C"), (10, 3, "T"), (13, 4, "A")]


@pytest.mark.parametrize("k", range(len(MATCH_CASES)))
def test_match_homopolymers_literals(oracle, k):
    reads, refs = MATCH_CASES[k]
    for rd, rf in zip(reads, refs):
        got = oracle.match_homopolymers([rf], [rd])
        pos, rlen = matchcheck(rf, rd)
        assert np.asarray(got[0]).tolist() == [0] * len(pos)
        assert np.asarray(got[1]).tolist() == pos and np.asarray(got[2]).tolist() == rlen, (rd, rf)


@pytest.mark.parametrize("k", range(len(ERROR_CASES)))
def test_find_errors_literals(oracle, k):
    reads, refs = ERROR_CASES[k]
    for rd, rf in zip(reads, refs):
        got = oracle.find_errors([rf], [rd])
        want = checkfun([rf], [rd])
        assert got[0] == want[0]
        for a, b in zip(got[1:6], want[1:6]):
            assert np.asarray(a).tolist() == b, (rd, rf)
        ins = {}
        for p, l in zip(np.asarray(got[6]).tolist(), np.asarray(got[7]).tolist()):
            ins.setdefault(p, []).append(l)
        assert {p: sorted(v) for p, v in ins.items()} == want[6], (rd, rf)


def test_find_errors_many_alignments_and_errors(oracle):
    ref = "GGAAAC-GATCAGCTACGAACACT"
    refs = [ref, ref.replace("-", ""), "GG-AAACGATCAGCTACGAACACT--"]
    reads = ["GGAAACTGATCAGCTACGAACACT", "GGAAACGATCAGCTACGAACAC-", "GGTAAACGA-CAGCTACGAACACTAA"]
    got = oracle.find_errors(refs, reads)
    want = checkfun(refs, reads)
    assert got[0] == want[0] and [np.asarray(x).tolist() for x in got[1:6]] == list(want[1:6])
    with pytest.raises(oracle.OracleError, match="lengths of alignment vectors should match up"):
        oracle.find_errors(refs, reads[:2])
    with pytest.raises(oracle.OracleError, match="equal length"):
        oracle.find_errors(["ACGT"], ["ACG"])
    with pytest.raises(oracle.OracleError, match="same for all alignments"):
        oracle.find_errors(["ACGT", "ACGTA"], ["ACGT", "ACGTA"])
    with pytest.raises(oracle.OracleError, match="unknown character 'N'"):
        oracle.find_errors(["ACGT"], ["ACNT"])
    with pytest.raises(oracle.OracleError, match="lengths of alignment vectors should match up"):
        oracle.match_homopolymers(["AACC"], [])
    with pytest.raises(oracle.OracleError, match="equal length"):
        oracle.match_homopolymers(["AACC"], ["AAC"])
