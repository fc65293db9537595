"""Pins the alignment oracle (oracle/align.c) without a GPU.

The reference cannot be built here, so the oracle is pinned the way the reference's
own tests pin the C++ (tests/testthat/test-adaptor-align.R, test-general-align.R):
  * literal known answers those tests assert (-21 for an empty read, zeros for an
    empty adaptor), plus the reference outputs recorded in SURVEY.md section 8c;
  * an independent textbook affine-gap DP (three-matrix Gotoh, written here in
    Python -- the stand-in for Biostrings::pairwiseAlignment used by the R tests):
    optimal scores must agree, positions must agree on the hand-made reads;
  * the STRIPPER invariants for alignment strings.
"""
import math

import numpy as np
import pytest

ADAPTOR = "AAAAGGGGCCCCTTTT"
READS = [r.upper() for r in [
    "AAAAGGGGCCCCTTTT", "acgtacgtacgtAAAAGGGGCCCCTTTT", "AAAAGGGGCCCCTTTTacgtacgtacgt",
    "GGGGCCCCTTTT", "AAAAGGGGCCCC", "acgtacgtacgtAAAAGGGGCCCCTTTTacgtacgtacgt",
    "acgtacgtacgtAAAAGGGGCCCC", "GGGGCCCCTTTTacgtacgtacgt", "GGGGCCCC",
    "AAAAGGGGacgtCCCCTTTT", "AAAAGGCCTTTT"]]
# positions of the read that pairwiseAlignment(type="local-global") reports for these
# hand-made cases (test-adaptor-align.R:40-42 asserts identity with them)
EXPECT_POS = [(1, 16), (13, 28), (1, 16), (1, 12), (1, 12), (13, 28), (13, 24), (1, 12), (1, 8), (1, 20), (1, 12)]


def sub_score(r, obs, err):
    """log2 odds of the reference's scoring scheme (src/reference_align.cpp:15-47)."""
    deg = {"A": 1, "C": 1, "G": 1, "T": 1, "M": 2, "R": 2, "W": 2, "S": 2, "Y": 2, "K": 2,
           "V": 3, "H": 3, "D": 3, "B": 3, "N": 4}[r]
    g = 1.0 / deg
    if deg == 1:
        matched = r == obs
    elif deg == 2:
        matched = False      # quirk Q2: decided on the reference character alone
    else:
        matched = True
    if not matched:
        g = 1 - g
    v = g * (1 - err) * 4 + (1 - g) * err * (4 / 3)
    return math.log(v) / math.log(2) if v > 0 else -math.inf


def gotoh(ref, read, errs, go, ge, local):
    """Textbook three-state affine DP; gap of length k costs go + k*ge.
    local: free leading read bases, free trailing read bases (adaptor fully aligned)."""
    R, L = len(ref), len(read)
    NEG = -math.inf
    M = [[NEG] * (R + 1) for _ in range(L + 1)]   # best ending in any state
    X = [[NEG] * (R + 1) for _ in range(L + 1)]   # gap in reference (consumes read): vertical
    Y = [[NEG] * (R + 1) for _ in range(L + 1)]   # gap in read (consumes reference): horizontal
    M[0][0] = 0.0
    for i in range(1, L + 1):
        if local:
            M[i][0] = 0.0
        else:
            X[i][0] = -(go + ge * i)
            M[i][0] = X[i][0]
    for c in range(1, R + 1):
        Y[0][c] = -(go + ge * c)
        M[0][c] = Y[0][c]
    for i in range(1, L + 1):
        for c in range(1, R + 1):
            free_v = local and c == R
            X[i][c] = max(M[i - 1][c] - (0 if free_v else go + ge), X[i - 1][c] - (0 if free_v else ge))
            Y[i][c] = max(M[i][c - 1] - (go + ge), Y[i][c - 1] - ge)
            d = M[i - 1][c - 1] + sub_score(ref[c - 1], read[i - 1], errs[i - 1])
            M[i][c] = max(d, X[i][c], Y[i][c])
    return M[L][R]


def errs_of(qual):
    return [10 ** (-(ord(c) - 33) / 10) for c in qual]


def test_known_answers(oracle, oenc):
    # tests/testthat/test-adaptor-align.R:48-56
    out = oracle.adaptor_align(READS, ["5" * len(r) for r in READS], oenc, 5, 1, "")
    assert not out[0].any() and not out[1].any() and not out[2].any()
    out = oracle.adaptor_align([""] * 3, [""] * 3, oenc, 5, 1, ADAPTOR)
    assert out[0].tolist() == [-(len(ADAPTOR) + 5.0)] * 3
    assert not out[1].any() and not out[2].any()
    # output of the reference itself for 12-nt flank + adaptor at Q20
    seq = "ACGTACGTACGT" + ADAPTOR
    out = oracle.adaptor_align([seq], ["5" * len(seq)], oenc, 5, 1, ADAPTOR, [4], [8])
    assert out[0][0] == 31.768006884878162
    assert (out[1][0], out[2][0], out[3][0][0], out[4][0][0]) == (13, 28, 17, 4)


def test_literal_reads_positions_and_scores(oracle, oenc):
    quals = ["5" * len(r) for r in READS]
    out = oracle.adaptor_align(READS, quals, oenc, 5, 1, ADAPTOR)
    for k, (r, q) in enumerate(zip(READS, quals)):
        assert abs(out[0][k] - gotoh(ADAPTOR, r, errs_of(q), 5, 1, True)) < 1e-9
        assert (out[1][k], out[2][k]) == EXPECT_POS[k], (k, r)
    # full-adaptor section with gaps returns the whole read (test-adaptor-align.R:120-121)
    out = oracle.adaptor_align(READS, quals, oenc, 5, 1, ADAPTOR, [0], [len(ADAPTOR)])
    assert out[3][0].tolist() == [1] * len(READS)
    assert out[4][0].tolist() == [len(r) for r in READS]


def test_affine_gap_traps(oracle, oenc):
    # tests/testthat/test-adaptor-align.R:59-85: one mismatch is cheaper than a gap,
    # several mismatches are dearer than one long gap
    for a1, read in (("AAACCCAAATTTAAA", "AAAAAAAAA"), ("AAAAAA", "AAACCCAAA")):
        q = "+" * len(read)
        out = oracle.adaptor_align([read], [q], oenc, 5, 1, a1)
        assert abs(out[0][0] - gotoh(a1, read, errs_of(q), 5, 1, True)) < 1e-9
        assert out[1][0] == 1 and out[2][0] == len(read)


@pytest.mark.parametrize("seed", range(6))
def test_scores_equal_textbook_dp(oracle, oenc, seed):
    rng = np.random.default_rng(seed)
    alphabet = list("ACGTMRWSYKVHDBN") if seed % 2 else list("ACGT")
    for _ in range(25):
        R = int(rng.integers(0, 25))
        L = int(rng.integers(0, 40))
        ref = "".join(rng.choice(alphabet, R))
        read = "".join(rng.choice(list("ACGT"), L))
        qual = "".join(chr(int(x)) for x in rng.integers(34, 127, L))
        go, ge = ((5, 1), (20, 1), (1, 1), (2.5, 0.75))[int(rng.integers(0, 4))]
        s_loc = oracle.adaptor_align([read], [qual], oenc, go, ge, ref)[0][0]
        s_glo = oracle.barcode_align([read], [qual], oenc, go, ge, ref)[0]
        if R == 0:
            assert s_loc == 0.0
            continue
        assert abs(s_loc - gotoh(ref, read, errs_of(qual), go, ge, True)) < 1e-8
        assert abs(s_glo - gotoh(ref, read, errs_of(qual), go, ge, False)) < 1e-8


def test_sections_match_alignment_strings(oracle, oenc):
    # section extraction must agree with the columns of the gapped strings (same idea as
    # test-adaptor-align.R:87-118, with general_align's strings standing in for alignedPattern)
    rng = np.random.default_rng(11)
    for _ in range(40):
        R = int(rng.integers(4, 20))
        ref = "".join(rng.choice(list("ACGT"), R))
        read = "".join(rng.choice(list("ACGT"), int(rng.integers(R - 3, R + 4))))
        qual = "".join(chr(int(x)) for x in rng.integers(40, 100, len(read)))
        _, _, rs, qs = oracle.general_align([read], [qual], oenc, 5, 1, ref)
        rs, qs = rs[0], qs[0]
        assert rs.replace("-", "") == ref and qs.replace("-", "") == read


def test_general_align_invariants(oracle, oenc):
    # STRIPPER (test-general-align.R:17-53) + optimal score
    rng = np.random.default_rng(32000)
    for go in (5, 20, 1):
        for _ in range(15):
            ref = "".join(rng.choice(list("ACGT"), 50))
            read = "".join(rng.choice(list("ACGT"), int(rng.integers(20, 81))))
            errs = 10 ** -rng.uniform(1, 5, len(read))
            qual = "".join(chr(33 + int(round(-10 * math.log10(e)))) for e in errs)
            sc, ed, rs, qs = oracle.general_align([read], [qual], oenc, go, 1, ref)
            assert abs(sc[0] - gotoh(ref, read, errs_of(qual), go, 1, False)) < 1e-8
            assert len(rs[0]) == len(qs[0])
            assert rs[0].replace("-", "") == ref and qs[0].replace("-", "") == read
            assert ed[0] == sum(a != b for a, b in zip(rs[0], qs[0]))
            assert not any(a == "-" and b == "-" for a, b in zip(rs[0], qs[0]))
    # edit_only returns the same numbers
    a = oracle.general_align([read], [qual], oenc, 5, 1, ref)
    b = oracle.general_align([read], [qual], oenc, 5, 1, ref, edit_only=True)
    assert a[0][0] == b[0][0] and a[1][0] == b[1][0] and b[2] == []


def test_cost_table_quirks(oracle):
    # SURVEY App.B Q2/Q21: N column scores ~0; Q0 base can never match
    err, _ = oracle.phred_encoding()
    m, mm = oracle.cost_tables(err)
    assert np.all(np.abs(m[3]) < 1e-12)
    assert m[0][0] == -np.inf and abs(mm[0][0] - math.log2(4 / 3)) < 1e-12
    assert np.allclose(m[1], mm[1])          # 2-fold is symmetric
    assert abs(m[0][20] - math.log2(4 * (1 - 0.01))) < 1e-12


def test_errors(oracle, oenc):
    with pytest.raises(oracle.OracleError, match="same length"):
        oracle.adaptor_align(["ACGT"], ["III"], oenc, 5, 1, ADAPTOR)
    with pytest.raises(oracle.OracleError, match="unrecognized base"):
        oracle.adaptor_align(["ACGT"], ["IIII"], oenc, 5, 1, "ACXT")
    with pytest.raises(oracle.OracleError, match="quality cannot be lower"):
        oracle.adaptor_align(["ACGT"], ["II I"], oenc, 5, 1, ADAPTOR)
    with pytest.raises(oracle.OracleError, match="error probabilities should decrease"):
        oracle.adaptor_align(["ACGT"], ["!!!!"], (np.array([0.1, 0.2]), b"!\""), 5, 1, ADAPTOR)
    # Biostrings' PhredQuality runs to Q 99 = byte 132; the reference's check compares a (signed) char with an int
    # (src/quality_encoding.cpp:21), so a table that runs past byte 127 is rejected by the reference itself on x86
    full = (np.power(10.0, -np.arange(100) / 10.0), bytes(range(33, 133)))
    with pytest.raises(oracle.OracleError, match="should increase consecutively"):
        oracle.adaptor_align(["ACGT"], ["IIII"], full, 5, 1, ADAPTOR)
    printable = (np.power(10.0, -np.arange(95) / 10.0), bytes(range(33, 128)))   # up to byte 127: accepted
    oracle.adaptor_align(["ACGT"], ["IIII"], printable, 5, 1, ADAPTOR)


def test_mask_bad_bases(oracle, oenc):
    # tests/testthat/test-masking.R:4-42: base -> N where error > threshold (strict)
    seqs = ["AAAATTTTCCCCGGGG", "GGGGTTTTCCCCAAAA", "AAAACCCCTTTTGGGG"]
    quals = ["".join(chr(33 + q) * 4 for q in qs) for qs in ((10, 20, 30, 40), (43, 33, 23, 13), (27, 7, 17, 37))]
    err, _ = oenc
    for thr in (0.001, 0.01, 0.05, 0.1):
        out = oracle.mask_bad_bases(seqs, quals, oenc, thr)
        for s, q, o in zip(seqs, quals, out):
            want = "".join("N" if err[ord(c) - 33] > thr else b for b, c in zip(s, q))
            assert o == want


# the literals of tests/testthat/test-masking.R:45-100: upper case = masked position
UNMASK_CASES = [
    ["acacgtagtgtcagtctaacatcagctacgttacat", "ACACGTAGTGTCAGTCTAACATCAGCTACGTTACAT", "acacgtcgtgtcagtctaaCatcagctacgttacat",
     "Acacgttgtgtcagtctaacatcagctacgttacat", "acacgtggtgtcagtctaacatcagctacgttacaT"],
    ["acacgtagtgtcagtc-taacatcagctacgttacat", "ACACGTAGTGTCAGTC-TAACATCAGCTACGTTACAT", "acacgtcgtgtcagtc-taaCatcagctacgttacat",
     "Acacgttgtgtcagtc-taacatcagctacgttacat", "acacgtggtgtcagtc-taacatcagctacgttacaT"],
    ["-acacgtcgtgtcagtctaacatcagctacgttacat", "-ACACGTAGTGTCAGTCTAACATCAGCTACGTTACAT", "-acacgtcgtgtcagtctaaCatcagctacgttacat",
     "-Acacgtcgtgtcagtctaacatcagctacgttacat", "-acacgtcgtgtcagtctaacatcagctacgttacaT"],
    ["acacgtcgtgtcagtctaacatcagctacgttacat-", "ACACGTAGTGTCAGTCTAACATCAGCTACGTTACAT-", "acacgtcgtgtcagtctaaCatcagctacgttacat-",
     "Acacgtcgtgtcagtctaacatcagctacgttacat-", "acacgtcgtgtcagtctaacatcagctacgttacaT-"],
]


def unmask_inputs(seq_in):
    """MASKFUN / ORIGINAL of the reference's test (DNAStringSet upper-cases what is left)."""
    import re
    masked = [re.sub("[ACTG]", "N", s).upper() for s in seq_in]
    original = [s.replace("-", "").upper() for s in seq_in]
    return masked, original


@pytest.mark.parametrize("case", range(len(UNMASK_CASES)))
def test_unmask_alignment_reference_literals(oracle, case):
    seq_in = UNMASK_CASES[case]
    masked, original = unmask_inputs(seq_in)
    assert oracle.unmask_alignment(masked, original) == [s.upper() for s in seq_in]


def test_unmask_alignment_errors(oracle):
    # tests/testthat/test-masking.R:93-98
    with pytest.raises(oracle.OracleError, match="different lengths"):
        oracle.unmask_alignment(["AA-AA"], ["AAA"])
    with pytest.raises(oracle.OracleError, match="sequence in alignment string is longer than the original"):
        oracle.unmask_alignment(["NNNNN"], ["AAA"])
    with pytest.raises(oracle.OracleError, match="alignment and original sequences should have the same number of entries"):
        oracle.unmask_alignment(["AAAA", "GGGG"], ["AAA"])
    assert oracle.unmask_alignment(["AA-A"], ["AAA"]) == ["AA-A"]
    assert oracle.unmask_alignment([], []) == []
