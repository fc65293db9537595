"""GPU parity: HIP consensus kernels vs the CPU oracle, through the C ABI.
Mirrors /root/reference/tests/testthat/test-consensus.R (same literal alignments,
coverage / pseudo-count grid, Ns, empty input, error messages) plus randomized
MSA-shaped batches.  Consensus and Phred strings must be identical; log-errors
within 1e-11 relative (device vs host libm in the last step only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TEST_ALIGN = ["AAAAGAAAAA-AAATAAAA", "ACACA-AAAA--AAT-AGA", "GA-AG-C-A-T-AAT-AAA",
              "AT-AG-T-AGTAAGA-AGA", "-AAAGAT-AGTCAGA-AGA", "AGAAAAT-AGAAATA-AGA"]
N_ALIGN = ["NAAAAANNN", "NNAANA---", "NNNANNN--", "NNNN--NN-"]


def rand_alignment(rng, nrows, width, gap=0.1, nfrac=0.02):
    truth = rng.choice(list("ACGT"), width)
    rows, quals = [], []
    for _ in range(nrows):
        r = truth.copy()
        sub = rng.random(width) < 0.08
        r[sub] = rng.choice(list("ACGT"), int(sub.sum()))
        r[rng.random(width) < nfrac] = "N"
        r[rng.random(width) < gap] = "-"
        row = "".join(r)
        rows.append(row)
        n = len(row.replace("-", ""))
        quals.append("".join(chr(int(x)) for x in rng.integers(33, 127, n)))
    return rows, quals


def ungapped_quals(aln, ch="5"):
    return [ch * len(r.replace("-", "")) for r in aln]


@pytest.mark.parametrize("aln", [TEST_ALIGN, N_ALIGN])
@pytest.mark.parametrize("cov,pc", [(0.6, 1), (0.6, 2), (0.2, 2), (0.9, 1), (0.0, 0.5), (1.0, 1)])
def test_basic_literals(oracle, aln, cov, pc):
    from sarlacc_amd import calls
    want = oracle.create_consensus_basic(aln, cov, pc)
    got = calls.create_consensus_basic(aln, cov, pc)
    assert got[0] == want[0]
    assert np.allclose(got[1], want[1], rtol=1e-11, atol=0)
    wl = oracle.create_consensus_basic_loop([aln, TEST_ALIGN], cov, pc)
    gl = calls.create_consensus_basic_loop([aln, TEST_ALIGN], cov, pc)
    assert gl[0] == wl[0] and gl[1] == wl[1]


def test_basic_known_answer_and_empty(oracle):
    from sarlacc_amd import SarlaccError, calls
    assert calls.create_consensus_basic(TEST_ALIGN, 0.6, 1)[0] == "AAAAGTAGTAAAAAGA"
    got = calls.create_consensus_basic([], 0.6, 1)
    assert got[0] == "" and got[1].size == 0
    assert calls.create_consensus_basic_loop([], 0.6, 1) == [[], []]
    assert calls.create_consensus_basic_loop([[], TEST_ALIGN, []], 0.6, 1)[0] == ["", "AAAAGTAGTAAAAAGA", ""]
    with pytest.raises(SarlaccError, match="alignment strings should have the same length"):
        calls.create_consensus_basic(["AAA", "AA"], 0.6, 1)
    with pytest.raises(SarlaccError, match="unknown character 'X' in alignment string"):
        calls.create_consensus_basic(["AXA", "AAA"], 0.6, 1)
    with pytest.raises(SarlaccError, match="minimum coverage should be a numeric scalar"):
        calls.create_consensus_basic(["AAA"], [0.6, 0.7], 1)


@pytest.mark.parametrize("aln", [TEST_ALIGN, N_ALIGN])
def test_quality_literals(oracle, oenc, enc, aln):
    from sarlacc_amd import calls
    rng = np.random.default_rng(len(aln))
    for lo, hi in ((53, 73), (40, 60), (33, 50), (33, 126)):
        quals = ["".join(chr(int(x)) for x in rng.integers(lo, hi + 1, len(r.replace("-", "")))) for r in aln]
        for cov in (0.6, 0.2, 0.9):
            want = oracle.create_consensus_quality(aln, cov, quals, oenc)
            got = calls.create_consensus_quality(aln, cov, quals, enc)
            assert got[0] == want[0]
            assert np.allclose(got[1], want[1], rtol=1e-11, atol=1e-300)
    alns = [TEST_ALIGN, N_ALIGN]
    qs = [ungapped_quals(a) for a in alns]
    wl = oracle.create_consensus_quality_loop(alns, 0.6, qs, oenc)
    gl = calls.create_consensus_quality_loop(alns, 0.6, qs, enc)
    assert gl[0] == wl[0] and gl[1] == wl[1]


def test_quality_errors(enc):
    from sarlacc_amd import SarlaccError, calls
    got = calls.create_consensus_quality([], 0.6, [], enc)
    assert got[0] == "" and got[1].size == 0
    with pytest.raises(SarlaccError, match="different numbers of entries"):
        calls.create_consensus_quality(N_ALIGN[:1], 0.6, [], enc)
    with pytest.raises(SarlaccError, match="quality vector is shorter than the alignment sequence"):
        calls.create_consensus_quality(N_ALIGN[:1], 0.6, ["I"], enc)
    with pytest.raises(SarlaccError, match="quality vector is longer than the alignment sequence"):
        calls.create_consensus_quality(N_ALIGN[:1], 0.6, ["I" * 1000], enc)
    with pytest.raises(SarlaccError, match="quality cannot be lower than smallest encoded value"):
        calls.create_consensus_quality(["ACGT"], 0.6, ["II I"], enc)
    with pytest.raises(SarlaccError, match="alignment strings should have the same length"):
        calls.create_consensus_quality(["ACGT", "ACG"], 0.6, ["IIII", "III"], enc)


@pytest.mark.parametrize("seed,ngroups,nrows,width", [(1, 40, 10, 300), (2, 7, 3, 2200), (3, 100, 6, 63),
                                                       (4, 5, 50, 129), (5, 3, 70, 500), (6, 30, 1, 64)])
def test_random_batches(oracle, oenc, enc, seed, ngroups, nrows, width):
    from sarlacc_amd import calls
    rng = np.random.default_rng(seed)
    alns, quals = [], []
    for g in range(ngroups):
        a, q = rand_alignment(rng, int(rng.integers(1, nrows + 1)), int(rng.integers(1, width + 1)))
        alns.append(a)
        quals.append(q)
    alns[len(alns) // 2] = []
    quals[len(quals) // 2] = []
    for cov in (0.6, 0.35):
        w = oracle.create_consensus_quality_loop(alns, cov, quals, oenc)
        g = calls.create_consensus_quality_loop(alns, cov, quals, enc)
        assert g[0] == w[0], "quality consensus strings differ"
        assert g[1] == w[1], "quality phred strings differ"
        w = oracle.create_consensus_basic_loop(alns, cov, 1.0)
        g = calls.create_consensus_basic_loop(alns, cov, 1.0)
        assert g[0] == w[0] and g[1] == w[1]


def test_consensus_of_a_very_deep_alignment(oracle, oenc, enc):
    """3 000 rows: the per-row offsets of the one-column kernel need more than 64 KB of LDS."""
    from sarlacc_amd import calls
    rng = np.random.default_rng(17)
    truth = rng.choice(list("ACGT"), 90)
    rows, quals = [], []
    for _ in range(3000):
        r = truth.copy()
        sub = rng.random(90) < 0.1
        r[sub] = rng.choice(list("ACGT"), int(sub.sum()))
        r[rng.random(90) < 0.05] = "-"
        row = "".join(r)
        rows.append(row)
        quals.append("".join(chr(int(c)) for c in rng.integers(40, 80, len(row.replace("-", "")))))
    got = calls.create_consensus_quality_loop([rows], 0.6, [quals], enc)
    want = oracle.create_consensus_quality_loop([rows], 0.6, [quals], oenc)
    assert got[0] == list(want[0]) and got[1] == list(want[1])
    gb = calls.create_consensus_basic_loop([rows], 0.6, 1.0)
    wb = oracle.create_consensus_basic_loop([rows], 0.6, 1.0)
    assert gb[0] == list(wb[0]) and gb[1] == list(wb[1])
