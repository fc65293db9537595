"""GPU parity: HIP quality-weighted DP vs the CPU oracle, through the C ABI.

Mirrors /root/reference/tests/testthat/test-adaptor-align.R and
test-general-align.R: same literal reads/adaptors, same edge cases (empty adaptor,
empty read, affine-gap traps, section extraction), plus randomized differential
batches.  Scores must be bit-identical (fp64, same operation order); positions,
sections, edit distances and alignment strings identical.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ADAPTOR = "AAAAGGGGCCCCTTTT"
READS = ["AAAAGGGGCCCCTTTT", "acgtacgtacgtAAAAGGGGCCCCTTTT", "AAAAGGGGCCCCTTTTacgtacgtacgt",
         "GGGGCCCCTTTT", "AAAAGGGGCCCC", "acgtacgtacgtAAAAGGGGCCCCTTTTacgtacgtacgt",
         "acgtacgtacgtAAAAGGGGCCCC", "GGGGCCCCTTTTacgtacgtacgt", "GGGGCCCC",
         "AAAAGGGGacgtCCCCTTTT", "AAAAGGCCTTTT"]  # test-adaptor-align.R:5-19 (DNAStringSet upper-cases)
READS = [r.upper() for r in READS]
IUPAC = "ACGTMRWSYKVHDBN"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def rand_quals(reads, seed, lo=33, hi=126):
    rng = np.random.default_rng(seed)
    return [rng.integers(lo, hi + 1, len(r)).astype(np.uint8).tobytes().decode() for r in reads]


def compare_adaptor(oracle, oenc, enc, reads, quals, adaptor, go, ge, ss=(), se=()):
    from sarlacc_amd import calls
    ref = oracle.adaptor_align(reads, quals, oenc, go, ge, adaptor, ss, se)
    out = calls.adaptor_align(reads, quals, enc, go, ge, adaptor, ss, se)
    assert np.array_equal(bits(ref[0]), bits(out[0])), "scores differ"
    assert np.array_equal(ref[1], out[1]), "starts differ"
    assert np.array_equal(ref[2], out[2]), "ends differ"
    assert len(ref[3]) == len(out[3])
    for a, b in zip(ref[3], out[3]):
        assert np.array_equal(a, b), "section starts differ"
    for a, b in zip(ref[4], out[4]):
        assert np.array_equal(a, b), "section widths differ"
    so = calls.adaptor_align_score_only(reads, quals, enc, go, ge, adaptor)
    assert np.array_equal(bits(ref[0]), bits(so))
    return out


def test_wave_shift_selftest(enc):
    # known answer from the reference's own test (test-adaptor-align.R:53-56):
    # empty read vs 16-mer, go=5, ge=1 -> -(16+5)
    from sarlacc_amd import calls
    out = calls.adaptor_align(["", ""], ["", ""], enc, 5, 1, ADAPTOR, [], [])
    assert out[0].tolist() == [-21.0, -21.0]
    assert out[1].tolist() == [0, 0] and out[2].tolist() == [0, 0]


def test_reference_literals(oracle, oenc, enc):
    quals = rand_quals(READS, 141000)
    out = compare_adaptor(oracle, oenc, enc, READS, quals, ADAPTOR, 5, 1)
    assert len(out[0]) == len(READS)
    # clean copies of the adaptor must be found where they were planted
    q20 = [chr(53) * len(r) for r in READS]
    out = compare_adaptor(oracle, oenc, enc, READS, q20, ADAPTOR, 5, 1, [4], [8])
    assert out[1][0] == 1 and out[2][0] == 16
    assert out[1][1] == 13 and out[2][1] == 28
    assert abs(out[0][1] - 31.768006884878162) < 1e-12  # SURVEY section 8c recorded reference output
    assert out[3][0][1] == 17 and out[4][0][1] == 4


def test_empty_adaptor_and_reads(oracle, oenc, enc):
    quals = rand_quals(READS, 7)
    out = compare_adaptor(oracle, oenc, enc, READS, quals, "", 5, 1)
    assert not out[0].any() and not out[1].any() and not out[2].any()
    compare_adaptor(oracle, oenc, enc, ["", "ACGT", ""], ["", "IIII", ""], ADAPTOR, 5, 1, [0, 3], [16, 9])
    from sarlacc_amd import calls
    out = calls.adaptor_align([], [], enc, 5, 1, ADAPTOR, [], [])
    assert len(out[0]) == 0


def test_affine_gap_traps(oracle, oenc, enc):
    # test-adaptor-align.R:59-85
    compare_adaptor(oracle, oenc, enc, ["AAAAAAAAA"], ["+" * 9], "AAACCCAAATTTAAA", 5, 1)
    compare_adaptor(oracle, oenc, enc, ["AAACCCAAA"], ["+" * 9], "AAAAAA", 5, 1)


def test_section_extraction(oracle, oenc, enc):
    # every internal (start,end) pair, like combn(nchar(adaptor)-1, 2) in the reference test
    R = len(ADAPTOR)
    ss, se = [], []
    for a in range(1, R):
        for b in range(a + 1, R):
            ss.append(a)
            se.append(b)
    quals = rand_quals(READS, 3)
    compare_adaptor(oracle, oenc, enc, READS, quals, ADAPTOR, 5, 1, ss, se)
    out = compare_adaptor(oracle, oenc, enc, READS, quals, ADAPTOR, 5, 1, [0], [R])
    assert out[3][0].tolist() == [1] * len(READS)
    assert out[4][0].tolist() == [len(r) for r in READS]


@pytest.mark.parametrize("R", [1, 2, 3, 7, 15, 16, 17, 22, 30, 31, 32, 33, 50, 64, 65, 70, 100, 128, 129, 200, 300, 600])
def test_random_differential(oracle, oenc, enc, R):
    rng = np.random.default_rng(1000 + R)
    from sarlacc_amd.mock import random_reads
    adaptor = "".join(rng.choice(list(IUPAC if R % 2 else "ACGT"), R))
    nreads = 37 if R < 100 else 9
    reads, quals = random_reads(nreads, 0, 150 if R < 100 else 80, seed=R)
    # plant noisy copies so that real alignments exist
    core = "".join(c if c in "ACGT" else "A" for c in adaptor)
    reads[1] = reads[1][:20] + core + reads[1][20:]
    quals[1] = rand_quals([reads[1]], R)[0]
    go, ge = ((5, 1), (2.5, 0.75), (1, 1), (20, 1))[R % 4]
    nsec = min(R, 3)
    ss = sorted(rng.integers(0, R, nsec).tolist())
    se = [int(rng.integers(s, R) + 1) for s in ss]
    compare_adaptor(oracle, oenc, enc, reads, quals, adaptor, go, ge, ss, se)


@pytest.mark.parametrize("R", [3, 16, 17, 30, 32])
def test_row16_shapes_without_interleaved_alignments(oracle, oenc, enc, R):
    # references of up to 32 columns run eight 8-lane alignments per wavefront (two interleaved per DPP row); the shapes with
    # four 16-lane alignments they replaced stay reachable (align_interleave = -1) and give the same bits
    from sarlacc_amd import calls
    from sarlacc_amd.mock import random_reads
    rng = np.random.default_rng(77 + R)
    adaptor = "".join(rng.choice(list("ACGTN"), R))
    reads, quals = random_reads(41, 0, 200, seed=R + 5)
    core = "".join(c if c in "ACGT" else "G" for c in adaptor)
    for k in (1, 7, 20):
        reads[k] = reads[k][:30] + core + reads[k][30:]
        quals[k] = rand_quals([reads[k]], R + k)[0]
    try:
        calls.set_option("align_interleave", -1)
        compare_adaptor(oracle, oenc, enc, reads, quals, adaptor, 5, 1, [0, R // 2], [R // 2 + 1, R])
    finally:
        calls.set_option("align_interleave", 0)
    compare_adaptor(oracle, oenc, enc, reads, quals, adaptor, 5, 1, [0, R // 2], [R // 2 + 1, R])


def _penalty_batch(oracle, oenc, enc, R, go, ge, seed):
    from sarlacc_amd import calls
    from sarlacc_amd.mock import random_reads
    rng = np.random.default_rng(seed)
    adaptor = "".join(rng.choice(list(IUPAC if R % 2 else "ACGT"), R))
    reads, quals = random_reads(21, 0, 140, seed=seed)
    core = "".join(c if c in "ACGT" else "C" for c in adaptor)
    reads[2] = reads[2][:10] + core[: R // 2] + "GG" + core[R // 2:] + reads[2][10:]   # insertion in the read
    reads[4] = reads[4][:15] + core[: R // 3] + core[R // 3 + 2:] + reads[4][15:]      # deletion from the read
    quals[2] = rand_quals([reads[2]], seed)[0]
    quals[4] = rand_quals([reads[4]], seed + 1)[0]
    ss, se = ([R // 3], [max(R // 3, (2 * R) // 3)]) if R > 2 else ([], [])
    compare_adaptor(oracle, oenc, enc, reads, quals, adaptor, go, ge, ss, se)
    a = oracle.general_align(reads[:8], quals[:8], oenc, go, ge, core)
    b = calls.general_align(reads[:8], quals[:8], enc, go, ge, core, False)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert np.array_equal(bits(oracle.barcode_align(reads, quals, oenc, go, ge, core)),
                          bits(calls.barcode_align(reads, quals, enc, go, ge, core)))


@pytest.mark.parametrize("R", [5, 30, 33, 70, 129, 600])
@pytest.mark.parametrize("go,ge", [(0, 1), (0, 0.3), (-1, 2), (-0.25, 0.25), (0.1, 0.7), (3, 0)])
def test_gap_penalty_variants(oracle, oenc, enc, R, go, ge):
    """gapopen == 0 (open == extend), gapopen < 0 (the kernel variant that selects penalties
    explicitly), penalties with non-zero low mantissa bits, free extension."""
    _penalty_batch(oracle, oenc, enc, R, go, ge, seed=7000 + R)


@pytest.mark.parametrize("R", [3, 16, 30, 31, 64, 100, 300])
def test_forced_penalty_select_kernel(oracle, oenc, enc, R):
    """Both kernel variants must agree with the oracle for ordinary penalties."""
    from sarlacc_amd import calls
    calls.set_option("align_pensel", 1)
    try:
        _penalty_batch(oracle, oenc, enc, R, 5, 1, seed=8000 + R)
        _penalty_batch(oracle, oenc, enc, R, 2.5, 0.75, seed=8100 + R)
    finally:
        calls.set_option("align_pensel", 0)


def test_c2_shape_sample(oracle, oenc, enc):
    # BASELINE config 2 shape on a sample the oracle finishes in seconds:
    # 2 kb mock reads vs a 30-bp adaptor (9 fixed + 12 N + 9 fixed), go=5, ge=1
    from sarlacc_amd.mock import mock_reads
    a1 = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"
    a2 = "CACACTGAGCAGCGACTAGACA"
    sim = mock_reads(a1, a2, nmolecules=12, nreads=10, seqlen=2000 - 52, seed=1000)
    reads, quals = sim["reads"].to_strings(), sim["quals"].to_strings()
    out = compare_adaptor(oracle, oenc, enc, reads, quals, a1, 5, 1, [9], [21])
    fwd = ~sim["flipped"]
    assert (out[0][fwd] > 15).mean() > 0.9  # the planted adaptor is found on forward reads
    assert (out[1][fwd] <= 3).mean() > 0.8


def test_ragged_and_long(oracle, oenc, enc):
    from sarlacc_amd.mock import random_reads
    reads, quals = random_reads(23, 0, 3000, seed=5)
    reads[3] = ""
    quals[3] = ""
    compare_adaptor(oracle, oenc, enc, reads, quals, "ACGTNNNNACGTRYACGT", 5, 1, [4], [8])


def test_adaptor_anywhere_in_long_reads(oracle, oenc, enc):
    """The traceback of adaptor_align restores a snapshot of the lane state above the landing row and recomputes a window; the
    snapshots are dense near the read start and sparse beyond (align.hip, SNAP_DENSE / SNAP_SPARSE).  Adaptors planted so
    that the alignment lands around every kind of boundary -- inside the dense part, on and just past its end, around the
    sparse snapshots, at the very end of the read -- exact or with substitutions and indels, and once with a long stretch
    of read inside the adaptor (the path leaves its window through the top and a second window is recomputed)."""
    rng = np.random.default_rng(31)
    nuc = np.array(list("ACGT"))
    adaptor = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"
    filled = "ACGATCAGC" + "ACGTTGCAAGTC" + "GTCAGTCAG"
    reads = []
    ends = [30, 64, 100, 192, 250, 256, 257, 290, 297, 298, 300, 330, 511, 512, 553, 554, 600, 767, 768, 810, 1024, 1065, 1500, 2000]
    for e in ends:
        for variant in range(3):
            L = 2000 if e <= 1990 else e
            body = "".join(nuc[rng.integers(0, 4, L)])
            a = filled
            if variant == 1:   # a substitution and a deletion
                a = a[:5] + "T" + a[6:14] + a[15:]
            if variant == 2:   # an insertion in the N stretch
                a = a[:12] + "GGG" + a[12:]
            start = max(e - len(a), 0)
            reads.append(body[:start] + a + body[start + len(a):])
    # 40 read bases inside the adaptor, landing row just past a sparse snapshot + head: the path runs 70 rows up
    body = "".join(nuc[rng.integers(0, 4, 2000)])
    reads.append(body[:500] + filled[:15] + "".join(nuc[rng.integers(0, 4, 40)]) + filled[15:] + body[570:])
    reads.append(body[:230] + filled[:15] + "".join(nuc[rng.integers(0, 4, 40)]) + filled[15:] + body[300:])
    quals = rand_quals(reads, 8, lo=40, hi=75)
    out = compare_adaptor(oracle, oenc, enc, reads, quals, adaptor, 5, 1, [9], [21])
    assert (out[0][:len(ends) * 3] > 10).all()    # every planted adaptor is what the alignment found


def test_int32_directions_for_very_long_reads(oracle, oenc, enc):
    from sarlacc_amd.mock import random_reads
    reads, quals = random_reads(3, 33000, 34000, seed=9)
    compare_adaptor(oracle, oenc, enc, reads, quals, "ACGTACGTAC", 5, 1, [2], [5])


def test_quality_extremes(oracle, oenc, enc):
    # '!' (error prob 1: match score -inf) and '~'
    reads = ["ACGTACGTACGT", "ACGTACGTACGT", "ACGTACGTACGT"]
    quals = ["!" * 12, "~" * 12, "!~" * 6]
    compare_adaptor(oracle, oenc, enc, reads, quals, "ACGTACGT", 5, 1, [2], [4])


def test_barcode_align_global(oracle, oenc, enc):
    from sarlacc_amd import calls
    from sarlacc_amd.mock import random_reads
    for R, seed in ((12, 1), (8, 2), (40, 3), (0, 4)):
        rng = np.random.default_rng(seed)
        ref = "".join(rng.choice(list("ACGT"), R))
        reads, quals = random_reads(50, 0, 30, seed=seed)
        a = oracle.barcode_align(reads, quals, oenc, 5, 1, ref)
        b = calls.barcode_align(reads, quals, enc, 5, 1, ref)
        assert np.array_equal(bits(a), bits(b))


def test_general_align(oracle, oenc, enc):
    from sarlacc_amd import calls
    from sarlacc_amd.mock import random_reads
    rng = np.random.default_rng(32000)
    for trial, go in enumerate((5, 20, 1)):
        ref = "".join(rng.choice(list("ACGT"), 50))
        reads, quals = random_reads(40, 20, 80, seed=32000 + trial, qual_lo=43, qual_hi=83)
        reads.append("")
        quals.append("")
        a = oracle.general_align(reads, quals, oenc, go, 1, ref)
        b = calls.general_align(reads, quals, enc, go, 1, ref, False)
        assert np.array_equal(bits(a[0]), bits(b[0]))
        assert np.array_equal(a[1], b[1])
        assert a[2] == b[2] and a[3] == b[3]
        for r, q, read in zip(b[2], b[3], reads):
            assert len(r) == len(q)
            assert r.replace("-", "") == ref and q.replace("-", "") == read
        c = calls.general_align(reads, quals, enc, go, 1, ref, True)
        assert np.array_equal(a[1], c[1]) and c[2] == [] and c[3] == []
    # batch == one at a time (test-general-align.R:81-93)
    one = [calls.general_align([r], [q], enc, 5, 1, ref, False) for r, q in zip(reads[:5], quals[:5])]
    many = calls.general_align(reads[:5], quals[:5], enc, 5, 1, ref, False)
    assert [o[2][0] for o in one] == many[2]
    # empty reference
    a = oracle.general_align(reads[:4], quals[:4], oenc, 5, 1, "")
    b = calls.general_align(reads[:4], quals[:4], enc, 5, 1, "", False)
    assert np.array_equal(bits(a[0]), bits(b[0])) and a[2] == b[2] and a[3] == b[3] and np.array_equal(a[1], b[1])


def _mutate(rng, s, sub=0.05, indel=0.03):
    out = []
    for ch in s:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            out.append("ACGT"[rng.integers(0, 4)])
        out.append("ACGT"[rng.integers(0, 4)] if rng.random() < sub else ch)
    return "".join(out)


def test_references_beyond_1024_columns(oracle, oenc, enc):
    """References longer than one wavefront's 64 x 16 columns (qualityAlign of consensus reads against a transcript:
    R/qualityAlign.R:13-15, src/general_align.cpp:12-16; src/reference_align.cpp:7-13 takes any length) run with one
    workgroup per alignment (k_align_wide): scores bit for bit, edit distances, gapped strings, and the adaptor-mode map
    (starts, ends, sections) -- global and local, IUPAC columns, gap opening below zero, empty and short reads, and
    references beyond 8 192 columns, which go through in strips of 8 192 (the row state at a strip's last column reaches the
    next strip through HBM)."""
    from sarlacc_amd import calls
    rng = np.random.default_rng(1025)
    nuc = list("ACGT")
    for R, nreads, go in ((1025, 12, 5), (2000, 10, 5), (3000, 6, -0.5), (8200, 5, 5), (16500, 4, 5)):   # (8 200: two strips of columns; 16 500: three)
        ref = "".join(rng.choice(nuc, R))
        if R == 2000:   # ambiguity codes in the reference (2-fold, 3-fold, N)
            ref = ref[:100] + "RYNNB" + ref[105:1500] + "N" * 20 + ref[1520:]
        core = ref.replace("R", "A").replace("Y", "C").replace("N", "G").replace("B", "T")
        reads = [_mutate(rng, core) for _ in range(nreads - 3)]
        reads += ["", core[:37], _mutate(rng, core[200:900])]
        reads[0] = reads[0][:500] + "N" * 7 + reads[0][507:]
        quals = rand_quals(reads, R, lo=40, hi=83)
        a = oracle.general_align(reads, quals, oenc, go, 1, ref)
        b = calls.general_align(reads, quals, enc, go, 1, ref, False)
        assert np.array_equal(bits(a[0]), bits(b[0])), "scores differ (R = %d)" % R
        assert np.array_equal(a[1], b[1]), "edit distances differ (R = %d)" % R
        assert a[2] == b[2] and a[3] == b[3]
        for r, q, read in zip(b[2], b[3], reads):
            assert r.replace("-", "") == ref and q.replace("-", "") == read
        c = calls.general_align(reads, quals, enc, go, 1, ref, True)
        assert np.array_equal(a[1], c[1])
        # global scores alone (barcode_align) and the local mode with its map (adaptor_align)
        assert np.array_equal(bits(oracle.barcode_align(reads, quals, oenc, go, 1, ref)), bits(calls.barcode_align(reads, quals, enc, go, 1, ref)))
        if R <= 3000:
            long_reads = [rd + "".join(rng.choice(nuc, 300)) for rd in reads]
            compare_adaptor(oracle, oenc, enc, long_reads, rand_quals(long_reads, R + 1, lo=40, hi=83), ref, go, 1, [10, 0, 1500 % R], [R - 5, 40, R])
    from sarlacc_amd import SarlaccError
    with pytest.raises(SarlaccError, match="longer than 1048576 columns"):
        calls.barcode_align(["ACGT"], ["IIII"], enc, 5, 1, "A" * ((1 << 20) + 1))


def test_wide_references_both_kernels(oracle, oenc, enc):
    """Global mode with gapopen >= 0 on a reference of one strip runs on k_align_wide_q (wavefronts hand the row state over
    through LDS queues, no barrier per step); `align_wide_barrier` = 1 keeps k_align_wide, which everything else still takes
    (local mode, gapopen < 0, strips).  Both against the oracle on the same calls: reads from empty to longer than the
    reference, fewer rows than a wavefront has lanes, 2 to 16 wavefronts per alignment, packed input through the chunked
    host path."""
    from sarlacc_amd import _lib, calls
    rng = np.random.default_rng(4242)
    nuc = list("ACGT")
    try:
        for R in (1025, 1100, 2048, 4100, 8192):
            ref = "".join(rng.choice(nuc, R))
            reads = [_mutate(rng, ref) for _ in range(3)]
            reads += ["", ref[:5], ref[100:163], ref[7:71], ref[:700] + "N" * 9 + ref[709:1000],
                      "".join(rng.choice(nuc, 90)) + ref + "".join(rng.choice(nuc, 150)),
                      "".join(rng.choice(nuc, R - 40)), ref[R // 2:], ref[:R // 3] + ref[R // 3 + 150:]]   # unrelated; half; a 150-base deletion
            quals = rand_quals(reads, R, lo=35, hi=90)
            want = oracle.general_align(reads, quals, oenc, 2.5, 0.75, ref)
            want_s = oracle.barcode_align(reads, quals, oenc, 0, 1, ref)
            # (barrier, band): the old kernel; the queue kernel with the codes of every cell, with its default band of codes around
            # the main diagonal (the unrelated reads and the ones with flanks leave it: second launch), and with a band of 8 rows
            # (nearly every read is aligned twice)
            for barrier, band in ((1, 0), (0, -1), (0, 0), (0, 8)):
                calls.set_option("align_wide_barrier", barrier)
                calls.set_option("align_wide_band", band)
                got = calls.general_align(reads, quals, enc, 2.5, 0.75, ref, False)
                assert np.array_equal(bits(want[0]), bits(got[0])) and np.array_equal(want[1], got[1]), (R, barrier, band)
                assert want[2] == got[2] and want[3] == got[3], (R, barrier, band)
                assert np.array_equal(want[1], calls.general_align(reads, quals, enc, 2.5, 0.75, ref, True)[1]), (R, barrier, band)
                assert np.array_equal(bits(want_s), bits(calls.barcode_align(reads, quals, enc, 0, 1, ref))), (R, barrier, band)
    finally:
        calls.set_option("align_wide_barrier", 0)
        calls.set_option("align_wide_band", 0)


def test_wide_local_mode_across_strips(oracle, oenc, enc):
    """adaptor_align against references of two and three strips of 8 192 columns with reads of 1 to 3 kb: free vertical gaps only in
    the last column of the LAST strip, the map walk crossing strip tiles, boundary rows staged in several refills (reads longer
    than the 512 rows of one), sections straddling column 8 192."""
    rng = np.random.default_rng(8200)
    nuc = list("ACGT")
    for R in (8200, 16500):
        ref = "".join(rng.choice(nuc, R))
        reads = []
        for lo, n in ((7300, 1800), (R - 2600, 2500), (6000, 3000), (8100, 1000)):
            body = _mutate(rng, ref[lo:lo + n])
            reads.append("".join(rng.choice(nuc, int(rng.integers(0, 200)))) + body + "".join(rng.choice(nuc, int(rng.integers(0, 200)))))
        reads.append(ref[8150:8230])
        quals = rand_quals(reads, R, lo=40, hi=83)
        compare_adaptor(oracle, oenc, enc, reads, quals, ref, 5, 1, [0, 8000, 8100, 8191], [8192, min(8300, R), R, 8193])


def test_error_behaviour(oracle, oenc, enc):
    from sarlacc_amd import SarlaccError, calls
    with pytest.raises(SarlaccError, match="same length"):
        calls.adaptor_align(["ACGT"], ["III"], enc, 5, 1, ADAPTOR, [], [])
    with pytest.raises(SarlaccError, match="same length"):
        calls.adaptor_align(["ACGT"], ["IIII", "I"], enc, 5, 1, ADAPTOR, [], [])
    with pytest.raises(SarlaccError, match="unrecognized base in reference sequence"):
        calls.adaptor_align(["ACGT"], ["IIII"], enc, 5, 1, "ACXT", [], [])
    # no cell is evaluated for an empty read, so a bad adaptor character is not noticed
    out = calls.adaptor_align([""], [""], enc, 5, 1, "ACXT", [], [])
    assert out[0][0] == oracle.adaptor_align([""], [""], oenc, 5, 1, "ACXT")[0][0]
    with pytest.raises(SarlaccError, match="quality cannot be lower than smallest encoded value"):
        calls.adaptor_align(["ACGT"], ["II I"], enc, 5, 1, ADAPTOR, [], [])
    with pytest.raises(SarlaccError, match="gap opening penalty should be a numeric scalar"):
        calls.adaptor_align(["ACGT"], ["IIII"], enc, [5, 6], 1, ADAPTOR, [], [])
    with pytest.raises(SarlaccError, match="section starts and ends should have the same length"):
        calls.adaptor_align(["ACGT"], ["IIII"], enc, 5, 1, ADAPTOR, [1, 2], [3])
    import sarlacc_amd
    bad = sarlacc_amd.Encoding([0.1, 0.2], b"!\"")
    with pytest.raises(SarlaccError, match="error probabilities should decrease"):
        calls.adaptor_align(["ACGT"], ["!!!!"], bad, 5, 1, ADAPTOR, [], [])


def test_device_resident_and_packed_paths(oracle, oenc, enc):
    """sarlacc_dev_align on resident ASCII reads and on the 2-bit packed format
    (sarlacc_dev_pack_reads) must both equal the oracle, including reads with non-ACGT bases."""
    torch = pytest.importorskip("torch")
    from sarlacc_amd import device as sdev
    from sarlacc_amd.mock import random_reads
    from sarlacc_amd.strset import StringSet
    reads, quals = random_reads(301, 0, 400, seed=77, alphabet=b"ACGTACGTACGTNR")
    adaptor = "ACGTNNNNACGTRYACGT"
    want = oracle.adaptor_align(reads, quals, oenc, 5, 1, adaptor, [4], [8])
    s, q = StringSet.from_strings(reads), StringSet.from_strings(quals)
    dev = torch.device("cuda", 0)
    n, total = len(s), s.total
    d_seq = torch.from_numpy(s.chars).to(dev)
    d_qual = torch.from_numpy(q.chars).to(dev)
    d_off = torch.from_numpy(s.off).to(dev)
    max_len = int(s.widths().max())
    stream = torch.cuda.current_stream().cuda_stream
    packed = torch.zeros(total // 4 + 2, dtype=torch.uint8, device=dev)
    nmask = torch.zeros(total // 8 + 1, dtype=torch.uint8, device=dev)
    sdev.dev_pack_reads(d_seq, total, packed, nmask, stream)
    for seq_buf, mask in ((d_seq, None), (packed, nmask)):
        sc = torch.zeros(n, dtype=torch.float64, device=dev)
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        en = torch.zeros_like(st)
        so = torch.zeros_like(st)
        sw = torch.zeros_like(st)
        sdev.dev_align(seq_buf, d_qual, d_off, n, max_len, enc, 5, 1, adaptor, True, [4], [8], sc, st, en, so, sw,
                       stream, d_nmask=mask)
        torch.cuda.synchronize()
        assert np.array_equal(bits(sc.cpu().numpy()), bits(want[0]))
        assert np.array_equal(st.cpu().numpy(), want[1]) and np.array_equal(en.cpu().numpy(), want[2])
        assert np.array_equal(so.cpu().numpy(), want[3][0]) and np.array_equal(sw.cpu().numpy(), want[4][0])
        sc2 = torch.zeros(n, dtype=torch.float64, device=dev)
        sdev.dev_align(seq_buf, d_qual, d_off, n, max_len, enc, 5, 1, adaptor, False, (), (), sc2, None, None, None, None,
                       stream, d_nmask=mask)
        torch.cuda.synchronize()
        assert np.array_equal(bits(sc2.cpu().numpy()), bits(oracle.barcode_align(reads, quals, oenc, 5, 1, adaptor)))


def test_wide_references_on_resident_packed_and_chunked_paths(oracle, oenc, enc):
    """k_align_wide behind the other entry points: sarlacc_dev_align on resident ASCII reads and on the 2-bit packed format
    (reads with non-ACGT bases; local mode with sections, global scores), and a host call cut into upload chunks
    (option align_chunks: every chunk one launch on a slice, sections strided over the whole batch, the first bad quality of
    the batch reported)."""
    torch = pytest.importorskip("torch")
    from sarlacc_amd import SarlaccError, calls
    from sarlacc_amd import device as sdev
    from sarlacc_amd.mock import random_reads
    from sarlacc_amd.strset import StringSet
    rng = np.random.default_rng(1500)
    ref = "".join(rng.choice(list("ACGT"), 1500))
    ref = ref[:700] + "NNNNRY" + ref[706:]
    reads, quals = random_reads(60, 0, 500, seed=78, alphabet=b"ACGTACGTACGTNR")
    for k in range(0, 60, 7):   # some reads that carry a piece of the reference
        lo = int(rng.integers(0, 1200))
        piece = ref[lo:lo + int(rng.integers(50, 300))].replace("N", "A").replace("R", "G").replace("Y", "T")
        reads[k] = reads[k][:100] + piece + reads[k][100:]
        quals[k] = quals[k][:100] + "I" * len(piece) + quals[k][100:]
    want = oracle.adaptor_align(reads, quals, oenc, 5, 1, ref, [700, 10], [706, 1400])
    want_global = oracle.barcode_align(reads, quals, oenc, 5, 1, ref)
    s, q = StringSet.from_strings(reads), StringSet.from_strings(quals)
    dev = torch.device("cuda", 0)
    n, total = len(s), s.total
    d_seq = torch.from_numpy(s.chars).to(dev)
    d_qual = torch.from_numpy(q.chars).to(dev)
    d_off = torch.from_numpy(s.off).to(dev)
    max_len = int(s.widths().max())
    stream = torch.cuda.current_stream().cuda_stream
    packed = torch.zeros(total // 4 + 2, dtype=torch.uint8, device=dev)
    nmask = torch.zeros(total // 8 + 1, dtype=torch.uint8, device=dev)
    sdev.dev_pack_reads(d_seq, total, packed, nmask, stream)
    for seq_buf, mask in ((d_seq, None), (packed, nmask)):
        sc = torch.zeros(n, dtype=torch.float64, device=dev)
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        en = torch.zeros_like(st)
        so = torch.zeros(2 * n, dtype=torch.int32, device=dev)
        sw = torch.zeros_like(so)
        sdev.dev_align(seq_buf, d_qual, d_off, n, max_len, enc, 5, 1, ref, True, [700, 10], [706, 1400], sc, st, en, so, sw,
                       stream, d_nmask=mask)
        torch.cuda.synchronize()
        assert np.array_equal(bits(sc.cpu().numpy()), bits(want[0]))
        assert np.array_equal(st.cpu().numpy(), want[1]) and np.array_equal(en.cpu().numpy(), want[2])
        assert np.array_equal(so.cpu().numpy().reshape(2, n), np.stack(want[3])) and np.array_equal(sw.cpu().numpy().reshape(2, n), np.stack(want[4]))
        sc2 = torch.zeros(n, dtype=torch.float64, device=dev)
        sdev.dev_align(seq_buf, d_qual, d_off, n, max_len, enc, 5, 1, ref, False, (), (), sc2, None, None, None, None,
                       stream, d_nmask=mask)
        torch.cuda.synchronize()
        assert np.array_equal(bits(sc2.cpu().numpy()), bits(want_global))
    calls.set_option("align_chunks", 3)
    try:
        got = calls.adaptor_align(reads, quals, enc, 5, 1, ref, [700, 10], [706, 1400])
        assert np.array_equal(bits(got[0]), bits(want[0])) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        for k in range(2):
            assert np.array_equal(got[3][k], want[3][k]) and np.array_equal(got[4][k], want[4][k])
        assert np.array_equal(bits(calls.barcode_align(reads, quals, enc, 5, 1, ref)), bits(want_global))
        bad_q = list(quals)
        bad_q[50] = " " + bad_q[50][1:]
        with pytest.raises(SarlaccError, match="quality cannot be lower"):
            calls.adaptor_align(reads, bad_q, enc, 5, 1, ref, [700], [706])
    finally:
        calls.set_option("align_chunks", 0)


def test_unmask_alignment(oracle):
    """unmask_alignment (src/unmask_alignment.cpp): reference literals, random rows wider than one
    wave step, and the reference's error cases (tests/testthat/test-masking.R:45-100)."""
    from sarlacc_amd import calls
    from sarlacc_amd._lib import SarlaccError
    from tests.test_oracle_align import UNMASK_CASES, unmask_inputs
    for seq_in in UNMASK_CASES:
        masked, original = unmask_inputs(seq_in)
        assert calls.unmask_alignment(masked, original) == [s.upper() for s in seq_in]
    rng = np.random.default_rng(12)
    W = 700
    rows, originals = [], []
    for _ in range(40):
        row = rng.choice(list("ACGT-"), W, p=[0.22, 0.22, 0.22, 0.22, 0.12])
        originals.append("".join(c for c in row if c != "-"))
        m = row.copy()
        hit = (rng.random(W) < 0.3) & (row != "-")
        m[hit] = rng.choice(list("Nn"), int(hit.sum()))
        rows.append("".join(m))
    assert calls.unmask_alignment(rows, originals) == oracle.unmask_alignment(rows, originals)
    assert calls.unmask_alignment([], []) == []
    for aln, orig, msg in ((["AA-AA"], ["AAA"], "different lengths"),
                           (["NNNNN"], ["AAA"], "sequence in alignment string is longer than the original"),
                           (["AAAA", "GGGG"], ["AAA"], "same number of entries"),
                           (["AAAA", "GGG"], ["AAAA", "GGG"], "alignment strings should have the same length"),
                           (["ANAA", "GGNNN"[:4] + "N"], ["AAAA", "GG"], "alignment strings should have the same length")):
        with pytest.raises(SarlaccError, match=msg):
            calls.unmask_alignment(aln, orig)
    # first failing row decides the message
    with pytest.raises(SarlaccError, match="longer than the original"):
        calls.unmask_alignment(["AAAA", "NNNN", "AA-A"], ["AAAA", "GG", "A"])


@pytest.fixture
def chunked(request):
    from sarlacc_amd import calls
    calls.set_option("align_chunks", request.param)
    yield request.param
    calls.set_option("align_chunks", 0)


@pytest.mark.parametrize("chunked", [2, 3, 7], indirect=True)
def test_chunked_host_calls_match_single_launch(oracle, oenc, enc, chunked):
    """Large host-pointer calls are sent to the device in chunks whose upload overlaps the
    previous chunk's kernel (option align_chunks forces it on a small batch): same scores,
    positions, sections and the same first error as the single launch and the oracle."""
    from sarlacc_amd import calls
    from sarlacc_amd._lib import SarlaccError
    from sarlacc_amd.mock import random_reads
    reads, quals = random_reads(101, 0, 300, seed=40 + chunked)
    adaptor = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"
    want = oracle.adaptor_align(reads, quals, oenc, 5, 1, adaptor, [9, 0], [21, 30])
    got = calls.adaptor_align(reads, quals, enc, 5, 1, adaptor, [9, 0], [21, 30])
    assert np.array_equal(got[0].view(np.int64), want[0].view(np.int64))
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    for k in range(2):
        assert np.array_equal(got[3][k], want[3][k]) and np.array_equal(got[4][k], want[4][k])
    assert np.array_equal(calls.adaptor_align_score_only(reads, quals, enc, 5, 1, adaptor).view(np.int64), want[0].view(np.int64))
    assert np.array_equal(calls.barcode_align(reads, quals, enc, 5, 1, "ACGTTGCA").view(np.int64),
                          oracle.barcode_align(reads, quals, oenc, 5, 1, "ACGTTGCA").view(np.int64))
    # a quality below the offset in a late chunk and an earlier, longer-than-quality read: the first one wins
    bad_q = list(quals)
    bad_q[90] = " " + bad_q[90][1:] if bad_q[90] else bad_q[90]
    if bad_q[90]:
        with pytest.raises(SarlaccError, match="quality cannot be lower"):
            calls.adaptor_align(reads, bad_q, enc, 5, 1, adaptor, [9], [21])
