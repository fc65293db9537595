"""Randomised differential fuzzing of every C-ABI entry point against the CPU oracle.
Usage: python tools/fuzz.py [seconds] [seed] [case name, e.g. case_msa].  Prints the first mismatch (with a
reproducer seed) or a summary of the cases run."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls
from oracle import oracle as O

IUPAC = "ACGTMRWSYKVHDBN"
enc, oenc = sarlacc_amd.phred_encoding(), O.phred_encoding()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def rstr(rng, n, alphabet):
    return "".join(rng.choice(list(alphabet), n)) if n else ""


def rqual(rng, n, lo=33, hi=126):
    return "".join(chr(int(x)) for x in rng.integers(lo, hi + 1, n)) if n else ""


def both(f_gpu, f_orc):
    """Run both; equal exceptions (by message) count as agreement."""
    try:
        g = f_gpu()
        ge = None
    except sarlacc_amd.SarlaccError as e:
        g, ge = None, str(e)
    try:
        o = f_orc()
        oe = None
    except O.OracleError as e:
        o, oe = None, str(e)
    if (ge is None) != (oe is None) or (ge is not None and oe not in ge and ge not in oe):
        raise AssertionError("error behaviour differs: gpu=%r oracle=%r" % (ge, oe))
    return g, o, ge is not None


def case_align(rng):
    R = int(rng.choice([0, 1, 2, 5, 12, 16, 17, 22, 30, 31, 33, 48, 64, 65, 90, 130, 257, 700]))
    n = int(rng.integers(1, 40))
    Lmax = int(rng.choice([0, 3, 30, 150, 700]))
    alpha_r = IUPAC if rng.random() < 0.5 else "ACGT"
    adaptor = rstr(rng, R, alpha_r)
    read_alpha = "ACGT" if rng.random() < 0.7 else "ACGTN"
    reads = [rstr(rng, int(rng.integers(0, Lmax + 1)), read_alpha) for _ in range(n)]
    if R and rng.random() < 0.7:   # plant a noisy copy
        k = int(rng.integers(0, n))
        core = "".join(c if c in "ACGT" else "ACGT"[int(rng.integers(0, 4))] for c in adaptor)
        pos = int(rng.integers(0, len(reads[k]) + 1))
        reads[k] = reads[k][:pos] + core + reads[k][pos:]
    qlo = 33 if rng.random() < 0.5 else 40
    quals = [rqual(rng, len(r), qlo, int(rng.choice([75, 126]))) for r in reads]
    go, ge = [(5, 1), (20, 1), (1, 1), (2.5, 0.75), (0, 1), (3, 0), (7.25, 2.5), (-1, 2), (0.1, 0.7), (-0.5, 0.5)][int(rng.integers(0, 10))]
    nsec = int(rng.integers(0, 4)) if R else 0
    ss = sorted(int(x) for x in rng.integers(0, max(R, 1), nsec))
    se = [int(rng.integers(s, R) + 1) for s in ss]
    g, o, err = both(lambda: calls.adaptor_align(reads, quals, enc, go, ge, adaptor, ss, se),
                     lambda: O.adaptor_align(reads, quals, oenc, go, ge, adaptor, ss, se))
    if not err:
        assert np.array_equal(bits(g[0]), bits(o[0])), "adaptor scores"
        assert np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2]), "adaptor positions"
        for a, b in zip(g[3] + g[4], o[3] + o[4]):
            assert np.array_equal(a, b), "sections"
    g, o, err = both(lambda: calls.barcode_align(reads, quals, enc, go, ge, adaptor),
                     lambda: O.barcode_align(reads, quals, oenc, go, ge, adaptor))
    if not err:
        assert np.array_equal(bits(g), bits(o)), "barcode scores"
    if R <= 130 and Lmax <= 150:
        g, o, err = both(lambda: calls.general_align(reads, quals, enc, go, ge, adaptor, False),
                         lambda: O.general_align(reads, quals, oenc, go, ge, adaptor))
        if not err:
            assert np.array_equal(bits(g[0]), bits(o[0])) and np.array_equal(g[1], o[1]), "general scores/edits"
            assert g[2] == o[2] and g[3] == o[3], "general strings"


def case_align_wide(rng):
    """References beyond 1 024 columns (k_align_wide: one workgroup per alignment): every mode, reads from empty to longer
    than the reference, with and without a noisy copy of the reference inside."""
    R = int(rng.choice([1025, 1030, 1088, 1500, 2047, 2048, 2049, 2600]))
    strips = rng.random() < 0.08   # beyond 8 192 columns: strips of columns (short reads only: the oracle fills every cell)
    if strips:
        R = int(rng.choice([8193, 8200, 9000, 16390]))
    n = int(rng.integers(1, 7))
    ref = rstr(rng, R, IUPAC if rng.random() < 0.3 else "ACGT")
    core = "".join(c if c in "ACGT" else "ACGT"[int(rng.integers(0, 4))] for c in ref)
    reads = []
    for _ in range(n):
        kind = int(rng.integers(0, 2 if strips else 4))
        if kind == 0:
            reads.append(rstr(rng, int(rng.integers(0, 60)), "ACGTN"))
        elif kind == 1:
            lo = int(rng.integers(0, R - 200)); hi = int(rng.integers(lo + 1, min(R, lo + 600 if strips else R) + 1))
            reads.append("".join(c for c in core[lo:hi] if rng.random() > 0.03))
        else:   # a noisy copy, sometimes with flanks
            body = "".join(("ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.06 else c) for c in core if rng.random() > 0.02)
            reads.append(rstr(rng, int(rng.integers(0, 80)) * (kind == 3), "ACGT") + body + rstr(rng, int(rng.integers(0, 80)) * (kind == 3), "ACGT"))
    quals = [rqual(rng, len(r), 40, int(rng.choice([75, 126]))) for r in reads]
    go, ge = [(5, 1), (20, 1), (1, 1), (2.5, 0.75), (0, 1), (3, 0), (-1, 2), (-0.5, 0.5)][int(rng.integers(0, 8))]
    g, o, err = both(lambda: calls.general_align(reads, quals, enc, go, ge, ref, False),
                     lambda: O.general_align(reads, quals, oenc, go, ge, ref))
    if not err:
        assert np.array_equal(bits(g[0]), bits(o[0])) and np.array_equal(g[1], o[1]), "wide general scores/edits"
        assert g[2] == o[2] and g[3] == o[3], "wide general strings"
    nsec = int(rng.integers(0, 4))
    ss = sorted(int(x) for x in rng.integers(0, R, nsec))
    se = [int(rng.integers(s_, R) + 1) for s_ in ss]
    g, o, err = both(lambda: calls.adaptor_align(reads, quals, enc, go, ge, ref, ss, se),
                     lambda: O.adaptor_align(reads, quals, oenc, go, ge, ref, ss, se))
    if not err:
        assert np.array_equal(bits(g[0]), bits(o[0])), "wide adaptor scores"
        assert np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2]), "wide adaptor positions"
        for a, b in zip(g[3] + g[4], o[3] + o[4]):
            assert np.array_equal(a, b), "wide sections"
    g, o, err = both(lambda: calls.barcode_align(reads, quals, enc, go, ge, ref),
                     lambda: O.barcode_align(reads, quals, oenc, go, ge, ref))
    if not err:
        assert np.array_equal(bits(g), bits(o)), "wide barcode scores"


def umi_switches(rng):
    """Thresholds 2 and 3 through the split-key search also on small sets (half of the cases), dense graphs clustered
    by rounds over every list instead of candidate sets (a quarter)."""
    calls.set_option("umi_split_min", int(rng.choice([0, 16, 200])))
    calls.set_option("umi_full_rounds", int(rng.random() < 0.25))


def case_umi(rng):
    n = int(rng.integers(2, 400))
    length = int(rng.choice([0, 3, 8, 12, 20, 32, 33, 48, 80, 127, 129, 260]))   # beyond 32: the 4-word path (up to 128 bases); beyond 128: words from HBM
    base = [rstr(rng, min(127 if length <= 127 else 1023, max(0, length + int(rng.integers(-2, 3)))), "ACGT") for _ in range(max(1, n // 8))]
    alpha = "ACGT" if rng.random() < 0.7 else "ACGTN"
    umis = []
    for _ in range(n):
        b = list(base[int(rng.integers(0, len(base)))])
        for k in range(len(b)):
            if rng.random() < 0.08:
                b[k] = alpha[int(rng.integers(0, len(alpha)))]
        if b and rng.random() < 0.1:
            del b[int(rng.integers(0, len(b)))]
        if len(b) < (128 if length <= 127 else 1024) and rng.random() < 0.1:
            b.insert(int(rng.integers(0, len(b) + 1)), "ACGT"[int(rng.integers(0, 4))])
        umis.append("".join(b))
    limit = int(rng.integers(0, 5))
    umi_switches(rng)
    ngr = int(rng.choice([1, 1, 3, 20]))
    lab = rng.integers(0, ngr, n)
    groups = [(np.flatnonzero(lab == k) + 1).astype(np.int32) for k in range(ngr)]
    use2 = rng.random() < 0.3
    umi2 = [rstr(rng, 5, "ACGT") if rng.random() < 0.5 else "ACGTA" for _ in range(n)] if use2 else None
    l2 = int(rng.integers(0, 3))
    g, o, err = both(lambda: calls.umi_group(umis, limit, umi2, l2, groups), lambda: O.umi_group(umis, limit, umi2, l2, groups, fast=True))
    if not err:
        assert len(g) == len(o) and all(np.array_equal(a, b) for a, b in zip(g, o)), "umi_group"
    g, o, err = both(lambda: calls.fast_levdist_test(umis, limit), lambda: O.fast_levdist_test(umis, limit))
    if not err:
        assert all(np.array_equal(a, b) for a, b in zip(g, o)), "fast_levdist"
    if n <= 120:
        g, o, err = both(lambda: calls.compute_lev_masked(umis), lambda: O.compute_lev_masked(umis))
        if not err:
            assert g.tolist() == o.tolist(), "lev dense"


def case_consensus(rng):
    ngroups = int(rng.integers(1, 12))
    alns, quals = [], []
    p_n = float(rng.choice([0.0, 0.0, 0.03]))          # without N the groups take the fast kernel (k_consensus_qf)
    for _ in range(ngroups):
        nrows = int(rng.integers(0, int(rng.choice([30, 30, 80]))))   # beyond 64 rows: the generic kernel
        W = int(rng.integers(0, int(rng.choice([400, 400, 1500]))))
        truth = rng.choice(list("ACGT"), W) if W else np.array([], dtype="<U1")
        rows, qs = [], []
        for _ in range(nrows):
            r = truth.copy()
            if W:
                sub = rng.random(W) < 0.1
                r[sub] = rng.choice(list("ACGT"), int(sub.sum()))
                r[rng.random(W) < p_n] = "N"
                if rng.random() < 0.02:
                    r[int(rng.integers(0, W))] = "acgtRY"[int(rng.integers(0, 6))]
                r[rng.random(W) < rng.choice([0.05, 0.4])] = "-"
            row = "".join(r)
            rows.append(row)
            qs.append(rqual(rng, len(row.replace("-", ""))))
        alns.append(rows)
        quals.append(qs)
    cov = float(rng.choice([0.0, 0.2, 0.5, 0.6, 0.9, 1.0]))
    pc = float(rng.choice([0.5, 1.0, 2.0, 4.0]))
    g, o, err = both(lambda: calls.create_consensus_quality_loop(alns, cov, quals, enc),
                     lambda: O.create_consensus_quality_loop(alns, cov, quals, oenc))
    if not err:
        assert g[0] == list(o[0]) and g[1] == list(o[1]), "consensus quality"
    g, o, err = both(lambda: calls.create_consensus_basic_loop(alns, cov, pc), lambda: O.create_consensus_basic_loop(alns, cov, pc))
    if not err:
        assert g[0] == list(o[0]) and g[1] == list(o[1]), "consensus basic"


def case_msa(rng):
    from sarlacc_amd.mock import NUC, mutate
    reads, groups = [], []
    many = rng.random() < 0.15   # many small groups: workgroups of the merge kernel take several groups each
    for _ in range(int(rng.integers(8, 48)) if many else int(rng.integers(1, 6))):
        L = int(rng.choice([0, 5, 40, 90])) if many else int(rng.choice([0, 5, 60, 300, 900]))
        truth = NUC[rng.integers(0, 4, L)]
        idx = []
        for _ in range(int(rng.integers(0, 9))):
            r = mutate(truth, rng, 0.08, 0.03).tobytes().decode() if L else ""
            if rng.random() < 0.1:
                r = r[: len(r) // 2]
            reads.append(r)
            idx.append(len(reads))
        groups.append(idx)
    if not reads:
        reads = ["ACGT"]
    if rng.random() < 0.2 and len(groups) >= 2 and groups[0] and groups[1]:   # a UMI collision: two molecules in one cluster
        groups[0] = groups[0] + groups[1]
        groups[1] = []
    if rng.random() < 0.25:   # a large cluster of one to four molecules: 13-64 reads, the 4- and 8-wavefront workgroups of spec v2 (64-bit member sets from 33 on)
        L = int(rng.choice([30, 80, 200]))
        truths = [NUC[rng.integers(0, 4, int(L * rng.uniform(0.7, 1.3)))] for _ in range(int(rng.integers(1, 5)))]
        idx = []
        for k in range(int(rng.integers(13, 33)) if rng.random() < 0.7 else int(rng.integers(33, 65))):
            reads.append(mutate(truths[int(rng.integers(0, len(truths)))], rng, 0.06, 0.02).tobytes().decode())
            idx.append(len(reads))
        groups.insert(int(rng.integers(0, len(groups) + 1)), idx)
    if rng.random() < 0.15 and groups and groups[0]:     # a length outlier and a huge bandwidth: the band cap of the spec
        k = groups[0][0] - 1
        reads[k] = reads[k] + mutate(NUC[rng.integers(0, 4, int(rng.choice([300, 1100, 1500])))], rng, 0.0, 0.0).tobytes().decode()
    params = [(0, -1, -5, -1), (0, -1, -1, -5), (1, -2, -2, -2), (2, -3, -1, -4), (0, -1, -1, -1)][int(rng.integers(0, 5))]
    bw = int(rng.choice([0, 3, 20, 100, 180, 600, 5000]))
    spec = int(rng.choice([1, 2, 2, 2]))
    opts = [o for o in ("msa2_general_rows", "msa2_chain_hbm", "msa2_single_wave", "msa2_batches", "msa2_simple_extend", "msa2_wide_extend") if rng.random() < 0.15]   # the other code paths of spec v2
    calls.set_msa_spec(spec)
    for o_ in opts:
        calls.set_option(o_, 3 if o_ == "msa2_batches" else (64 if o_ == "msa2_wide_extend" else 1))   # (msa2_wide_extend = 64: the four-positions kernel for every group size)
    try:
        g, o, err = both(lambda: calls.quick_msa(groups, reads, *params, bw), lambda: O.quick_msa(groups, reads, *params, bw, spec=spec))
    finally:
        calls.set_msa_spec(0)
        for o_ in opts:
            calls.set_option(o_, 0)
    if not err:
        assert g == o, "msa rows (spec %d)" % spec


def case_mask(rng):
    n = int(rng.integers(0, 50))
    seqs = [rstr(rng, int(rng.integers(0, 80)), "ACGTN") for _ in range(n)]
    quals = [rqual(rng, len(s)) for s in seqs]
    thr = float(rng.choice([0.0, 0.001, 0.01, 0.1, 0.5, 1.0]))
    g, o, err = both(lambda: calls.mask_bad_bases(seqs, quals, enc, thr), lambda: O.mask_bad_bases(seqs, quals, oenc, thr))
    if not err:
        assert g == o, "mask"


def case_umi_large(rng):
    """Enough UMIs for the tile-prefix prefilter (>= 16 tiles of 256): few pre-groups, variable
    lengths, occasional N, limits 0-3; umi_group and the neighbour lists against the oracle."""
    n = int(rng.integers(4200, 9000))
    length = int(rng.choice([6, 8, 10, 12, 16]))
    nbase = int(rng.choice([n // 10, n // 3, n]))
    NUCS = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = NUCS[rng.integers(0, 4, (nbase, length))]
    pick = base[rng.integers(0, nbase, n)].copy()
    sub = rng.random(pick.shape) < 0.04
    pick[sub] = NUCS[rng.integers(0, 4, int(sub.sum()))]
    if rng.random() < 0.3:
        pick[rng.random(pick.shape) < 0.002] = ord("N")
    umis = []
    for row in pick:
        t = row.tobytes().decode()
        r = rng.random()
        if r < 0.05 and len(t) > 1:
            k = int(rng.integers(0, len(t)))
            t = t[:k] + t[k + 1:]
        elif r < 0.1:
            k = int(rng.integers(0, len(t) + 1))
            t = t[:k] + "ACGT"[int(rng.integers(0, 4))] + t[k:]
        umis.append(t)
    limit = int(rng.integers(0, 4))
    umi_switches(rng)
    ngr = int(rng.choice([1, 1, 2, 5]))
    lab = np.sort(rng.integers(0, ngr, n)) if rng.random() < 0.5 else rng.integers(0, ngr, n)
    groups = [(np.flatnonzero(lab == k) + 1).astype(np.int32) for k in range(ngr)]
    g, o, err = both(lambda: calls.umi_group(umis, limit, None, limit, groups), lambda: O.umi_group(umis, limit, None, limit, groups, fast=True))
    if not err:
        assert len(g) == len(o) and all(np.array_equal(a, b) for a, b in zip(g, o)), "umi_group (large)"


def case_unmask(rng):
    n = int(rng.integers(0, 30))
    W = int(rng.integers(0, 300))
    rows, orig = [], []
    for _ in range(n):
        row = rng.choice(list("ACGT-"), W, p=[0.2, 0.2, 0.2, 0.2, 0.2]) if W else np.array([], dtype="<U1")
        o = "".join(c for c in row if c != "-")
        m = row.copy()
        if W:
            hit = (rng.random(W) < 0.3) & (row != "-")
            m[hit] = rng.choice(list("Nn"), int(hit.sum()))
        r = rng.random()
        if r < 0.05:
            o = o[:-1] if o else o + "A"          # original too short
        elif r < 0.1:
            o = o + "C"                            # original too long
        rows.append("".join(m))
        orig.append(o)
    if n and rng.random() < 0.05:
        orig = orig[:-1]                           # entry counts differ
    if n > 1 and rng.random() < 0.05:
        rows[-1] = rows[-1] + "A"                  # ragged alignment
    g, o, err = both(lambda: calls.unmask_alignment(rows, orig), lambda: O.unmask_alignment(rows, orig))
    if not err:
        assert g == o, "unmask"


def case_fused(rng):
    """sarlacc_msa_consensus against the oracle's quick_msa followed by its consensus."""
    from sarlacc_amd.mock import NUC, mutate
    from sarlacc_amd.strset import csr_from_lists
    reads, groups = [], []
    for _ in range(int(rng.integers(1, 8))):
        L = int(rng.choice([0, 5, 60, 300, 700]))
        truth = NUC[rng.integers(0, 4, L)]
        idx = []
        for _ in range(int(rng.integers(0, 9))):
            reads.append(mutate(truth, rng, 0.08, 0.03).tobytes().decode() if L else "")
            idx.append(len(reads))
        groups.append(idx)
    if not reads:
        reads = ["ACGT"]
    if rng.random() < 0.3:      # N in the reads, other characters in single-read groups (returned verbatim by the MSA)
        reads = ["".join("N" if rng.random() < 0.03 else c for c in r) for r in reads]
        for g in groups:
            if len(g) == 1 and reads[g[0] - 1]:
                r = list(reads[g[0] - 1]); r[int(rng.integers(0, len(r)))] = "acgtRYn"[int(rng.integers(0, 7))]; reads[g[0] - 1] = "".join(r)
    quals = [rqual(rng, len(r), 40, 90) for r in reads]
    if rng.random() < 0.05 and quals[0]:   # a quality below the encoding: the reference's error
        quals[0] = " " + quals[0][1:]
    params = [(0, -1, -5, -1), (0, -1, -1, -5), (1, -2, -2, -2)][int(rng.integers(0, 3))]
    bw = int(rng.choice([3, 20, 100]))
    cov = float(rng.choice([0.0, 0.5, 0.6, 1.0]))
    goff, gvals = csr_from_lists(groups)
    rows = O.quick_msa(groups, reads, *params, bw)
    if rng.random() < 0.5:
        g, o, err = both(lambda: calls.msa_consensus_flat(goff, gvals, reads, *params, bw, cov, quals=quals, encoding=enc),
                         lambda: O.create_consensus_quality_loop(rows, cov, [[quals[i - 1] for i in g] for g in groups], oenc))
    else:
        g, o, err = both(lambda: calls.msa_consensus_flat(goff, gvals, reads, *params, bw, cov, pseudo_count=1.0),
                         lambda: O.create_consensus_basic_loop(rows, cov, 1.0))
    if not err:
        assert g[0].to_strings() == list(o[0]) and g[1].to_strings() == list(o[1]), "fused msa+consensus"


def case_fastq(rng):
    from sarlacc_amd.resident import DeviceReads
    n = int(rng.integers(0, 60))
    eol = "\r\n" if rng.random() < 0.3 else "\n"
    recs, seqs, quals, names = [], [], [], []
    for i in range(n):
        L = int(rng.integers(0, int(rng.choice([5, 80, 900, 9000])) + 1))
        sq = rstr(rng, L, "ACGTNacgt")
        q = rqual(rng, L)
        nm = "r%d %s" % (i, rstr(rng, int(rng.integers(0, 12)), "abc:/ =0123"))
        recs.append("@%s%s%s%s+%s%s" % (nm, eol, sq, eol, eol, q))
        seqs.append(sq.upper()); quals.append(q); names.append(nm)
    text = eol.join(recs) + (eol if (n and rng.random() < 0.7) else "") + ("\n" * int(rng.integers(0, 3)) if n else "")
    dev = DeviceReads.from_fastq(text.encode())
    s_, q_ = dev.download()
    assert len(dev) == n and s_.to_strings() == seqs and q_.to_strings() == quals and dev.names == names, "fastq"
    # the same file streamed in chunks of `number` records through blocks that end anywhere
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".fastq") as fh:
        fh.write(text.encode()); fh.flush()
        number, block = int(rng.integers(1, 40)), int(rng.choice([37, 512, 4096, 1 << 20]))
        cs, cq, cn, sizes = [], [], [], []
        for chunk in DeviceReads.stream_fastq(fh.name, number, block_bytes=block):
            a, b = chunk.download()
            cs += a.to_strings(); cq += b.to_strings(); cn += list(chunk.names); sizes.append(len(chunk))
        assert cs == seqs and cq == quals and cn == names and all(x == number for x in sizes[:-1]) and (not sizes or 0 < sizes[-1] <= number), "fastq chunks"


def case_profile(rng):
    """find_homopolymers / match_homopolymers / find_errors on random gapped pairwise alignments (occasionally
    malformed: unequal lengths, a stray character, a reference that changes)."""
    n = int(rng.integers(0, 40))
    L = int(rng.choice([0, 1, 8, 60, 400]))
    ref = rng.choice(list("AACGTT"), L) if L else np.array([], dtype="<U1")
    refs, reads = [], []
    for _ in range(n):
        a, b = [], []
        for c in ref:
            u = rng.random()
            if u < 0.05:
                a.append(c); b.append("-")
            elif u < 0.1:
                k = int(rng.integers(1, 4)); a += ["-"] * k; b += list(rng.choice(list("ACGT"), k)); a.append(c); b.append(c)
            else:
                a.append(c); b.append(c if rng.random() > 0.1 else "ACGT"[int(rng.integers(0, 4))])
        refs.append("".join(a)); reads.append("".join(b))
    if n and rng.random() < 0.1:
        k = int(rng.integers(0, n))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            reads[k] = reads[k] + "A"
        elif kind == 1 and reads[k]:
            reads[k] = "N" + reads[k][1:]
        else:
            refs[k] = refs[k] + "ACGT"; reads[k] = reads[k] + "ACGT"
    same = lambda g, o: all((x if isinstance(x, (list, str)) else np.asarray(x).tolist()) == (y if isinstance(y, (list, str)) else np.asarray(y).tolist()) for x, y in zip(g, o))
    g, o, err = both(lambda: calls.find_homopolymers(refs + reads), lambda: O.find_homopolymers(refs + reads))
    assert err or same(g, o), "find_homopolymers"
    g, o, err = both(lambda: calls.match_homopolymers(refs, reads), lambda: O.match_homopolymers(refs, reads))
    assert err or same(g, o), "match_homopolymers"
    g, o, err = both(lambda: calls.find_errors(refs, reads), lambda: O.find_errors(refs, reads))
    assert err or same(g, o), "find_errors"


CASES = [case_align, case_align, case_umi, case_consensus, case_msa, case_mask, case_unmask, case_fused, case_fastq, case_profile]

if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3:
        CASES = [globals()[sys.argv[3]]]
    t0 = last_note = time.time()
    counts = {}
    k = 0
    while time.time() - t0 < budget:
        seed = seed0 * 1_000_003 + k
        rng = np.random.default_rng(seed)
        fn = case_umi_large if k % 97 == 96 else (case_align_wide if k % 41 == 40 else CASES[k % len(CASES)])
        try:
            fn(rng)
        except Exception:
            print("MISMATCH in %s with seed %d" % (fn.__name__, seed))
            traceback.print_exc()
            sys.exit(1)
        counts[fn.__name__] = counts.get(fn.__name__, 0) + 1
        k += 1
        if time.time() - last_note > 60:   # a long run has to show signs of life
            last_note = time.time()
            print("... %d cases after %.0f s" % (k, last_note - t0), flush=True)
    print("fuzz ok: %d cases in %.0f s: %s" % (k, time.time() - t0, counts))
