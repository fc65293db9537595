"""`.Call`-level interface: one function per native routine the reference registers
(/root/reference/src/init.cpp:9-35), same names and argument order, operating on
Python lists of strings / numpy arrays instead of SEXPs.  Each function is a thin
ctypes shim over the C ABI in include/sarlacc_amd.h -- the exact counterpart of the
R glue shown in INTEGRATION.md.  All arithmetic happens in the HIP library.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SarlaccError, check, ptr
from .encoding import as_encoding
from .strset import StringSet, csr_from_lists, lists_from_csr


def _scalar(x, what, kind):
    """check_*_scalar of the reference (src/utils.cpp:5-32)."""
    a = np.atleast_1d(np.asarray(x))
    if a.size != 1:
        raise SarlaccError("%s should be %s" % (what, kind))
    return a[0]


def _numeric(x, what):
    return float(_scalar(x, what, "a numeric scalar"))


def _integer(x, what):
    return int(_scalar(x, what, "an integer scalar"))


def _string(x, what):
    if isinstance(x, (str, bytes)):
        x = [x]
    if len(x) != 1:
        raise SarlaccError("%s should be a string" % what)
    s = x[0]
    return s.encode() if isinstance(s, str) else bytes(s)


def _seq_qual(seq, qual):
    s = StringSet.from_strings(seq)
    q = StringSet.from_strings(qual)
    if len(s) != len(q):
        raise SarlaccError("sequence and quality vectors should have the same length")
    return s, q


# ---------------------------------------------------------------------------
def adaptor_align(readseq, readqual, encoding, gapopen, gapext, adaptor, sec_starts, sec_ends):
    """.Call adaptor_align (src/adaptor_align.cpp:11-77).
    Returns [scores, starts, ends, [section starts...], [section widths...]]."""
    ad = _string(adaptor, "adaptor sequence")
    go = _numeric(gapopen, "gap opening penalty")
    ge = _numeric(gapext, "gap extension penalty")
    s, q = _seq_qual(readseq, readqual)
    enc = as_encoding(encoding)
    ss = np.ascontiguousarray(sec_starts, dtype=np.int32).reshape(-1)
    se = np.ascontiguousarray(sec_ends, dtype=np.int32).reshape(-1)
    if ss.size != se.size:
        raise SarlaccError("section starts and ends should have the same length")
    n, ns = len(s), ss.size
    scores = np.zeros(n, np.float64)
    starts = np.zeros(n, np.int32)
    ends = np.zeros(n, np.int32)
    so = np.zeros((max(ns, 1), max(n, 1)), np.int32)
    sw = np.zeros((max(ns, 1), max(n, 1)), np.int32)
    if ns == 0:
        ss = np.zeros(1, np.int32)
        se = np.zeros(1, np.int32)
    check(_lib.lib().sarlacc_adaptor_align(
        ptr(s.chars), ptr(s.off), ptr(q.chars), ptr(q.off), C.c_int64(n),
        ptr(enc.errors), enc.names, len(enc), C.c_double(go), C.c_double(ge),
        ad, len(ad), ptr(ss), ptr(se), ns,
        ptr(scores), ptr(starts), ptr(ends), ptr(so), ptr(sw)))
    sol = [so.reshape(-1)[k * n:(k + 1) * n].copy() for k in range(ns)]
    swl = [sw.reshape(-1)[k * n:(k + 1) * n].copy() for k in range(ns)]
    return [scores, starts, ends, sol, swl]


def _scores_call(fn, seq, qual, encoding, gapopen, gapext, ref, what):
    rf = _string(ref, what)
    go = _numeric(gapopen, "gap opening penalty")
    ge = _numeric(gapext, "gap extension penalty")
    s, q = _seq_qual(seq, qual)
    enc = as_encoding(encoding)
    n = len(s)
    scores = np.zeros(n, np.float64)
    check(fn(ptr(s.chars), ptr(s.off), ptr(q.chars), ptr(q.off), C.c_int64(n),
             ptr(enc.errors), enc.names, len(enc), C.c_double(go), C.c_double(ge),
             rf, len(rf), ptr(scores)))
    return scores


def adaptor_align_score_only(readseq, readqual, encoding, gapopen, gapext, adaptor):
    """.Call adaptor_align_score_only (src/adaptor_align.cpp:79-110)."""
    return _scores_call(_lib.lib().sarlacc_adaptor_align_score_only, readseq, readqual, encoding,
                        gapopen, gapext, adaptor, "adaptor sequence")


def barcode_align(barcodeseq, barcodequal, encoding, gapopen, gapext, reference):
    """.Call barcode_align (src/barcode_align.cpp:10-44)."""
    return _scores_call(_lib.lib().sarlacc_barcode_align, barcodeseq, barcodequal, encoding,
                        gapopen, gapext, reference, "barcode sequence")


def general_align(inputseq, inputqual, encoding, gapopen, gapext, reference, edit_only):
    """.Call general_align (src/general_align.cpp:10-62).
    Returns [scores, edit distances, reference strings, query strings]."""
    rf = _string(reference, "reference sequence")
    go = _numeric(gapopen, "gap opening penalty")
    ge = _numeric(gapext, "gap extension penalty")
    s, q = _seq_qual(inputseq, inputqual)
    only = bool(_scalar(edit_only, "edit-only specification", "a logical scalar"))
    enc = as_encoding(encoding)
    n = len(s)
    scores = np.zeros(n, np.float64)
    edits = np.zeros(n, np.int32)
    cap = s.total + n * len(rf) + 1
    ar = np.zeros(1 if only else cap, np.uint8)
    aq = np.zeros(1 if only else cap, np.uint8)
    ao = np.zeros(n + 1, np.int64)
    check(_lib.lib().sarlacc_general_align(
        ptr(s.chars), ptr(s.off), ptr(q.chars), ptr(q.off), C.c_int64(n),
        ptr(enc.errors), enc.names, len(enc), C.c_double(go), C.c_double(ge),
        rf, len(rf), int(only), ptr(scores), ptr(edits), ptr(ar), ptr(aq), ptr(ao), C.c_int64(cap)))
    if only:
        return [scores, edits, [], []]
    return [scores, edits, StringSet(ar, ao).to_strings(), StringSet(aq, ao).to_strings()]


# ---------------------------------------------------------------------------
def _aln_list(alignments):
    """list of alignments (each a list of equal-width strings) -> flat rows + row ranges"""
    rows = []
    grp = np.zeros(len(alignments) + 1, dtype=np.int64)
    for k, a in enumerate(alignments):
        a = a.to_strings() if isinstance(a, StringSet) else list(a)
        rows.extend(a)
        grp[k + 1] = grp[k] + len(a)
    return StringSet.from_strings(rows), grp


def _consensus(alignments, min_cov, pseudo, qualities, encoding, want_lerr):
    s, grp = _aln_list(alignments)
    ng = len(alignments)
    cap = max(s.total, 1)
    cons = np.zeros(cap, np.uint8)
    phred = np.zeros(cap, np.uint8)
    coff = np.zeros(ng + 1, np.int64)
    lerr = np.zeros(cap, np.float64) if want_lerr else None
    if qualities is None:
        check(_lib.lib().sarlacc_create_consensus_basic_loop(
            ptr(s.chars), ptr(s.off), ptr(grp), C.c_int64(ng), C.c_double(min_cov), C.c_double(pseudo),
            ptr(cons), ptr(phred), ptr(coff), ptr(lerr)))
    else:
        q, qgrp = _aln_list(qualities)
        enc = as_encoding(encoding)
        check(_lib.lib().sarlacc_create_consensus_quality_loop(
            ptr(s.chars), ptr(s.off), ptr(grp), C.c_int64(ng), ptr(q.chars), ptr(q.off), ptr(qgrp),
            C.c_double(min_cov), ptr(enc.errors), enc.names, len(enc),
            ptr(cons), ptr(phred), ptr(coff), ptr(lerr)))
    cs = StringSet(cons, coff).to_strings()
    ps = StringSet(phred, coff).to_strings()
    return cs, ps, (lerr, coff)


def create_consensus_basic(alignments, min_cov, pseudo_count):
    """.Call create_consensus_basic (src/create_consensus.cpp:137-148): [consensus, log errors]."""
    mc = _numeric(min_cov, "minimum coverage")
    pc = _numeric(pseudo_count, "pseudo count")
    cs, _, (lerr, coff) = _consensus([alignments], mc, pc, None, None, True)
    return [cs[0], lerr[:coff[1]].copy()]


def create_consensus_basic_loop(alignments, min_cov, pseudo_count):
    """.Call create_consensus_basic_loop (src/create_consensus.cpp:150-170): [consensus strings, Phred strings]."""
    mc = _numeric(min_cov, "minimum coverage")
    pc = _numeric(pseudo_count, "pseudo count")
    cs, ps, _ = _consensus(alignments, mc, pc, None, None, False)
    return [cs, ps]


def create_consensus_quality(alignments, min_cov, qualities, encoding):
    """.Call create_consensus_quality (src/create_consensus.cpp:274-285)."""
    mc = _numeric(min_cov, "minimum coverage")
    cs, _, (lerr, coff) = _consensus([alignments], mc, 0.0, [qualities], encoding, True)
    return [cs[0], lerr[:coff[1]].copy()]


def create_consensus_quality_loop(alignments, min_cov, qualities, encoding):
    """.Call create_consensus_quality_loop (src/create_consensus.cpp:287-308)."""
    mc = _numeric(min_cov, "minimum coverage")
    if len(qualities) != len(alignments):
        raise SarlaccError("sarlacc_amd: alignments and qualities lists differ in length")
    cs, ps, _ = _consensus(alignments, mc, 0.0, qualities, encoding, False)
    return [cs, ps]


# ---------------------------------------------------------------------------
def mask_bad_bases(sequences, qualities, encoding, threshold):
    """.Call mask_bad_bases (src/mask_bad_bases.cpp:10-52): base -> 'N' where error > threshold."""
    s, q = _seq_qual(sequences, qualities)
    enc = as_encoding(encoding)
    thr = _numeric(threshold, "quality threshold")
    out = np.zeros(max(s.total, 1), np.uint8)
    check(_lib.lib().sarlacc_mask_bad_bases(
        ptr(s.chars), ptr(s.off), ptr(q.chars), ptr(q.off), C.c_int64(len(s)),
        ptr(enc.errors), enc.names, len(enc), C.c_double(thr), ptr(out)))
    return StringSet(out, s.off.copy()).to_strings()


def unmask_alignment(alignments, originals):
    """.Call unmask_alignment (src/unmask_alignment.cpp:12-59): alignment rows with the masked
    bases ('N'/'n') restored from the original sequences."""
    a = StringSet.from_strings(alignments)
    o = StringSet.from_strings(originals)
    out = np.zeros(max(a.total, 1), np.uint8)
    check(_lib.lib().sarlacc_unmask_alignment(ptr(a.chars), ptr(a.off), C.c_int64(len(a)), ptr(o.chars), ptr(o.off),
                                              C.c_int64(len(o)), ptr(out)))
    return StringSet(out, a.off.copy()).to_strings()


def compute_lev_masked(sequences):
    """.Call compute_lev_masked (src/compute_lev_masked.cpp:13-64): lower triangle, R 'dist' order."""
    s = StringSet.from_strings(sequences)
    n = len(s)
    out = np.zeros(max(n * (n - 1) // 2, 1), np.float64)
    check(_lib.lib().sarlacc_compute_lev_masked(ptr(s.chars), ptr(s.off), C.c_int64(n), ptr(out)))
    return out[: n * (n - 1) // 2]


def fast_levdist_test(sequences, limit, sorted=True):
    """.Call fast_levdist_test (src/sorted_trie.cpp:304-337): per sequence the 1-based
    indices of everything within `limit`, in the reference's trie order.  `sorted` only
    changes the processing order inside the reference and never its output."""
    s = StringSet.from_strings(sequences)
    lim = _integer(limit, "limit")
    _scalar(sorted, "sort specification", "a logical scalar")
    n = len(s)
    off = np.zeros(n + 1, np.int64)
    need = C.c_int64(0)
    cap = max(32 * n, 1024)
    while True:
        nbr = np.zeros(cap, np.int32)
        check(_lib.lib().sarlacc_fast_levdist_test(ptr(s.chars), ptr(s.off), C.c_int64(n), lim,
                                                   ptr(off), ptr(nbr), C.c_int64(cap), C.byref(need)))
        if need.value <= cap:
            break
        cap = need.value
    return lists_from_csr(off, nbr)


def cluster_umis_test(links):
    """.Call cluster_umis_test (src/cluster_umis_test.cpp:8-30): list of 1-based link vectors
    -> list of 1-based clusters in the reference's output order."""
    off, vals = csr_from_lists(links)
    n = len(links)
    ncl = C.c_int64(0)
    co = np.zeros(n + 2, np.int64)
    cl = np.zeros(max(n, 1), np.int32)
    check(_lib.lib().sarlacc_cluster_umis_test(ptr(off), ptr(vals), C.c_int64(n), C.byref(ncl), ptr(co), ptr(cl)))
    return lists_from_csr(co, cl, ncl.value)


def umi_group(umi1, thresh1, umi2, thresh2, pregroup):
    """.Call umi_group (src/umi_group.cpp:14-116) + the unlist(recursive=FALSE) of
    R/umiGroup.R:22: flattened list of clusters of 1-based read ids."""
    s1 = StringSet.from_strings(umi1)
    t1 = _integer(thresh1, "threshold 1")
    s2 = None
    if umi2 is not None:
        s2 = StringSet.from_strings(umi2)
        if len(s2) != len(s1):
            raise SarlaccError("'umi1' and 'umi2' should have the same length")
    t2 = _integer(thresh2, "threshold 2")
    goff, gvals = csr_from_lists(pregroup)
    total = int(goff[-1])
    ncl = C.c_int64(0)
    co = np.zeros(total + 2, np.int64)
    cl = np.zeros(max(total, 1), np.int32)
    check(_lib.lib().sarlacc_umi_group(
        ptr(s1.chars), ptr(s1.off), ptr(s2.chars) if s2 is not None else None,
        ptr(s2.off) if s2 is not None else None, C.c_int64(len(s1)), t1, t2,
        ptr(goff), ptr(gvals), C.c_int64(len(pregroup)), C.byref(ncl), ptr(co), ptr(cl)))
    return lists_from_csr(co, cl, ncl.value)


def umi_group_flat(umi1, thresh1, umi2, thresh2, pregroup_off, pregroup):
    """umi_group on CSR pre-groups, CSR clusters out: (cluster_off int64[nclusters+1], members
    int32[...]) -- same clusters, same order as umi_group, without building Python lists."""
    s1 = StringSet.from_strings(umi1)
    s2 = StringSet.from_strings(umi2) if umi2 is not None else None
    if s2 is not None and len(s2) != len(s1):
        raise SarlaccError("'umi1' and 'umi2' should have the same length")
    goff = np.ascontiguousarray(pregroup_off, dtype=np.int64)
    gvals = np.ascontiguousarray(pregroup, dtype=np.int32)
    if gvals.size == 0:
        gvals = np.zeros(1, np.int32)
    total = int(goff[-1])
    ncl = C.c_int64(0)
    co = np.zeros(total + 2, np.int64)
    cl = np.zeros(max(total, 1), np.int32)
    check(_lib.lib().sarlacc_umi_group(
        ptr(s1.chars), ptr(s1.off), ptr(s2.chars) if s2 is not None else None,
        ptr(s2.off) if s2 is not None else None, C.c_int64(len(s1)), _integer(thresh1, "threshold 1"),
        _integer(thresh2, "threshold 2"), ptr(goff), ptr(gvals), C.c_int64(goff.size - 1), C.byref(ncl), ptr(co), ptr(cl)))
    return co[:ncl.value + 1], cl[:int(co[ncl.value])]


def csr_select(off, vals, keep):
    """Rows of a CSR list selected by the boolean mask `keep` (numpy only)."""
    sizes = np.diff(off)
    noff = np.zeros(int(keep.sum()) + 1, np.int64)
    np.cumsum(sizes[keep], out=noff[1:])
    return noff, vals[np.repeat(keep, sizes)]


def set_msa_spec(spec):
    """sarlacc_set_msa_spec: 2 = consistency-based progressive alignment (default), 1 = centre-star, 0 = default."""
    check(_lib.lib().sarlacc_set_msa_spec(int(spec)))


def set_option(name, value):
    """sarlacc_set_option: the A/B switches of the tests and perf tools (include/sarlacc_amd.h lists them); 0 = product path."""
    check(_lib.lib().sarlacc_set_option(str(name).encode(), int(value)))


def quick_msa(groupings, sequences, match, mismatch, gapExtension, gapOpening, bandwidth):
    """.Call quick_msa (src/quick_msa.cpp:15-80), same argument order (the R caller passes
    -gapOpening as gapExtension and -gapExtension as gapOpening, R/multiReadAlign.R:47).
    Returns one list of equal-width gapped strings per group."""
    s = StringSet.from_strings(sequences)
    ma = _numeric(match, "match score")
    mm = _numeric(mismatch, "mismatch score")
    gx = _numeric(gapExtension, "gap extension score")
    go = _numeric(gapOpening, "gap opening score")
    bw = _integer(bandwidth, "bandwidth")
    goff, gvals = csr_from_lists(groupings)
    ng = len(groupings)
    width = np.zeros(max(ng, 1), np.int32)
    ooff = np.zeros(ng + 1, np.int64)
    args = (ptr(goff), ptr(gvals), C.c_int64(ng), ptr(s.chars), ptr(s.off), C.c_int64(len(s)),
            C.c_double(ma), C.c_double(mm), C.c_double(gx), C.c_double(go), bw, ptr(width), ptr(ooff))
    # generous first guess: every read padded to twice the longest member
    sizes = np.diff(goff)
    cap = int(s.total * 2 + 64) if ng else 1
    for attempt in range(2):
        out = np.zeros(max(cap, 1), np.uint8)
        try:
            check(_lib.lib().sarlacc_quick_msa(*args, ptr(out), C.c_int64(cap)))
            break
        except SarlaccError as e:
            if attempt == 0 and "buffer too small" in str(e):
                cap = int(ooff[ng])
                continue
            raise
    res = []
    for g in range(ng):
        w, m = int(width[g]), int(sizes[g])
        blk = out[ooff[g]:ooff[g + 1]].tobytes()
        res.append([blk[r * w:(r + 1) * w].decode() for r in range(m)])
    return res


# ---------------------------------------------------------------------------
# Flat variants (numpy in, numpy out) for large batches: same C ABI calls without
# materialising Python string lists.

def quick_msa_flat(grp_off, grp, seqs, match, mismatch, gapExtension, gapOpening, bandwidth):
    """quick_msa on CSR groups.  Returns (rows StringSet, grp_rows int64[ngroups+1], width int32[ngroups]):
    rows are the gapped strings of all groups in order, grp_rows the row range of each group."""
    s = StringSet.from_strings(seqs)
    goff = np.ascontiguousarray(grp_off, dtype=np.int64)
    gvals = np.ascontiguousarray(grp, dtype=np.int32)
    if gvals.size == 0:
        gvals = np.zeros(1, np.int32)
    ng = goff.size - 1
    width = np.zeros(max(ng, 1), np.int32)
    ooff = np.zeros(ng + 1, np.int64)
    args = (ptr(goff), ptr(gvals), C.c_int64(ng), ptr(s.chars), ptr(s.off), C.c_int64(len(s)),
            C.c_double(match), C.c_double(mismatch), C.c_double(gapExtension), C.c_double(gapOpening), int(bandwidth),
            ptr(width), ptr(ooff))
    # one pass in the common case: rows are rarely more than 1.5x the reads they hold; the
    # library reports the exact size (out_off) when the guess was too small
    cap = int(1.5 * s.widths()[gvals[:int(goff[-1])].astype(np.int64) - 1].sum()) + 1024 if goff[-1] else 1
    for attempt in range(2):
        out = np.zeros(max(cap, 1), np.uint8)
        try:
            check(_lib.lib().sarlacc_quick_msa(*args, ptr(out), C.c_int64(cap)))
            break
        except SarlaccError as e:
            if attempt == 0 and "buffer too small" in str(e):
                cap = int(ooff[ng])
                continue
            raise
    sizes = np.diff(goff)
    grp_rows = np.zeros(ng + 1, np.int64)
    np.cumsum(sizes, out=grp_rows[1:])
    row_w = np.repeat(width[:ng].astype(np.int64), sizes)
    row_off = np.zeros(row_w.size + 1, np.int64)
    np.cumsum(row_w, out=row_off[1:])
    return StringSet(out, row_off), grp_rows, width[:ng]


def create_consensus_flat(rows, grp_rows, min_cov, pseudo_count=1.0, quals=None, qgrp_rows=None, encoding=None):
    """Consensus over alignments given as (rows StringSet, grp_rows).  Returns (consensus StringSet,
    phred StringSet).  With `quals` (ungapped quality strings per row) the quality-weighted vote runs."""
    ng = len(grp_rows) - 1
    grp_rows = np.ascontiguousarray(grp_rows, dtype=np.int64)
    cap = max(rows.total, 1)
    cons = np.zeros(cap, np.uint8)
    phred = np.zeros(cap, np.uint8)
    coff = np.zeros(ng + 1, np.int64)
    if quals is None:
        check(_lib.lib().sarlacc_create_consensus_basic_loop(
            ptr(rows.chars), ptr(rows.off), ptr(grp_rows), C.c_int64(ng), C.c_double(min_cov), C.c_double(pseudo_count),
            ptr(cons), ptr(phred), ptr(coff), None))
    else:
        enc = as_encoding(encoding)
        q = StringSet.from_strings(quals)
        qg = grp_rows if qgrp_rows is None else np.ascontiguousarray(qgrp_rows, dtype=np.int64)
        check(_lib.lib().sarlacc_create_consensus_quality_loop(
            ptr(rows.chars), ptr(rows.off), ptr(grp_rows), C.c_int64(ng), ptr(q.chars), ptr(q.off), ptr(qg),
            C.c_double(min_cov), ptr(enc.errors), enc.names, len(enc), ptr(cons), ptr(phred), ptr(coff), None))
    return StringSet(cons, coff.copy()), StringSet(phred, coff.copy())


def msa_consensus_flat(grp_off, grp, seqs, match, mismatch, gapExtension, gapOpening, bandwidth, min_cov,
                       pseudo_count=1.0, quals=None, encoding=None):
    """quick_msa_flat followed by create_consensus_flat in one native call (sarlacc_msa_consensus):
    the gapped rows stay in HBM and `quals` are the quality strings of ALL reads, in read order.
    Returns (consensus StringSet, phred StringSet), one entry per group."""
    import re
    s = StringSet.from_strings(seqs)
    goff = np.ascontiguousarray(grp_off, dtype=np.int64)
    gvals = np.ascontiguousarray(grp, dtype=np.int32)
    if gvals.size == 0:
        gvals = np.zeros(1, np.int32)
    ng = goff.size - 1
    q = enc = None
    if quals is not None:
        q = StringSet.from_strings(quals)
        if len(q) != len(s):
            raise SarlaccError("sequence and quality vectors should have the same length")
        enc = as_encoding(encoding)
    coff = np.zeros(ng + 1, np.int64)
    w = s.widths()
    sizes = np.diff(goff)
    longest = np.maximum.reduceat(w[gvals[:int(goff[-1])].astype(np.int64) - 1], goff[:-1][sizes > 0]) if goff[-1] else np.zeros(0)
    cap = int(1.5 * longest.sum()) + 1024
    for attempt in range(2):
        cons = np.zeros(cap, np.uint8)
        phred = np.zeros(cap, np.uint8)
        try:
            check(_lib.lib().sarlacc_msa_consensus(
                ptr(goff), ptr(gvals), C.c_int64(ng), ptr(s.chars), ptr(s.off),
                ptr(q.chars) if q is not None else None, ptr(q.off) if q is not None else None, C.c_int64(len(s)),
                C.c_double(match), C.c_double(mismatch), C.c_double(gapExtension), C.c_double(gapOpening), int(bandwidth),
                C.c_double(min_cov), C.c_double(pseudo_count), ptr(enc.errors) if enc is not None else None,
                enc.names if enc is not None else None, len(enc) if enc is not None else 0,
                ptr(cons), ptr(phred), ptr(coff), C.c_int64(cap)))
            break
        except SarlaccError as e:
            m = re.search(r"buffer too small \((\d+) needed\)", str(e))
            if attempt == 0 and m:
                cap = int(m.group(1)) + 16
                continue
            raise
    return StringSet(cons, coff.copy()), StringSet(phred, coff.copy())


def umi_pairs_shard(umi, limit, shard_index, shard_count):
    """sarlacc_umi_pairs_shard: neighbour pairs (rank_i << 32 | rank_j) found in this shard's row
    tiles of the all-pairs matrix of one pre-group (ranks = positions in the trie order)."""
    s = StringSet.from_strings(umi)
    lim = _integer(limit, "limit")
    need = C.c_int64(0)
    cap = max(16 * len(s) // max(shard_count, 1), 1024)
    while True:
        pairs = np.zeros(cap, np.uint64)
        check(_lib.lib().sarlacc_umi_pairs_shard(ptr(s.chars), ptr(s.off), C.c_int64(len(s)), lim, int(shard_index),
                                                 int(shard_count), ptr(pairs), C.c_int64(cap), C.byref(need)))
        if need.value <= cap:
            return pairs[:need.value].copy()
        cap = need.value


def umi_group_from_pairs(umi, limit, pairs, flat=False):
    """sarlacc_umi_group_from_pairs: umi_group of a single pre-group given its neighbour pairs.
    flat=True returns the clusters as CSR (offsets, members) instead of a list of arrays."""
    s = StringSet.from_strings(umi)
    lim = _integer(limit, "limit")
    pairs = np.ascontiguousarray(pairs, dtype=np.uint64)
    n = len(s)
    ncl = C.c_int64(0)
    co = np.zeros(n + 2, np.int64)
    cl = np.zeros(max(n, 1), np.int32)
    pp = pairs if pairs.size else np.zeros(1, np.uint64)
    check(_lib.lib().sarlacc_umi_group_from_pairs(ptr(s.chars), ptr(s.off), C.c_int64(n), lim, ptr(pp), C.c_int64(pairs.size),
                                                  C.byref(ncl), ptr(co), ptr(cl)))
    if flat:
        return co[:ncl.value + 1], cl[:int(co[ncl.value])]
    return lists_from_csr(co, cl, ncl.value)


def dev_umi_pairs_shard(umi, limit, shard_index, shard_count):
    """sarlacc_dev_umi_pairs_shard: the shard's neighbour pairs stay in the library's workspace on the device; returns
    their number.  dev_umi_pairs_fetch must be the next library call."""
    s = StringSet.from_strings(umi)
    need = C.c_int64(0)
    check(_lib.lib().sarlacc_dev_umi_pairs_shard(ptr(s.chars), ptr(s.off), C.c_int64(len(s)), _integer(limit, "limit"), int(shard_index),
                                                 int(shard_count), C.byref(need)))
    return int(need.value)


def dev_umi_pairs_fetch(d_pairs, cap):
    """sarlacc_dev_umi_pairs_fetch: copies the pairs of the shard search just run into device memory (a torch tensor of
    8-byte elements or a raw address) holding at least `cap` entries."""
    addr = d_pairs.data_ptr() if hasattr(d_pairs, "data_ptr") else int(d_pairs)
    check(_lib.lib().sarlacc_dev_umi_pairs_fetch(C.c_void_p(addr), C.c_int64(int(cap))))


def dev_umi_group_from_pairs(umi, limit, d_pairs, npairs, flat=False):
    """sarlacc_dev_umi_group_from_pairs: umi_group of a single pre-group from neighbour pairs held in device memory."""
    s = StringSet.from_strings(umi)
    n = len(s)
    ncl = C.c_int64(0)
    co = np.zeros(n + 2, np.int64)
    cl = np.zeros(max(n, 1), np.int32)
    addr = d_pairs.data_ptr() if hasattr(d_pairs, "data_ptr") else int(d_pairs or 0)
    check(_lib.lib().sarlacc_dev_umi_group_from_pairs(ptr(s.chars), ptr(s.off), C.c_int64(n), _integer(limit, "limit"), C.c_void_p(addr),
                                                      C.c_int64(int(npairs)), C.byref(ncl), ptr(co), ptr(cl)))
    if flat:
        return co[:ncl.value + 1], cl[:int(co[ncl.value])]
    return lists_from_csr(co, cl, ncl.value)


# ---------------------------------------------------------------------------
# alignment profiling
def find_homopolymers(sequences):
    """.Call find_homopolymers (src/homopolymer.cpp:87-134): [index (0-based), position (1-based, ungapped),
    size, base] of every run of two or more equal bases."""
    s = StringSet.from_strings(sequences)
    n = len(s)
    cnt = C.c_int64(0)
    cap = max(64, s.total // 8)
    while True:
        idx, pos, size = (np.zeros(cap, np.int32) for _ in range(3))
        base = np.zeros(cap, np.uint8)
        check(_lib.lib().sarlacc_find_homopolymers(ptr(s.chars), ptr(s.off), C.c_int64(n), ptr(idx), ptr(pos), ptr(size), ptr(base),
                                                   C.c_int64(cap), C.byref(cnt)))
        if cnt.value <= cap:
            break
        cap = cnt.value
    k = cnt.value
    return [idx[:k], pos[:k], size[:k], [chr(c) for c in base[:k]]]


def match_homopolymers(ref_align, read_align):
    """.Call match_homopolymers (src/homopolymer.cpp:141-209): [alignment (0-based), position of the reference
    homopolymer, longest overlapping run of the same base in the read]."""
    r, q = StringSet.from_strings(ref_align), StringSet.from_strings(read_align)
    cnt = C.c_int64(0)
    cap = max(64, r.total // 8)
    while True:
        idx, pos, rlen = (np.zeros(cap, np.int32) for _ in range(3))
        check(_lib.lib().sarlacc_match_homopolymers(ptr(r.chars), ptr(r.off), C.c_int64(len(r)), ptr(q.chars), ptr(q.off), C.c_int64(len(q)),
                                                    ptr(idx), ptr(pos), ptr(rlen), C.c_int64(cap), C.byref(cnt)))
        if cnt.value <= cap:
            break
        cap = cnt.value
    k = cnt.value
    return [idx[:k], pos[:k], rlen[:k]]


def find_errors(ref_align, read_align):
    """.Call find_errors (src/find_errors.cpp:9-121): [bases, to A, to C, to G, to T, deletions, insertion
    positions (0-based, position of the next reference base), insertion lengths]."""
    r, q = StringSet.from_strings(ref_align), StringSet.from_strings(read_align)
    cap_b = int(r.off[1] - r.off[0]) if len(r) else 0
    cap_i = max(64, r.total // 16)
    sl, ni = C.c_int64(0), C.c_int64(0)
    while True:
        bases = np.zeros(max(cap_b, 1), np.uint8)
        cols = [np.zeros(max(cap_b, 1), np.int32) for _ in range(5)]
        ip, il = np.zeros(cap_i, np.int32), np.zeros(cap_i, np.int32)
        check(_lib.lib().sarlacc_find_errors(ptr(r.chars), ptr(r.off), C.c_int64(len(r)), ptr(q.chars), ptr(q.off), C.c_int64(len(q)),
                                             C.byref(sl), ptr(bases), *[ptr(c) for c in cols], C.c_int64(cap_b), ptr(ip), ptr(il),
                                             C.c_int64(cap_i), C.byref(ni)))
        if ni.value <= cap_i:
            break
        cap_i = ni.value
    n = sl.value
    return ["".join(chr(c) for c in bases[:n])] + [c[:n] for c in cols] + [ip[:ni.value], il[:ni.value]]
