"""ctypes front-end for oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the sarlacc_amd package (see oracle.h).
The function names follow the reference's .Call entry points
(/root/reference/src/sarlacc.h:14-36) so parity tests can call oracle and
product side by side.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleError(RuntimeError):
    pass


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_last_error.restype = C.c_char_p
    return _LIB


def _check(rc):
    if rc:
        raise OracleError(lib().orc_last_error().decode())


def pack(strings):
    """list of str/bytes -> (uint8 buffer, int64 offsets[n+1])"""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return buf, off


def unpack(buf, off):
    raw = buf.tobytes()
    return [raw[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def phred_encoding():
    """Counterpart of sarlacc:::.create_encoding_vector for PhredQuality
    (/root/reference/R/qualityMask.R:19-28): names '!'..'~', errors 10^(-q/10)."""
    q = np.arange(94, dtype=np.float64)
    return np.power(10.0, -q / 10.0), bytes(range(33, 127))


def _enc(encoding):
    errors, names = encoding
    errors = np.ascontiguousarray(errors, dtype=np.float64)
    names = names.encode() if isinstance(names, str) else bytes(names)
    return errors, names, len(names)


def cost_tables(errors):
    errors = np.ascontiguousarray(errors, dtype=np.float64)
    n = errors.size
    m = np.zeros((4, n))
    mm = np.zeros((4, n))
    _check(lib().orc_cost_tables(_p(errors), n, _p(m), _p(mm)))
    return m, mm


def align_one(ref, seq, qual, encoding, gapopen, gapext, local=True, want_dirs=False):
    errors, names, nenc = _enc(encoding)
    ref_b, seq_b, qual_b = ref.encode(), seq.encode(), qual.encode()
    score = C.c_double()
    L, R = len(seq_b), len(ref_b)
    dirs = np.zeros((R + 1, L + 1), dtype=np.int32) if want_dirs else None
    _check(lib().orc_align_one(ref_b, R, seq_b, qual_b, L, _p(errors), names, nenc,
                               C.c_double(gapopen), C.c_double(gapext), int(local),
                               C.byref(score), _p(dirs) if want_dirs else None))
    return (score.value, dirs) if want_dirs else score.value


def adaptor_align(readseq, readqual, encoding, gapopen, gapext, adaptor, sec_starts=(), sec_ends=()):
    """-> (scores, starts, ends, [sec_start...], [sec_width...]) like the .Call list."""
    errors, names, nenc = _enc(encoding)
    sb, so = pack(readseq)
    qb, qo = pack(readqual)
    if len(so) != len(qo):
        raise OracleError("sequence and quality vectors should have the same length")
    n = len(so) - 1
    ss = np.ascontiguousarray(sec_starts, dtype=np.int32)
    se = np.ascontiguousarray(sec_ends, dtype=np.int32)
    if ss.size != se.size:
        raise OracleError("section starts and ends should have the same length")
    ns = ss.size
    scores = np.zeros(n)
    starts = np.zeros(n, np.int32)
    ends = np.zeros(n, np.int32)
    so_ = np.zeros((max(ns, 1), max(n, 1)), np.int32)
    sw_ = np.zeros((max(ns, 1), max(n, 1)), np.int32)
    ad = adaptor.encode()
    _check(lib().orc_adaptor_align(_p(sb), _p(so), _p(qb), _p(qo), C.c_int64(n), _p(errors), names, nenc,
                                   C.c_double(gapopen), C.c_double(gapext), ad, len(ad),
                                   _p(ss), _p(se), ns, _p(scores), _p(starts), _p(ends),
                                   _p(so_), _p(sw_)))
    return scores, starts, ends, [so_[k, :n].copy() for k in range(ns)], [sw_[k, :n].copy() for k in range(ns)]


def _align_scores(readseq, readqual, encoding, gapopen, gapext, ref, local):
    errors, names, nenc = _enc(encoding)
    sb, so = pack(readseq)
    qb, qo = pack(readqual)
    if len(so) != len(qo):
        raise OracleError("sequence and quality vectors should have the same length")
    n = len(so) - 1
    scores = np.zeros(n)
    rb = ref.encode()
    _check(lib().orc_align_scores(_p(sb), _p(so), _p(qb), _p(qo), C.c_int64(n), _p(errors), names, nenc,
                                  C.c_double(gapopen), C.c_double(gapext), rb, len(rb), int(local), _p(scores)))
    return scores


def adaptor_align_score_only(readseq, readqual, encoding, gapopen, gapext, adaptor):
    return _align_scores(readseq, readqual, encoding, gapopen, gapext, adaptor, True)


def barcode_align(seq, qual, encoding, gapopen, gapext, reference):
    return _align_scores(seq, qual, encoding, gapopen, gapext, reference, False)


def general_align(seq, qual, encoding, gapopen, gapext, reference, edit_only=False):
    errors, names, nenc = _enc(encoding)
    sb, so = pack(seq)
    qb, qo = pack(qual)
    if len(so) != len(qo):
        raise OracleError("sequence and quality vectors should have the same length")
    n = len(so) - 1
    rb = reference.encode()
    scores = np.zeros(n)
    edits = np.zeros(n, np.int32)
    cap = int(so[-1]) + n * len(rb) + 1
    ar = np.zeros(cap, np.uint8)
    aq = np.zeros(cap, np.uint8)
    ao = np.zeros(n + 1, np.int64)
    _check(lib().orc_general_align(_p(sb), _p(so), _p(qb), _p(qo), C.c_int64(n), _p(errors), names, nenc,
                                   C.c_double(gapopen), C.c_double(gapext), rb, len(rb),
                                   _p(scores), _p(edits),
                                   None if edit_only else _p(ar), None if edit_only else _p(aq),
                                   _p(ao), C.c_int64(cap)))
    if edit_only:
        return scores, edits, [], []
    return scores, edits, unpack(ar, ao), unpack(aq, ao)


def mask_bad_bases(seq, qual, encoding, threshold):
    errors, names, nenc = _enc(encoding)
    sb, so = pack(seq)
    qb, qo = pack(qual)
    if len(so) != len(qo):
        raise OracleError("sequence and quality vectors should have the same length")
    out = np.zeros_like(sb)
    _check(lib().orc_mask_bad_bases(_p(sb), _p(so), _p(qb), _p(qo), C.c_int64(len(so) - 1), _p(errors), names, nenc,
                                    C.c_double(threshold), _p(out)))
    return unpack(out, so)


def unmask_alignment(alignments, originals):
    ab, ao = pack(alignments)
    ob, oo = pack(originals)
    out = np.zeros_like(ab)
    _check(lib().orc_unmask_alignment(_p(ab), _p(ao), C.c_int64(len(ao) - 1), _p(ob), _p(oo), C.c_int64(len(oo) - 1), _p(out)))
    return unpack(out, ao)


def scramble(seq, qual, seed):
    sb, so = pack(seq)
    qb, _ = pack(qual)
    os_, oq = np.zeros_like(sb), np.zeros_like(qb)
    _check(lib().orc_scramble(_p(sb), _p(qb), _p(so), C.c_int64(len(so) - 1), C.c_uint64(seed), _p(os_), _p(oq)))
    return unpack(os_, so), unpack(oq, so)


def compute_lev_masked(seqs):
    sb, so = pack(seqs)
    n = len(so) - 1
    out = np.zeros(max(n * (n - 1) // 2, 1))
    _check(lib().orc_compute_lev_masked(_p(sb), _p(so), C.c_int64(n), _p(out)))
    return out[: n * (n - 1) // 2]


def fast_levdist_test(seqs, limit):
    """-> list of 1-based neighbour arrays in trie order (sorted flag has no effect on results)."""
    sb, so = pack(seqs)
    n = len(so) - 1
    off = np.zeros(n + 1, np.int64)
    cap = max(16 * n, 16)
    while True:
        nbr = np.zeros(cap, np.int32)
        need = C.c_int64()
        rc = lib().orc_fast_levdist(_p(sb), _p(so), C.c_int64(n), int(limit), _p(off), _p(nbr), C.c_int64(cap), C.byref(need))
        if rc == 2:
            cap = need.value
            continue
        _check(rc)
        break
    return [nbr[off[i]:off[i + 1]] + 1 for i in range(n)]


def _csr(lists, base):
    off = np.zeros(len(lists) + 1, np.int64)
    if lists:
        off[1:] = np.cumsum([len(x) for x in lists])
    flat = np.concatenate([np.asarray(x, dtype=np.int32) for x in lists]) - base if lists and off[-1] else np.zeros(1, np.int32)
    return off, np.ascontiguousarray(flat, dtype=np.int32)


def cluster_umis_test(links, fast=False):
    """links: list of 1-based integer vectors -> list of 1-based clusters."""
    off, flat = _csr(links, 1)
    n = len(links)
    ncl = C.c_int64()
    co = np.zeros(n + 2, np.int64)
    cl = np.zeros(max(n, 1), np.int32)
    f = lib().orc_cluster_umis_fast if fast else lib().orc_cluster_umis
    _check(f(_p(off), _p(flat), C.c_int64(n), C.byref(ncl), _p(co), _p(cl)))
    return [cl[co[i]:co[i + 1]] + 1 for i in range(ncl.value)]


def umi_group(umi1, thresh1, umi2, thresh2, pregroups, fast=False):
    """pregroups: list of 1-based index vectors -> flattened list of clusters (1-based),
    i.e. what R/umiGroup.R:21-22 returns."""
    b1, o1 = pack(umi1)
    n = len(o1) - 1
    if umi2 is not None:
        b2, o2 = pack(umi2)
        if len(o2) - 1 != n:
            raise OracleError("'umi1' and 'umi2' should have the same length")
    goff, gflat = _csr([np.asarray(g) for g in pregroups], 0)
    total = int(goff[-1])
    ncl = C.c_int64()
    co = np.zeros(total + 2, np.int64)
    cl = np.zeros(max(total, 1), np.int32)
    _check(lib().orc_umi_group(_p(b1), _p(o1), _p(b2) if umi2 is not None else None,
                               _p(o2) if umi2 is not None else None, C.c_int64(n), int(thresh1), int(thresh2),
                               _p(goff), _p(gflat), C.c_int64(len(pregroups)), int(fast),
                               C.byref(ncl), _p(co), _p(cl)))
    return [cl[co[i]:co[i + 1]].copy() for i in range(ncl.value)]


def create_consensus_basic(aln, min_cov, pseudo):
    ab, ao = pack(aln)
    n = len(ao) - 1
    W = int(ao[1] - ao[0]) if n else 0
    cons = np.zeros(W + 1, np.uint8)
    lerr = np.zeros(max(W, 1))
    k = C.c_int64()
    _check(lib().orc_consensus_basic(_p(ab), _p(ao), C.c_int64(n), C.c_double(min_cov), C.c_double(pseudo),
                                     _p(cons), _p(lerr), C.byref(k)))
    return cons[: k.value].tobytes().decode(), lerr[: k.value].copy()


def create_consensus_quality(aln, min_cov, quals, encoding):
    errors, names, nenc = _enc(encoding)
    ab, ao = pack(aln)
    qb, qo = pack(quals)
    n = len(ao) - 1
    W = int(ao[1] - ao[0]) if n else 0
    cons = np.zeros(W + 1, np.uint8)
    lerr = np.zeros(max(W, 1))
    k = C.c_int64()
    _check(lib().orc_consensus_quality(_p(ab), _p(ao), C.c_int64(n), _p(qb), _p(qo), C.c_int64(len(qo) - 1),
                                       C.c_double(min_cov), _p(errors), names, nenc,
                                       _p(cons), _p(lerr), C.byref(k)))
    return cons[: k.value].tobytes().decode(), lerr[: k.value].copy()


def errors_to_string(lerr):
    lerr = np.ascontiguousarray(lerr, dtype=np.float64)
    out = np.zeros(lerr.size + 1, np.uint8)
    lib().orc_errors_to_string(_p(lerr), C.c_int64(lerr.size), _p(out))
    return out[: lerr.size].tobytes().decode()


def create_consensus_basic_loop(aln_list, min_cov, pseudo):
    cons, quals = [], []
    for aln in aln_list:
        c, e = create_consensus_basic(aln, min_cov, pseudo)
        cons.append(c)
        quals.append(errors_to_string(e))
    return cons, quals


def create_consensus_quality_loop(aln_list, min_cov, qual_list, encoding):
    cons, quals = [], []
    for aln, q in zip(aln_list, qual_list):
        c, e = create_consensus_quality(aln, min_cov, q, encoding)
        cons.append(c)
        quals.append(errors_to_string(e))
    return cons, quals


def msa2_tree(reads, match, mismatch, gap_extension, gap_opening, bandwidth):
    """Guide tree and distances of spec v2 for one group: (joins int32[n-1, 2], dist float64[n, n])."""
    sb, so = pack(reads)
    n = len(reads)
    joins = np.zeros((max(n - 1, 1), 2), np.int32)
    dist = np.zeros((n, n), np.float64)
    _check(lib().orc_msa2_tree(_p(sb), _p(so), C.c_int64(n), int(match), int(mismatch), int(gap_opening),
                               int(gap_extension), int(bandwidth), _p(joins), _p(dist)))
    return joins[:max(n - 1, 0)], dist


MSA2_STAT_NAMES = ("joins", "rows", "rows_with_candidates", "rows_capped", "candidates_ignored_by_cap",
                   "entries_before_filter", "entries_filtered", "rows_filtered", "entries_kept", "triples",
                   "triples_2_positions", "triples_3_positions", "triples_gap_direct", "candidates",
                   "rows_multi_entry", "max_row_entries", "library_positions_ignored")


def msa2_set_rules(nocap=False, nofilter=False):
    """Switch spec v2's own rules (row cap of 16 partner columns, half-the-heaviest noise filter) off / on."""
    lib().orc_msa2_set_rules.restype = None
    lib().orc_msa2_set_rules(int(bool(nocap)), int(bool(nofilter)))


def msa2_set_library(others=3):
    """Partner positions the extended library keeps per (a, p, b) beside the direct one (spec v2: 3); -1: the unbounded
    library of rounds 2-4 (tools/msa2_rules.py compares them)."""
    lib().orc_msa2_set_library.restype = None
    lib().orc_msa2_set_library(int(others))


def msa2_stats(reset=True):
    """Counters of spec v2's library and rows since the last reset (names: MSA2_STAT_NAMES)."""
    buf = np.zeros(len(MSA2_STAT_NAMES), np.int64)
    lib().orc_msa2_stats(_p(buf), C.c_int64(buf.size), int(bool(reset)))
    return dict(zip(MSA2_STAT_NAMES, buf.tolist()))


def quick_msa(groupings, sequences, match, mismatch, gap_extension, gap_opening, bandwidth, spec=2, tcoffee_max=64, max_columns=65535):
    """Same argument order as the reference .Call (src/quick_msa.cpp:15): note that
    the R caller passes (-gapOpening, -gapExtension) into (gap_extension, gap_opening)
    (R/multiReadAlign.R:47, SURVEY App.B Q15).  spec 2 (default): consistency-based progressive
    alignment (msa2.c) for groups of up to `tcoffee_max` reads of at most 65 471 bases whose alignment has at most 65 535
    columns, spec 1 (centre-star, msa.c) beyond that and when spec == 1 -- the same policy as the product."""
    out = []
    for g in groupings:
        reads = [sequences[i - 1] for i in g]
        if len(reads) == 0:
            out.append([])
            continue
        sb, so = pack(reads)
        m = len(reads)
        cap = m * (int(so[-1]) + 16) + 16
        buf = np.zeros(cap, np.uint8)
        width = C.c_int64()
        longest = int(np.diff(so).max()) if m else 0
        v2 = spec == 2 and m <= tcoffee_max and longest + 64 <= 65535
        fn = lib().orc_msa2_group if v2 else lib().orc_msa_group
        _check(fn(_p(sb), _p(so), C.c_int64(m), int(match), int(mismatch), int(gap_opening),
                  int(gap_extension), int(bandwidth), _p(buf), C.c_int64(cap), C.byref(width)))
        if v2 and width.value > max_columns:   # (an alignment wider than 16-bit columns: spec v1, as the product decides after the fact)
            _check(lib().orc_msa_group(_p(sb), _p(so), C.c_int64(m), int(match), int(mismatch), int(gap_opening),
                                       int(gap_extension), int(bandwidth), _p(buf), C.c_int64(cap), C.byref(width)))
        W = width.value
        out.append([buf[r * W:(r + 1) * W].tobytes().decode() for r in range(m)])
    return out


# ---- alignment profiling (src/homopolymer.cpp, src/find_errors.cpp) ----
def find_homopolymers(seqs):
    """-> [index (0-based), position (1-based), size, base] as the reference's .Call returns them."""
    sb, so = pack(seqs)
    n = len(so) - 1
    cnt = C.c_int64(0)
    cap = 16
    while True:
        idx, pos, size = (np.zeros(cap, np.int32) for _ in range(3))
        base = np.zeros(cap, np.uint8)
        _check(lib().orc_find_homopolymers(_p(sb), _p(so), C.c_int64(n), _p(idx), _p(pos), _p(size), _p(base), C.c_int64(cap), C.byref(cnt)))
        if cnt.value <= cap:
            break
        cap = cnt.value
    k = cnt.value
    return [idx[:k], pos[:k], size[:k], [chr(c) for c in base[:k]]]


def match_homopolymers(refs, reads):
    rb, ro = pack(refs)
    qb, qo = pack(reads)
    cnt = C.c_int64(0)
    cap = 16
    while True:
        idx, pos, rlen = (np.zeros(cap, np.int32) for _ in range(3))
        _check(lib().orc_match_homopolymers(_p(rb), _p(ro), C.c_int64(len(ro) - 1), _p(qb), _p(qo), C.c_int64(len(qo) - 1),
                                            _p(idx), _p(pos), _p(rlen), C.c_int64(cap), C.byref(cnt)))
        if cnt.value <= cap:
            break
        cap = cnt.value
    k = cnt.value
    return [idx[:k], pos[:k], rlen[:k]]


def find_errors(refs, reads):
    """-> [bases, to A, to C, to G, to T, deletions, insertion positions (0-based), insertion lengths]"""
    rb, ro = pack(refs)
    qb, qo = pack(reads)
    cap_b = int(ro[1] - ro[0]) if len(ro) > 1 else 0
    cap_i = 16
    sl, ni = C.c_int64(0), C.c_int64(0)
    while True:
        bases = np.zeros(max(cap_b, 1), np.uint8)
        cols = [np.zeros(max(cap_b, 1), np.int32) for _ in range(5)]
        ip, il = np.zeros(cap_i, np.int32), np.zeros(cap_i, np.int32)
        rc = lib().orc_find_errors(_p(rb), _p(ro), C.c_int64(len(ro) - 1), _p(qb), _p(qo), C.c_int64(len(qo) - 1), C.byref(sl), _p(bases),
                                   *[_p(c) for c in cols], C.c_int64(cap_b), _p(ip), _p(il), C.c_int64(cap_i), C.byref(ni))
        if rc == 2:
            cap_b = sl.value
            continue
        _check(rc)
        if ni.value <= cap_i:
            break
        cap_i = ni.value
    n = sl.value
    return ["".join(chr(c) for c in bases[:n])] + [c[:n] for c in cols] + [ip[:ni.value], il[:ni.value]]
