"""Device-resident entry points (sarlacc_dev_* of include/sarlacc_amd.h): inputs and
outputs are raw device pointers (e.g. torch tensors' data_ptr()), work is enqueued on
the caller's HIP stream.  Used by bench.py and by pipelines that keep reads in HBM."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr
from .encoding import as_encoding


def _dp(x):
    """torch tensor / int / None -> void*"""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


def dev_pack_reads(d_seq, total, d_packed, d_nmask, stream=0):
    """sarlacc_dev_pack_reads: ASCII bases -> 2-bit packed bases + exception bit-mask (both on device)."""
    check(_lib.lib().sarlacc_dev_pack_reads(_dp(d_seq), C.c_int64(total), _dp(d_packed), _dp(d_nmask), C.c_void_p(int(stream))))


def dev_align(d_seq, d_qual, d_off, n, max_len, encoding, gapopen, gapext, reference, local=True,
              sec_starts=(), sec_ends=(), d_scores=None, d_starts=None, d_ends=None,
              d_sec_start=None, d_sec_width=None, stream=0, d_nmask=None):
    """sarlacc_dev_align(_packed): quality-weighted DP of `reference` against n device-resident reads.
    With d_starts/d_ends given the traceback (adaptor_align) variant runs, else scores only.
    With d_nmask given, d_seq holds 2-bit packed bases (dev_pack_reads)."""
    enc = as_encoding(encoding)
    rf = reference.encode() if isinstance(reference, str) else bytes(reference)
    ss = np.ascontiguousarray(sec_starts, dtype=np.int32).reshape(-1)
    se = np.ascontiguousarray(sec_ends, dtype=np.int32).reshape(-1)
    ns = ss.size
    if ns == 0:
        ss = np.zeros(1, np.int32)
        se = np.zeros(1, np.int32)
    if d_nmask is not None:
        check(_lib.lib().sarlacc_dev_align_packed(
            _dp(d_seq), _dp(d_nmask), _dp(d_qual), _dp(d_off), C.c_int64(n), C.c_int32(max_len),
            ptr(enc.errors), enc.names, len(enc), C.c_double(gapopen), C.c_double(gapext),
            rf, len(rf), 0 if local else 1, ptr(ss), ptr(se), ns,
            _dp(d_scores), _dp(d_starts), _dp(d_ends), _dp(d_sec_start), _dp(d_sec_width),
            C.c_void_p(int(stream))))
        return
    check(_lib.lib().sarlacc_dev_align(
        _dp(d_seq), _dp(d_qual), _dp(d_off), C.c_int64(n), C.c_int32(max_len),
        ptr(enc.errors), enc.names, len(enc), C.c_double(gapopen), C.c_double(gapext),
        rf, len(rf), 0 if local else 1, ptr(ss), ptr(se), ns,
        _dp(d_scores), _dp(d_starts), _dp(d_ends), _dp(d_sec_start), _dp(d_sec_width),
        C.c_void_p(int(stream))))


def dev_msa_consensus(grp_off, grp, d_seq, d_qual, off_host, match, mismatch, gapExtension, gapOpening, bandwidth,
                      min_cov, pseudo_count=1.0, encoding=None):
    """sarlacc_dev_msa_consensus: multiReadAlign + consensusReadSeq on reads (and qualities) already in
    HBM.  `off_host` is the host copy (numpy int64[n+1]) of the read offsets, grp_off/grp the CSR of
    1-based group lists.  d_qual None -> basic vote.  Returns (consensus StringSet, phred StringSet)."""
    import re

    from ._lib import SarlaccError
    from .strset import StringSet
    goff = np.ascontiguousarray(grp_off, dtype=np.int64)
    gvals = np.ascontiguousarray(grp, dtype=np.int32)
    if gvals.size == 0:
        gvals = np.zeros(1, np.int32)
    off = np.ascontiguousarray(off_host, dtype=np.int64)
    n = off.size - 1
    ng = goff.size - 1
    enc = as_encoding(encoding) if d_qual is not None else None
    coff = np.zeros(ng + 1, np.int64)
    w = np.diff(off)
    sizes = np.diff(goff)
    longest = np.maximum.reduceat(w[gvals[:int(goff[-1])].astype(np.int64) - 1], goff[:-1][sizes > 0]) if goff[-1] else np.zeros(0)
    cap = int(1.5 * longest.sum()) + 1024
    for attempt in range(2):
        cons = _lib.host_array(cap, np.uint8)    # (page-locked: 360 MB come back at a 10^6-read pass)
        phred = _lib.host_array(cap, np.uint8)
        try:
            check(_lib.lib().sarlacc_dev_msa_consensus(
                ptr(goff), ptr(gvals), C.c_int64(ng), _dp(d_seq), _dp(d_qual), ptr(off), C.c_int64(n),
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
