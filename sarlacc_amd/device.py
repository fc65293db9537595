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
