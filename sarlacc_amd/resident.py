"""Device-resident read batches (include/sarlacc_amd.h "resident batches"): reads are uploaded
once and stay in HBM while they are windowed, shuffled and aligned many times -- the access
pattern of tuneAlignment / getAdaptorThresholds (/root/reference/R/tuneAlignment.R,
R/getAdaptorThresholds.R).  No torch needed: allocation and copies go through the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr
from .encoding import as_encoding, phred_encoding
from .strset import StringSet


class DevBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = C.c_void_p()
        check(_lib.lib().sarlacc_dev_malloc(C.byref(self.ptr), C.c_int64(self.nbytes)))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(_lib.lib().sarlacc_dev_upload(b.ptr, ptr(a), C.c_int64(a.nbytes)))
        return b

    def to_numpy(self, dtype, count):
        out = np.zeros(max(count, 1), dtype=dtype)
        check(_lib.lib().sarlacc_dev_download(ptr(out), self.ptr, C.c_int64(count * out.itemsize)))
        return out[:count]

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().sarlacc_dev_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


class DeviceReads:
    """seq / qual / offsets in HBM + the host copy of the offsets."""

    def __init__(self, seq, qual, off_dev, off_host, encoding):
        self.seq, self.qual, self.off, self.off_host, self.encoding = seq, qual, off_dev, off_host, encoding

    @classmethod
    def upload(cls, reads):
        """reads: generics.Reads"""
        s, q = reads.seq, reads.qual
        if (s.widths() != q.widths()).any():
            raise _lib.SarlaccError("sequence and quality strings should have the same length")
        return cls(DevBuffer.from_numpy(s.chars), DevBuffer.from_numpy(q.chars), DevBuffer.from_numpy(s.off), s.off.copy(),
                   reads.encoding)

    def __len__(self):
        return len(self.off_host) - 1

    @property
    def total(self):
        return int(self.off_host[-1])

    @property
    def max_len(self):
        return int(np.diff(self.off_host).max()) if len(self) else 0

    def download(self):
        return (StringSet(self.seq.to_numpy(np.uint8, self.total), self.off_host.copy()),
                StringSet(self.qual.to_numpy(np.uint8, self.total), self.off_host.copy()))

    def _like(self, off_host):
        total = int(off_host[-1])
        return DeviceReads(DevBuffer(total), DevBuffer(total), DevBuffer.from_numpy(off_host), off_host, self.encoding)

    def front_and_back(self, tolerance):
        """.get_front_and_back (R/adaptorAlign.R:86-95) on the device."""
        w = np.minimum(int(tolerance), np.diff(self.off_host))
        woff = np.zeros(len(self) + 1, np.int64)
        np.cumsum(w, out=woff[1:])
        out = []
        for which in (0, 1):
            d = self._like(woff.copy())
            check(_lib.lib().sarlacc_dev_windows(self.seq.ptr, self.qual.ptr, self.off.ptr, C.c_int64(len(self)),
                                                 d.off.ptr, which, d.seq.ptr, d.qual.ptr, None))
            out.append(d)
        return out[0], out[1]

    def scramble(self, seed):
        """.scramble_input (R/getAdaptorThresholds.R:68-92) on the device, deterministic in `seed`."""
        d = self._like(self.off_host.copy())
        check(_lib.lib().sarlacc_dev_scramble(self.seq.ptr, self.qual.ptr, self.off.ptr, C.c_int64(len(self)),
                                              C.c_uint64(int(seed)), d.seq.ptr, d.qual.ptr, None))
        return d

    def align_scores(self, adaptor, gap_opening, gap_extension, local=True):
        """adaptor_align_score_only / barcode_align on the resident batch."""
        n = len(self)
        if n == 0:
            return np.zeros(0)
        enc = as_encoding(self.encoding if self.encoding is not None else phred_encoding())
        rf = adaptor.encode() if isinstance(adaptor, str) else bytes(adaptor)
        scores = DevBuffer(8 * n)
        one = np.zeros(1, np.int32)
        check(_lib.lib().sarlacc_dev_align(
            self.seq.ptr, self.qual.ptr, self.off.ptr, C.c_int64(n), C.c_int32(self.max_len),
            ptr(enc.errors), enc.names, len(enc), C.c_double(gap_opening), C.c_double(gap_extension),
            rf, len(rf), 0 if local else 1, ptr(one), ptr(one), 0, scores.ptr, None, None, None, None, None))
        return scores.to_numpy(np.float64, n)
