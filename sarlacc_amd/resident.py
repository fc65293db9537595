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
    def borrow(cls, address, nbytes):
        """A view of device memory somebody else owns (e.g. a torch tensor's data_ptr()): never freed here."""
        b = object.__new__(cls)
        b.nbytes = int(nbytes)
        b.ptr = C.c_void_p(int(address))
        b.owned = False
        return b

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(_lib.lib().sarlacc_dev_upload(b.ptr, ptr(a), C.c_int64(a.nbytes)))
        return b

    def to_numpy(self, dtype, count):
        out = _lib.host_array(max(count, 1), dtype)
        check(_lib.lib().sarlacc_dev_download(ptr(out), self.ptr, C.c_int64(count * out.itemsize)))
        return out[:count]

    def __del__(self):
        try:
            if self.ptr and getattr(self, "owned", True):
                _lib.lib().sarlacc_dev_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


class DeviceReads:
    """seq / qual / offsets in HBM + the host copy of the offsets."""

    def __init__(self, seq, qual, off_dev, off_host, encoding):
        self.seq, self.qual, self.off, self.off_host, self.encoding = seq, qual, off_dev, off_host, encoding
        self.names = None

    @classmethod
    def upload(cls, reads):
        """reads: generics.Reads"""
        s, q = reads.seq, reads.qual
        if (s.widths() != q.widths()).any():
            raise _lib.SarlaccError("sequence and quality strings should have the same length")
        return cls(DevBuffer.from_numpy(s.chars), DevBuffer.from_numpy(q.chars), DevBuffer.from_numpy(s.off), s.off.copy(),
                   reads.encoding)

    @classmethod
    def from_fastq(cls, source, encoding=None):
        """FASTQ text (path, bytes or uint8 array) -> resident batch, parsed on the device
        (sarlacc_dev_fastq_index / _extract; replaces FastqStreamer + .FASTQ2QSDS,
        R/adaptorAlign.R:26-37,:104-110).  Read names are kept on the host in `.names`."""
        if isinstance(source, (bytes, bytearray)):
            text = np.frombuffer(bytes(source), dtype=np.uint8)
        elif isinstance(source, np.ndarray):
            text = np.ascontiguousarray(source, dtype=np.uint8)
        else:
            text = np.fromfile(source, dtype=np.uint8)
        d_text = DevBuffer.from_numpy(text if text.size else np.zeros(1, np.uint8))
        return cls._from_device_text(d_text.ptr, text.size, encoding)

    @classmethod
    def _from_device_text(cls, text_ptr, nbytes, encoding):
        nrec, tb, tn = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(_lib.lib().sarlacc_dev_fastq_index(text_ptr, C.c_int64(nbytes), C.byref(nrec), C.byref(tb), C.byref(tn), None))
        n = nrec.value
        seq, qual = DevBuffer(max(tb.value, 1)), DevBuffer(max(tb.value, 1))
        off, names, noff = DevBuffer(8 * (n + 1)), DevBuffer(max(tn.value, 1)), DevBuffer(8 * (n + 1))
        check(_lib.lib().sarlacc_dev_fastq_extract(text_ptr, seq.ptr, qual.ptr, off.ptr, names.ptr, noff.ptr, None))
        out = cls(seq, qual, off, off.to_numpy(np.int64, n + 1), encoding)
        from .strset import StrList
        nraw = names.to_numpy(np.uint8, tn.value)
        out.names = StrList(StringSet(nraw if nraw.size else np.zeros(1, np.uint8), noff.to_numpy(np.int64, n + 1)))   # decoded on demand
        return out

    @classmethod
    def stream_fastq(cls, path, number, encoding=None, block_bytes=256 << 20):
        """Generator over the file in chunks of at most `number` records, each a resident batch: the
        FastqStreamer(filepath, n=number) + yield() loop of R/adaptorAlign.R:26-37.  The file is read in
        blocks of `block_bytes`; where a block ends inside a record the device reports the end of the
        last complete one (sarlacc_dev_fastq_split) and the rest is carried over to the next block, so
        neither the host nor the device ever holds more than one block plus one chunk."""
        number = int(number)
        if number < 1:
            raise ValueError("'number' must be a positive integer")
        carry = np.zeros(0, np.uint8)
        with open(path, "rb") as fh:
            eof = False
            while not eof:
                fresh = np.frombuffer(fh.read(int(block_bytes)), dtype=np.uint8)
                eof = fresh.size < int(block_bytes)
                text = np.concatenate([carry, fresh]) if carry.size else fresh
                if text.size == 0:
                    break
                d_text = DevBuffer.from_numpy(text)
                pos = 0
                while pos < text.size:
                    nrec, used = C.c_int64(0), C.c_int64(0)
                    here = C.c_void_p(d_text.ptr.value + pos)
                    check(_lib.lib().sarlacc_dev_fastq_split(here, C.c_int64(text.size - pos), C.c_int64(number), C.byref(nrec),
                                                             C.byref(used), None))
                    if nrec.value < number and not eof:
                        break                           # the chunk continues in the next block
                    if nrec.value < number:
                        used.value = text.size - pos    # last chunk: may end without a newline, or in blank lines
                    chunk = cls._from_device_text(here, used.value, encoding)
                    pos += used.value
                    if len(chunk):
                        yield chunk
                carry = text[pos:].copy()

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

    def subseq(self, start, width, other=None, from_other=None):
        """XVector::subseq on the resident batch, straight to a host StringSet: element r = `width[r]` bases from the 1-based
        position `start[r]` of read r -- of `other`'s read r where `from_other[r]` (sarlacc_dev_subseq)."""
        n = len(self)
        st = np.ascontiguousarray(start, dtype=np.int32)
        wd = np.ascontiguousarray(np.maximum(np.asarray(width, dtype=np.int64), 0), dtype=np.int32)
        off = np.zeros(n + 1, np.int64)
        total = int(wd.sum(dtype=np.int64))
        chars = _lib.host_array(max(total, 1), np.uint8)
        sel = None if from_other is None else np.ascontiguousarray(from_other, dtype=np.uint8)
        check(_lib.lib().sarlacc_dev_subseq(self.seq.ptr, self.off.ptr, other.seq.ptr if other is not None else None,
                                            other.off.ptr if other is not None else None, ptr(sel) if sel is not None else None,
                                            ptr(st), ptr(wd), C.c_int64(n), ptr(chars), C.c_int64(chars.size), ptr(off), None))
        return StringSet(chars, off)

    def realize(self, idx, reversed_, trim_start=None, trim_end=None):
        """Reads `idx` (0-based) of this batch, reverse-complemented where `reversed_`, cut to the
        1-based inclusive [trim_start, trim_end] of the oriented read (whole read when None):
        the device half of realizeReads (R/realizeReads.R:28-43)."""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        rev = np.ascontiguousarray(reversed_, dtype=np.uint8)
        w = np.diff(self.off_host)[idx]
        ts = np.ones(idx.size, np.int32) if trim_start is None else np.ascontiguousarray(trim_start, dtype=np.int32)
        te = w.astype(np.int32) if trim_end is None else np.ascontiguousarray(trim_end, dtype=np.int32)
        if idx.size and ((ts < 1).any() or (te > w).any()):
            raise _lib.SarlaccError("trim coordinates outside the read")
        ow = np.maximum(te.astype(np.int64) - ts + 1, 0)
        ooff = np.zeros(idx.size + 1, np.int64)
        np.cumsum(ow, out=ooff[1:])
        d = self._like(ooff)
        if idx.size:
            d_idx, d_rev, d_ts = DevBuffer.from_numpy(idx), DevBuffer.from_numpy(rev), DevBuffer.from_numpy(ts)   # kept alive over the call
            check(_lib.lib().sarlacc_dev_realize(self.seq.ptr, self.qual.ptr, self.off.ptr, d_idx.ptr, d_rev.ptr, d_ts.ptr,
                                                 C.c_int64(idx.size), d.off.ptr, d.seq.ptr, d.qual.ptr, None))
        return d

    def scramble(self, seed):
        """.scramble_input (R/getAdaptorThresholds.R:68-92) on the device, deterministic in `seed`."""
        d = self._like(self.off_host.copy())
        check(_lib.lib().sarlacc_dev_scramble(self.seq.ptr, self.qual.ptr, self.off.ptr, C.c_int64(len(self)),
                                              C.c_uint64(int(seed)), d.seq.ptr, d.qual.ptr, None))
        return d

    def align_block(self, adaptor, gap_opening, gap_extension, sec_starts=(), sec_ends=()):
        """adaptor_align (src/adaptor_align.cpp:11-77) on the resident batch, results left in HBM as ONE block (DevBuffer, n,
        number of sections): scores | starts | ends | section starts | section widths (include/sarlacc_amd.h,
        sarlacc_dev_choose_strand) -- one allocation per call, and one download for whoever wants them on the host."""
        n = len(self)
        ss = np.ascontiguousarray(sec_starts, dtype=np.int32).reshape(-1)
        se = np.ascontiguousarray(sec_ends, dtype=np.int32).reshape(-1)
        ns = ss.size
        enc = as_encoding(self.encoding if self.encoding is not None else phred_encoding())
        rf = adaptor.encode() if isinstance(adaptor, str) else bytes(adaptor)
        nsec = max(ns, 1)
        blk = DevBuffer(16 * n + 8 * n * nsec)
        base = blk.ptr.value
        pad = np.zeros(1, np.int32)
        check(_lib.lib().sarlacc_dev_align(
            self.seq.ptr, self.qual.ptr, self.off.ptr, C.c_int64(n), C.c_int32(self.max_len),
            ptr(enc.errors), enc.names, len(enc), C.c_double(gap_opening), C.c_double(gap_extension),
            rf, len(rf), 0, ptr(ss if ns else pad), ptr(se if ns else pad), ns,
            C.c_void_p(base), C.c_void_p(base + 8 * n), C.c_void_p(base + 12 * n), C.c_void_p(base + 16 * n),
            C.c_void_p(base + 16 * n + 4 * n * nsec), None))
        return blk, n, ns

    @staticmethod
    def block_to_host(block):
        """A result block of align_block / choose_strand as (scores, starts, ends, [section starts], [section widths])."""
        blk, n, ns = block
        nsec = max(ns, 1)
        o_st, o_en, o_so, o_sw = 8 * n, 12 * n, 16 * n, 16 * n + 4 * n * nsec
        host = _lib.host_array(blk.nbytes, np.uint8)
        check(_lib.lib().sarlacc_dev_download(ptr(host), blk.ptr, C.c_int64(blk.nbytes)))
        scores = host[:o_st].view(np.float64)
        starts, ends = host[o_st:o_en].view(np.int32), host[o_en:o_so].view(np.int32)
        so_h, sw_h = host[o_so:o_sw].view(np.int32), host[o_sw:].view(np.int32)
        return (scores, starts, ends, [so_h[k * n:(k + 1) * n] for k in range(ns)], [sw_h[k * n:(k + 1) * n] for k in range(ns)])

    @staticmethod
    def choose_strand(cs, ce, rs, re):
        """.resolve_strand + the row selection of .align_AA_internal (R/adaptorAlign.R:112-122, :190-207) on four result blocks
        in HBM (sarlacc_dev_choose_strand): (rows of adaptor 1, rows of adaptor 2, reversed) on the host -- half the bytes of
        the four blocks cross PCIe and no host pass selects rows."""
        n = cs[1]
        if not (ce[1] == rs[1] == re[1] == n and cs[2] == rs[2] and ce[2] == re[2]):
            raise _lib.SarlaccError("strand choice: result blocks of different shapes")
        out1, out2, rev = DevBuffer(cs[0].nbytes), DevBuffer(ce[0].nbytes), DevBuffer(max(n, 1))
        check(_lib.lib().sarlacc_dev_choose_strand(cs[0].ptr, ce[0].ptr, rs[0].ptr, re[0].ptr, C.c_int64(n), cs[2], ce[2],
                                                   out1.ptr, out2.ptr, rev.ptr, None))
        return (DeviceReads.block_to_host((out1, n, cs[2])), DeviceReads.block_to_host((out2, n, ce[2])),
                rev.to_numpy(np.uint8, n).view(np.bool_))

    def align_map(self, adaptor, gap_opening, gap_extension, sec_starts=(), sec_ends=()):
        """adaptor_align (src/adaptor_align.cpp:11-77) on the resident batch: (scores, starts, ends,
        [section starts], [section widths]) as numpy arrays, same conventions as calls.adaptor_align."""
        ns = np.asarray(sec_starts).size
        if len(self) == 0:
            return np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.int32), [np.zeros(0, np.int32)] * ns, [np.zeros(0, np.int32)] * ns
        return self.block_to_host(self.align_block(adaptor, gap_opening, gap_extension, sec_starts, sec_ends))

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
