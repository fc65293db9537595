"""Flat string-set representation used at the C-ABI boundary.

The reference hands `.Call` routines Biostrings XStringSet objects or character
vectors (/root/reference/src/DNA_input.cpp:82-88); across the C ABI these become
one concatenated byte buffer plus n+1 int64 offsets.
"""
import numpy as np


class StringSet:
    """Concatenated bytes + offsets.  `chars` always holds at least one byte so
    that its data pointer is valid for empty sets."""

    __slots__ = ("chars", "off")

    def __init__(self, chars, off):
        self.chars = chars
        self.off = off

    @classmethod
    def from_strings(cls, strings):
        if isinstance(strings, StringSet):
            return strings
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        if bs:
            np.cumsum([len(b) for b in bs], out=off[1:])
        joined = b"".join(bs)
        chars = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(1, np.uint8)
        return cls(chars, off)

    @classmethod
    def from_matrix(cls, mat):
        """n x L uint8 matrix of equal-length strings."""
        mat = np.ascontiguousarray(mat, dtype=np.uint8)
        n, L = mat.shape
        off = np.arange(n + 1, dtype=np.int64) * L
        chars = mat.reshape(-1) if mat.size else np.zeros(1, np.uint8)
        return cls(chars, off)

    def __len__(self):
        return len(self.off) - 1

    @property
    def total(self):
        return int(self.off[-1])

    def widths(self):
        return np.diff(self.off)

    def to_strings(self):
        raw = self.chars.tobytes()
        o = self.off
        return [raw[o[i]:o[i + 1]].decode() for i in range(len(o) - 1)]

    def __getitem__(self, i):
        return self.chars[self.off[i]:self.off[i + 1]].tobytes().decode()

    def slice(self, lo, hi):
        """Contiguous range [lo, hi) without a gather (views the same bytes)."""
        o = self.off[lo:hi + 1]
        chars = self.chars[int(o[0]):int(o[-1])]
        return StringSet(chars if chars.size else np.zeros(1, np.uint8), o - o[0])

    @classmethod
    def concat(cls, sets):
        """c(...) of string sets."""
        sets = list(sets)
        if not sets:
            return cls(np.zeros(1, np.uint8), np.zeros(1, np.int64))
        chars = np.concatenate([x.chars[:x.total] for x in sets])
        base = np.cumsum([0] + [x.total for x in sets[:-1]])
        off = np.concatenate([np.zeros(1, np.int64)] + [x.off[1:] + b for x, b in zip(sets, base)])
        return cls(chars if chars.size else np.zeros(1, np.uint8), off.astype(np.int64))

    def subset(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        w = self.off[idx + 1] - self.off[idx]
        off = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(w, out=off[1:])
        total = int(off[-1])
        chars = np.zeros(max(total, 1), np.uint8)
        if total:
            # gather positions: for each output byte, its source index
            src = np.repeat(self.off[idx] - off[:-1], w) + np.arange(total, dtype=np.int64)
            chars[:total] = self.chars[src]
        return StringSet(chars, off)


class StrList:
    """Read-only list of strings backed by a StringSet: the stand-in for an R character vector
    column (read names, extracted sub-sequences) that is only decoded when someone looks.
    Compares equal to a Python list with the same strings."""

    __slots__ = ("ss",)

    def __init__(self, ss):
        self.ss = ss if isinstance(ss, StringSet) else StringSet.from_strings(ss)

    def __len__(self):
        return len(self.ss)

    def __iter__(self):
        return iter(self.ss.to_strings())

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            n = len(self.ss)
            if i < 0:
                i += n
            if not 0 <= i < n:
                raise IndexError("index out of range")
            return self.ss[int(i)]
        if isinstance(i, slice):
            return StrList(self.ss.subset(np.arange(len(self.ss))[i]))
        i = np.asarray(i)
        return self.select(i) if i.dtype == bool else StrList(self.ss.subset(i))

    def __eq__(self, other):
        if isinstance(other, StrList):
            o = other.ss
            return (len(o) == len(self.ss) and np.array_equal(o.off, self.ss.off)
                    and np.array_equal(o.chars[:o.total], self.ss.chars[:self.ss.total]))
        if isinstance(other, (list, tuple)):
            return len(other) == len(self.ss) and self.ss.to_strings() == list(other)
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __repr__(self):
        return "StrList(%d strings)" % len(self.ss)

    def tolist(self):
        return self.ss.to_strings()

    def __add__(self, other):       # list concatenation yields a plain list
        return self.tolist() + list(other)

    def __radd__(self, other):
        return list(other) + self.tolist()

    def select(self, keep):
        """Rows where the boolean mask `keep` is set."""
        return StrList(self.ss.subset(np.flatnonzero(np.asarray(keep, dtype=bool))))

    @staticmethod
    def where(mask, a, b):
        """Element-wise a[i] if mask[i] else b[i] (both StrList of the same length)."""
        mask = np.asarray(mask, dtype=bool)
        n = len(a)
        both = StringSet(np.concatenate([a.ss.chars[:a.ss.total], b.ss.chars[:b.ss.total], np.zeros(1, np.uint8)]),
                         np.concatenate([a.ss.off, b.ss.off[1:] + a.ss.off[-1]]))
        return StrList(both.subset(np.where(mask, np.arange(n), n + np.arange(n))))


def csr_from_lists(lists):
    """list of integer sequences -> (int64 offsets, int32 values)"""
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    if len(lists):
        np.cumsum([len(x) for x in lists], out=off[1:])
    if off[-1]:
        vals = np.concatenate([np.asarray(x, dtype=np.int32).reshape(-1) for x in lists])
    else:
        vals = np.zeros(1, np.int32)
    return off, np.ascontiguousarray(vals, dtype=np.int32)


def lists_from_csr(off, vals, n=None):
    n = len(off) - 1 if n is None else n
    return [vals[off[i]:off[i + 1]].copy() for i in range(n)]
