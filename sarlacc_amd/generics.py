"""Counterparts of the reference's exported R generics on the hot path
(/root/reference/R/*.R), written against sarlacc_amd.calls the same way the R
functions are written against `.Call(cxx_*)`.  The reference's toolchain (R,
Biostrings, S4Vectors) is absent from the build image, so this host layer is Python;
names, argument meaning, defaults, return structure and error behaviour follow the R
code, with plain containers standing in for the Bioconductor classes:

    QualityScaledDNAStringSet  ->  Reads(seq, qual, names)
    DataFrame                  ->  dict of numpy arrays / lists (+ "metadata" dict)

Only orchestration lives here -- every arithmetic step is a call into the HIP library.
"""
import re

import os

import numpy as np

from . import calls
from .encoding import QUAL_TYPES, Encoding, encoding_for_qual_type, phred_encoding
from .mock import _COMP
from .strset import StringSet, StrList


class Reads:
    """Minimal stand-in for QualityScaledDNAStringSet: sequences + (optional) Phred+33 qualities."""

    def __init__(self, seq, qual=None, names=None, encoding=None):
        self.seq = StringSet.from_strings(seq)
        self.qual = None if qual is None else StringSet.from_strings(qual)
        if self.qual is not None and len(self.qual) != len(self.seq):
            raise ValueError("sequence and quality vectors should have the same length")
        self.names = list(names) if names is not None else None
        self.encoding = encoding if encoding is not None else phred_encoding()

    def __len__(self):
        return len(self.seq)

    def width(self):
        return self.seq.widths()

    def subset(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        return Reads(self.seq.subset(idx), None if self.qual is None else self.qual.subset(idx),
                     None if self.names is None else [self.names[i] for i in idx], self.encoding)


def read_fastq(path):
    """4-line FASTQ -> Reads (the reference streams with ShortRead::FastqStreamer,
    R/adaptorAlign.R:26-37).  Host reader; resident.DeviceReads.from_fastq parses on the device."""
    names, seqs, quals = [], [], []
    with open(path, "rb") as fh:
        while True:
            h = fh.readline()
            if not h:
                break
            if not h.strip():   # blank lines after the last record
                continue
            s = fh.readline().rstrip(b"\r\n")
            fh.readline()
            q = fh.readline().rstrip(b"\r\n")
            names.append(h[1:].rstrip(b"\r\n").decode())
            seqs.append(s.upper())
            quals.append(q)
    return Reads(seqs, quals, names)


def write_fastq(path, reads, append=False):
    with open(path, "ab" if append else "wb") as fh:
        for i, (s, q) in enumerate(zip(reads.seq.to_strings(), reads.qual.to_strings())):
            name = reads.names[i] if reads.names else "READ_%d" % (i + 1)
            fh.write(b"@" + name.encode() + b"\n" + s.encode() + b"\n+\n" + q.encode() + b"\n")


# ---------------------------------------------------------------------------
# adaptorAlign (R/adaptorAlign.R)

def _setup_subseqs(adaptor):
    """.setup_subseqs (R/adaptorAlign.R:136-143): maximal runs of non-ACTG, 1-based inclusive."""
    starts, ends = [], []
    for m in re.finditer(r"[^ACTG]+", adaptor):
        starts.append(m.start() + 1)
        ends.append(m.end())
    return {"starts": np.array(starts, dtype=np.int32), "ends": np.array(ends, dtype=np.int32)}


def _subseq(ss, start, width):
    """XVector::subseq on a StringSet with per-element 1-based start and width."""
    start = np.asarray(start, dtype=np.int64)
    width = np.asarray(width, dtype=np.int64)
    off = np.zeros(len(ss) + 1, dtype=np.int64)
    np.cumsum(width, out=off[1:])
    total = int(off[-1])
    chars = np.zeros(max(total, 1), np.uint8)
    if total:
        src = np.repeat(ss.off[:-1] + start - 1 - off[:-1], width) + np.arange(total, dtype=np.int64)
        chars[:total] = ss.chars[src]
    return StringSet(chars, off)


def _reverse_each(ss, complement):
    """Per-string reversal (optionally complemented) of a StringSet."""
    total = ss.total
    chars = np.zeros(max(total, 1), np.uint8)
    if total:
        w = ss.widths()
        pos = np.arange(total, dtype=np.int64) - np.repeat(ss.off[:-1], w)
        src = np.repeat(ss.off[:-1] + w - 1, w) - pos
        vals = ss.chars[src]
        chars[:total] = _COMP[vals] if complement else vals
    return StringSet(chars, ss.off.copy())


def _get_front_and_back(reads, tolerance):
    """.get_front_and_back (R/adaptorAlign.R:86-95): first min(tol,width) bases, and the
    reverse complement of the last min(tol,width) bases (qualities reversed)."""
    w = reads.width()
    tol = np.minimum(int(tolerance), w)
    front = Reads(_subseq(reads.seq, np.ones(len(w), np.int64), tol), _subseq(reads.qual, np.ones(len(w), np.int64), tol),
                  encoding=reads.encoding)
    bstart = w - tol + 1
    back = Reads(_reverse_each(_subseq(reads.seq, bstart, tol), True), _reverse_each(_subseq(reads.qual, bstart, tol), False),
                 encoding=reads.encoding)
    return front, back


def _align_and_extract(adaptor, reads, gap_opening, gap_extension, subseq_starts, subseq_ends):
    """.align_and_extract (R/adaptorAlign.R:145-178)."""
    out = calls.adaptor_align(reads.seq, reads.qual, reads.encoding, gap_opening, gap_extension, adaptor,
                              np.asarray(subseq_starts, dtype=np.int32) - 1, subseq_ends)
    res = {"score": out[0], "start": out[1], "end": out[2], "subseq": {}}
    for i, (st, wd) in enumerate(zip(out[3], out[4])):
        res["subseq"]["Sub%d" % (i + 1)] = _subseq(reads.seq, st, wd).to_strings()
    return res


def _resolve_strand(start_score, end_score, rc_start_score, rc_end_score):
    """.resolve_strand (R/adaptorAlign.R:112-122): reverse iff fscore < rscore (strict)."""
    fscore = np.maximum(start_score, 0) + np.maximum(end_score, 0)
    rscore = np.maximum(rc_start_score, 0) + np.maximum(rc_end_score, 0)
    rev = fscore < rscore
    return rev, np.where(rev, rscore, fscore)


def _swap_rows(a, b, mask):
    for key in ("score", "start", "end"):
        a[key] = np.where(mask, b[key], a[key])
    for k in a["subseq"]:
        if isinstance(a["subseq"][k], StrList):
            a["subseq"][k] = StrList.where(mask, b["subseq"][k], a["subseq"][k])
        else:
            a["subseq"][k] = [y if m else x for x, y, m in zip(a["subseq"][k], b["subseq"][k], mask)]


def _align_and_extract_resident(adaptor, dev, host_seq, gap_opening, gap_extension, subseq_starts, subseq_ends):
    """.align_and_extract on a resident window batch (DeviceReads + the host copy of its bases)."""
    out = dev.align_map(adaptor, gap_opening, gap_extension, np.asarray(subseq_starts, dtype=np.int32) - 1, subseq_ends)
    res = {"score": out[0], "start": out[1], "end": out[2], "subseq": {}}
    for i, (st, wd) in enumerate(zip(out[3], out[4])):
        res["subseq"]["Sub%d" % (i + 1)] = StrList(_subseq(host_seq, st, wd))   # decoded on demand
    return res


def _concat_aligned(parts):
    """rbind of the per-chunk alignment tables (R/adaptorAlign.R:59-60)."""
    if len(parts) == 1:
        return parts[0]
    out = {"subseq": {}}
    for key in ("score", "start", "end"):
        out[key] = np.concatenate([p[key] for p in parts])
    for k in parts[0]["subseq"]:
        cols = [p["subseq"][k] for p in parts]
        if all(isinstance(c, StrList) for c in cols):
            out["subseq"][k] = StrList(StringSet.concat([c.ss for c in cols]))
        else:
            out["subseq"][k] = [x for c in cols for x in c]
    return out


def _adaptor_align_chunk(adaptor1, adaptor2, sub1, sub2, args, tolerance, reads=None, dev=None):
    """One yield of the streamer: the four alignments of .align_AA_internal (R/adaptorAlign.R:180-207) and
    the strand choice, on a host batch (`reads`) or a resident one (`dev`)."""
    if dev is not None:
        # resident batch: the four alignments run on the device windows; the strand is chosen from the four score vectors, and
        # only then are the sub-sequences cut -- on the device, from the window of the chosen strand, a dozen bases per read
        # across PCIe instead of the windows themselves
        dfront, dback = dev.front_and_back(tolerance)
        if len(dev) == 0:
            empty = lambda sb: {"score": np.zeros(0), "start": np.zeros(0, np.int32), "end": np.zeros(0, np.int32),
                                "subseq": {"Sub%d" % (i + 1): StrList([]) for i in range(len(sb["starts"]))}}
            return empty(sub1), empty(sub2), np.zeros(0, bool), np.zeros(0, np.int32)
        runs = {}
        for key, (ad, d, sb) in {"cs": (adaptor1, dfront, sub1), "ce": (adaptor2, dback, sub2), "rs": (adaptor1, dback, sub1),
                                 "re": (adaptor2, dfront, sub2)}.items():
            runs[key] = d.align_block(ad, args[0], args[1], np.asarray(sb["starts"], dtype=np.int32) - 1, sb["ends"])
        # the strand is chosen and the rows are selected on the device (sarlacc_dev_choose_strand): the chosen rows come back,
        # not the four result sets (a dozen passes of numpy.where over 10^6 reads were half of this function's time)
        rows1, rows2, rev = type(dev).choose_strand(runs["cs"], runs["ce"], runs["rs"], runs["re"])

        def chosen(rows, d_cur, d_rc):
            res = {"score": rows[0], "start": rows[1], "end": rows[2], "subseq": {}}
            for i in range(len(rows[3])):
                res["subseq"]["Sub%d" % (i + 1)] = StrList(d_cur.subseq(rows[3][i], rows[4][i], other=d_rc, from_other=rev))   # decoded on demand
            return res

        cur_starts = chosen(rows1, dfront, dback)
        cur_ends = chosen(rows2, dback, dfront)
        return cur_starts, cur_ends, rev, np.diff(dev.off_host).astype(np.int32)
    else:
        front, back = _get_front_and_back(reads, tolerance)
        cur_starts = _align_and_extract(adaptor1, front, *args, sub1["starts"], sub1["ends"])
        cur_ends = _align_and_extract(adaptor2, back, *args, sub2["starts"], sub2["ends"])
        rc_starts = _align_and_extract(adaptor1, back, *args, sub1["starts"], sub1["ends"])
        rc_ends = _align_and_extract(adaptor2, front, *args, sub2["starts"], sub2["ends"])
        width = reads.width().astype(np.int32)
    rev, _ = _resolve_strand(cur_starts["score"], cur_ends["score"], rc_starts["score"], rc_ends["score"])
    _swap_rows(cur_starts, rc_starts, rev)
    _swap_rows(cur_ends, rc_ends, rev)
    return cur_starts, cur_ends, rev, width


def adaptorAlign(adaptor1, adaptor2, reads, tolerance=250, gapOpening=5, gapExtension=1, qual_type=QUAL_TYPES, number=1e5):
    """adaptorAlign (R/adaptorAlign.R:7-78).  `reads` is, as in the reference, the path of a FASTQ file:
    it is streamed in chunks of `number` records (FastqStreamer(filepath, n=number), :26), each chunk's
    text is parsed on the device, the front/back windows are cut there and the four alignments run on
    the resident windows (no host pass over the bases); `qual_type` ("phred", "solexa" or "illumina",
    match.arg semantics) names the quality class the file is read with (.qual2class, :97-99) and hence
    the encoding vector of every alignment.  A Reads object is accepted as well (one chunk; it carries
    its own encoding, so `qual_type` only lands in the metadata)."""
    adaptor1, adaptor2 = str(adaptor1).upper(), str(adaptor2).upper()
    qual_type, enc = encoding_for_qual_type(qual_type)
    filepath = None
    sub1, sub2 = _setup_subseqs(adaptor1), _setup_subseqs(adaptor2)
    args = (gapOpening, gapExtension)
    parts, names = [], []
    if isinstance(reads, (str, os.PathLike)):
        from .resident import DeviceReads
        filepath = os.fspath(reads)
        for dev in DeviceReads.stream_fastq(filepath, number, encoding=enc):
            parts.append(_adaptor_align_chunk(adaptor1, adaptor2, sub1, sub2, args, tolerance, dev=dev))
            names.append(dev.names)
        if not parts:   # "Guarantee some value is returned" (:42-50): the empty table, with its columns
            parts.append(_adaptor_align_chunk(adaptor1, adaptor2, sub1, sub2, args, tolerance, dev=DeviceReads.from_fastq(b"", enc)))
            names.append(StrList([]))
        names = names[0] if len(names) == 1 else StrList(StringSet.concat([x.ss for x in names]))
    else:
        parts.append(_adaptor_align_chunk(adaptor1, adaptor2, sub1, sub2, args, tolerance, reads=reads))
        names = reads.names
    cur_starts = _concat_aligned([p[0] for p in parts])
    cur_ends = _concat_aligned([p[1] for p in parts])
    rev = np.concatenate([p[2] for p in parts])
    width = np.concatenate([p[3] for p in parts])
    # adaptor 2 was aligned on the reverse strand: report in read coordinates (:67-71)
    cur_ends["start"], cur_ends["end"] = (width - cur_ends["start"] + 1).astype(np.int32), (width - cur_ends["end"] + 1).astype(np.int32)
    details = {"gapOpening": gapOpening, "gapExtension": gapExtension}
    cur_starts["metadata"] = dict(sequence=adaptor1, **details)
    cur_ends["metadata"] = dict(sequence=adaptor2, **details)
    return {"read.width": width, "adaptor1": cur_starts, "adaptor2": cur_ends, "reversed": rev,
            "names": names, "metadata": {"filepath": filepath, "qual.type": qual_type, "tolerance": tolerance}}


# ---------------------------------------------------------------------------
def _select(col, keep):
    """Rows of a character column (list or StrList) where `keep` is set."""
    return col.select(keep) if isinstance(col, StrList) else [x for x, k in zip(col, keep) if k]


def _subset_aligned(aligned, keep):
    """aligned[keep, ] of the reference's DataFrame (rows of every column, nested ones included)."""
    keep = np.asarray(keep, dtype=bool)
    out = {"metadata": aligned["metadata"], "read.width": aligned["read.width"][keep], "reversed": aligned["reversed"][keep],
           "names": None if aligned.get("names") is None else _select(aligned["names"], keep)}
    for key in ("adaptor1", "adaptor2"):
        a = aligned[key]
        out[key] = {"score": a["score"][keep], "start": a["start"][keep], "end": a["end"][keep], "metadata": a.get("metadata"),
                    "subseq": {k: _select(v, keep) for k, v in a["subseq"].items()}}
    for extra in ("trim.start", "trim.end"):
        if extra in aligned:
            out[extra] = aligned[extra][keep]
    return out


def filterReads(aligned, score1, score2, essential1=True, essential2=True):
    """filterReads (R/filterReads.R:2-43): drop reads whose essential adaptors score below the
    thresholds, then record the cut points between the adaptors as trim.start / trim.end."""
    n = len(aligned["read.width"])
    id1 = aligned["adaptor1"]["score"] >= score1 if essential1 else np.ones(n, bool)
    id2 = aligned["adaptor2"]["score"] >= score2 if essential2 else np.ones(n, bool)
    aligned = _subset_aligned(aligned, id1 & id2)
    start_point = np.ones(len(aligned["read.width"]), np.int32)
    has1 = aligned["adaptor1"]["score"] >= score1
    start_point[has1] = aligned["adaptor1"]["end"][has1] + 1
    end_point = aligned["read.width"].astype(np.int32).copy()
    has2 = aligned["adaptor2"]["score"] >= score2
    end_point[has2] = aligned["adaptor2"]["end"][has2] - 1
    keep = start_point < end_point
    out = _subset_aligned(aligned, keep)
    out["trim.start"], out["trim.end"] = start_point[keep], end_point[keep]
    return out


def realizeReads(aligned, number=1e5, trim=True, resident=False):
    """realizeReads (R/realizeReads.R:5-46): the reads named in `aligned`, re-read from its FASTQ file
    (streamed in chunks of `number` records with the quality class recorded by adaptorAlign, :11-25),
    reverse-complemented where `reversed` and trimmed to [trim.start, trim.end].  Every chunk is parsed
    on the device and the orientation / trimming happen there (DeviceReads.realize); the result comes
    back as Reads, or stays in HBM with resident=True."""
    import warnings
    from .resident import DeviceReads
    qual_type, enc = encoding_for_qual_type(aligned["metadata"].get("qual.type", "phred"))
    want = list(aligned["names"])
    rows_of = {}
    for r, nm in enumerate(want):
        rows_of.setdefault(nm, []).append(r)
    rev_all = np.asarray(aligned["reversed"], dtype=bool)
    ts = te = None
    if trim:
        if "trim.start" in aligned:
            ts, te = np.asarray(aligned["trim.start"]), np.asarray(aligned["trim.end"])
        else:
            warnings.warn("no 'trim.start' detected, run 'filterReads' first")
    parts, part_rows, seen = [], [], set()
    for dev in DeviceReads.stream_fastq(aligned["metadata"]["filepath"], number, encoding=enc):
        idx, rows = [], []
        for i, nm in enumerate(dev.names):
            if nm in rows_of and nm not in seen:   # match(): first occurrence in the file
                seen.add(nm)
                for r in rows_of[nm]:
                    idx.append(i)
                    rows.append(r)
        if not rows:
            continue
        rows = np.array(rows, dtype=np.int64)
        parts.append(dev.realize(np.array(idx, dtype=np.int64), rev_all[rows], None if ts is None else ts[rows],
                                 None if te is None else te[rows]))
        part_rows.append(rows)
    if len(seen) != len(rows_of):
        raise ValueError("read names in 'aligned' not present in FASTQ file")
    if len(parts) == 1 and np.array_equal(part_rows[0], np.arange(len(want))):
        out = parts[0]                              # one chunk, already in the order of `aligned`
        out.names = want
        if resident:
            return out
        seq, qual = out.download()
        return Reads(seq, qual, want, enc)
    # several chunks: each realized part comes back, the rows are put in the order of `aligned`
    back = np.zeros(len(want), np.int64)
    if part_rows:
        back[np.concatenate(part_rows)] = np.arange(len(want))
    host = [p.download() for p in parts]
    seq = StringSet.concat([h[0] for h in host]).subset(back)
    qual = StringSet.concat([h[1] for h in host]).subset(back)
    reads = Reads(seq, qual, want, enc)
    if resident:
        out = DeviceReads.upload(reads)
        out.names = want
        return out
    return reads


def getBarcodeThresholds(baligned, nmads=3):
    """getBarcodeThresholds (R/getBarcodeThresholds.R:2-15): outlier thresholds on the barcode
    score and on its gap to the next best, median - nmads * MAD (R's mad(): constant 1.4826)."""
    def lower(x):
        x = np.asarray(x, dtype=np.float64)
        med = np.median(x)
        return med - nmads * 1.4826 * np.median(np.abs(x - med))
    return {"score": lower(baligned["score"]), "gap": lower(baligned["gap"])}


# ---------------------------------------------------------------------------
def barcodeAlign(sequences, barcodes, gapOpening=5, gapExtension=1):
    """barcodeAlign (R/barcodeAlign.R:4-40): best barcode, its score, and the gap to the next best.
    The barcode reads are uploaded once and stay resident while every candidate is aligned
    (the reference re-marshals them for each barcode, R/barcodeAlign.R:20-24)."""
    n = len(sequences)
    current = np.full(n, -np.inf)
    nextbest = np.full(n, -np.inf)
    ident = np.full(n, -1, dtype=np.int64)  # NA_integer_
    dev = None
    if calls.__name__.endswith("sarlacc_amd.calls") and n:
        from .resident import DeviceReads
        dev = DeviceReads.upload(sequences)
    for b, bc in enumerate(barcodes):
        if dev is not None:
            calls._string(str(bc), "barcode sequence")
            scores = dev.align_scores(str(bc), gapOpening, gapExtension, local=False)
        else:
            scores = calls.barcode_align(sequences.seq, sequences.qual, sequences.encoding, gapOpening, gapExtension, str(bc))
        keep = scores > current
        second = ~keep & (scores > nextbest)
        ident[keep] = b + 1
        nextbest[keep] = current[keep]
        current[keep] = scores[keep]
        nextbest[second] = scores[second]
    return {"barcode": ident, "score": current, "gap": current - nextbest,
            "metadata": {"gapOpening": gapOpening, "gapExtension": gapExtension, "barcodes": list(barcodes)}}


def qualityAlign(sequences, reference, gapOpening=5, gapExtension=1, edit_only=False):
    """qualityAlign (R/qualityAlign.R:4-27)."""
    ref = str(reference).upper()
    out = calls.general_align(sequences.seq, sequences.qual, sequences.encoding, gapOpening, gapExtension, ref, edit_only)
    res = {"score": out[0], "edit": out[1]}
    if not edit_only:
        res["reference"], res["query"] = out[2], out[3]
    res["metadata"] = {"gapOpening": gapOpening, "gapExtension": gapExtension, "reference": reference}
    return res


# ---------------------------------------------------------------------------
def qualityMask(seq, max_err):
    """qualityMask (R/qualityMask.R:5-16): mask when max_err is given and qualities exist."""
    has_quals = isinstance(seq, Reads) and seq.qual is not None
    if max_err is not None and not (isinstance(max_err, float) and np.isnan(max_err)) and has_quals:
        return StringSet.from_strings(calls.mask_bad_bases(seq.seq, seq.qual, seq.encoding, max_err))
    return seq.seq if isinstance(seq, Reads) else StringSet.from_strings(seq)


def umiGroup(UMI1, threshold1=3, UMI2=None, threshold2=None, max_err=None, groups=None):
    """umiGroup (R/umiGroup.R:2-23): list of clusters of 1-based read indices."""
    if threshold2 is None:
        threshold2 = threshold1
    u1 = qualityMask(UMI1, max_err)
    u2 = qualityMask(UMI2, max_err) if UMI2 is not None else None
    n = len(u1)
    if groups is None:
        by_group = [np.arange(1, n + 1, dtype=np.int32)]
    elif len(groups) and np.ndim(groups[0]) == 0:
        g = np.asarray(groups)
        by_group = [np.flatnonzero(g == lev).astype(np.int32) + 1 for lev in np.unique(g)]  # split(seq_along, groups)
    else:
        by_group = [np.asarray(x, dtype=np.int32) for x in groups]
    return calls.umi_group(u1, threshold1, u2, threshold2, by_group)


def expectedDist(sequences, max_err=None):
    """expectedDist (R/expectedDist.R:2-11)."""
    return calls.compute_lev_masked(qualityMask(sequences, max_err))


# ---------------------------------------------------------------------------
def multiReadAlign(reads, groups, max_error=None, match=0, mismatch=-1, gapOpening=5, gapExtension=1, bandwidth=100):
    """multiReadAlign (R/multiReadAlign.R:7-44).  `groups`: list of 1-based index vectors, or a
    per-read grouping vector.  `max_error` is accepted and ignored like in the reference (App.B Q16)."""
    rd = reads if isinstance(reads, Reads) else Reads(reads)
    n = len(rd)
    names = None
    if len(groups) and np.ndim(groups[0]) == 0:
        g = np.asarray(groups)
        if g.size != n:
            raise ValueError("length of 'reads' and 'groups' should be the same")
        levels = np.unique(g)
        by_group = [np.flatnonzero(g == lev).astype(np.int32) + 1 for lev in levels]
        names = [str(x) for x in levels]
    else:
        by_group = [np.asarray(x, dtype=np.int32) for x in groups]
    aln = calls.quick_msa(by_group, rd.seq, match, mismatch, -gapOpening, -gapExtension, bandwidth)
    out = {"alignments": aln, "names": names}
    if rd.qual is not None:
        q = rd.qual.to_strings()
        out["qualities"] = [[q[i - 1] for i in idx] for idx in by_group]
        out["encoding"] = rd.encoding
    return out


def consensusReadSeq(alignments, pseudo_count=1, min_coverage=0.6):
    """consensusReadSeq (R/consensusReadSeq.R:5-26): quality-weighted vote when qualities are
    present, count-based otherwise.  Returns Reads(consensus, Phred+33 qualities)."""
    aln = alignments["alignments"]
    qual = alignments.get("qualities")
    if qual is not None:
        out = calls.create_consensus_quality_loop(aln, min_coverage, qual, alignments.get("encoding", phred_encoding()))
    else:
        out = calls.create_consensus_basic_loop(aln, min_coverage, pseudo_count)
    return Reads(out[0], out[1], alignments.get("names"))


# ---------------------------------------------------------------------------
# scrambled-control callers: same DP kernel, reads kept resident

def _tied_overlap(real, fake):
    """.tied_overlap (R/tuneAlignment.R:80-87): mean of findInterval with closed/open left ends."""
    fake = np.sort(np.asarray(fake, dtype=np.float64))
    real = np.asarray(real, dtype=np.float64)
    upper = np.searchsorted(fake, real, side="right")
    lower = np.searchsorted(fake, real, side="left")
    return float(((upper + lower) / 2).sum() / (real.size * fake.size))


def _alignment_scores(dev_start, dev_end, adaptor1, adaptor2, go, ge):
    """.get_alignment_scores (R/tuneAlignment.R:103-116) on resident windows."""
    return {"START": dev_start.align_scores(adaptor1, go, ge), "END": dev_end.align_scores(adaptor2, go, ge),
            "RSTART": dev_end.align_scores(adaptor1, go, ge), "REND": dev_start.align_scores(adaptor2, go, ge)}


def tuneAlignment(adaptor1, adaptor2, reads, tolerance=200, number=10000, gapOp_range=(4, 10), gapExt_range=(1, 5), seed=0):
    """tuneAlignment (R/tuneAlignment.R:6-77): grid search of the gap penalties that best separate
    real from scrambled alignment scores.  Reads are sampled, uploaded once and stay in HBM for the
    whole grid (windows and shuffles are built on the device).  `seed` drives the read sample and
    the shuffles (the reference uses R's global RNG)."""
    from .resident import DeviceReads
    adaptor1, adaptor2 = str(adaptor1).upper(), str(adaptor2).upper()
    if isinstance(reads, str):
        reads = read_fastq(reads)
    if len(reads) == 0:
        return {"parameters": {"gapOpening": None, "gapExtension": None}, "scores": {"reads": np.zeros(0), "scrambled": np.zeros(0)}}
    if len(reads) > number:  # FastqSampler(filepath, number)
        keep = np.sort(np.random.default_rng(seed).choice(len(reads), int(number), replace=False))
        reads = reads.subset(keep)
    dev = DeviceReads.upload(reads)
    start, end = dev.front_and_back(tolerance)
    sstart, send = start.scramble(2 * seed + 1), end.scramble(2 * seed + 2)
    go_lo, go_hi = (int(x) for x in np.maximum.accumulate(gapOp_range))
    ge_lo, ge_hi = (int(x) for x in np.maximum.accumulate(gapExt_range))
    best = {"score": 0.0, "go": None, "ge": None, "reads": None, "scrambled": None}
    for go in range(go_lo, go_hi + 1):
        for ge in range(ge_lo, ge_hi + 1):
            r = _alignment_scores(start, end, adaptor1, adaptor2, go, ge)
            x = _alignment_scores(sstart, send, adaptor1, adaptor2, go, ge)
            rs = _resolve_strand(r["START"], r["END"], r["RSTART"], r["REND"])[1]
            xs = _resolve_strand(x["START"], x["END"], x["RSTART"], x["REND"])[1]
            cur = _tied_overlap(rs, xs)
            if best["score"] < cur:
                best = {"score": cur, "go": go, "ge": ge, "reads": rs, "scrambled": xs}
    return {"parameters": {"gapOpening": best["go"], "gapExtension": best["ge"]},
            "scores": {"reads": best["reads"], "scrambled": best["scrambled"]}}


def _compute_threshold(real, scrambled, error):
    """.compute_threshold (R/getAdaptorThresholds.R:94-103)."""
    real = np.sort(np.asarray(real, dtype=np.float64))
    scrambled = np.sort(np.asarray(scrambled, dtype=np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        fdr = (scrambled.size - np.searchsorted(scrambled, real, side="right")) / (real.size - np.arange(1, real.size + 1))
    ok = np.flatnonzero(fdr <= error)
    return float(real[ok.min()]) if ok.size else float("nan")


def getAdaptorThresholds(aligned, reads, error=0.01, seed=0):
    """getAdaptorThresholds (R/getAdaptorThresholds.R:6-66): score thresholds achieving the given
    error rate against scrambled reads.  `aligned` is adaptorAlign's result for `reads`
    (the reference re-streams the FASTQ file named in the metadata; pass the same reads here)."""
    from .resident import DeviceReads
    if isinstance(reads, str):
        reads = read_fastq(reads)
    md1 = aligned["adaptor1"]["metadata"]
    go, ge = md1["gapOpening"], md1["gapExtension"]
    adaptor1, adaptor2 = md1["sequence"], aligned["adaptor2"]["metadata"]["sequence"]
    dev = DeviceReads.upload(reads)
    start, end = dev.front_and_back(aligned["metadata"]["tolerance"])
    x = _alignment_scores(start.scramble(2 * seed + 1), end.scramble(2 * seed + 2), adaptor1, adaptor2, go, ge)
    rev = _resolve_strand(x["START"], x["END"], x["RSTART"], x["REND"])[0]
    scram1 = np.where(rev, x["RSTART"], x["START"])
    scram2 = np.where(rev, x["REND"], x["END"])
    real1, real2 = aligned["adaptor1"]["score"], aligned["adaptor2"]["score"]
    return {"threshold1": _compute_threshold(real1, scram1, error), "threshold2": _compute_threshold(real2, scram2, error),
            "scores1": {"reads": real1, "scrambled": scram1}, "scores2": {"reads": real2, "scrambled": scram2}}


# ---------------------------------------------------------------------------
def extractSubseq(aligned, reads, subseq1=None, subseq2=None):
    """extractSubseq (R/extractSubseq.R:5-117): re-align with the orientation already known and
    pull out the read subsequences opposite the given adaptor positions (1-based inclusive
    `starts` / `ends`).  Like the reference it insists that the re-computed scores equal the
    stored ones -- scores are bit-reproducible here, so equality is exact."""
    if subseq1 is None and subseq2 is None:
        raise ValueError("at least one of 'subseq1' or 'subseq2' must be specified")
    if isinstance(reads, str):
        reads = read_fastq(reads)
    md1, md2 = aligned["adaptor1"]["metadata"], aligned["adaptor2"]["metadata"]
    go, ge = md1["gapOpening"], md1["gapExtension"]
    front, back = _get_front_and_back(reads, aligned["metadata"]["tolerance"])
    flipped = np.asarray(aligned["reversed"], dtype=bool)

    def pick(a, b):  # a[flipped] <- b[flipped]
        fs, bs = a.seq.to_strings(), b.seq.to_strings()
        fq, bq = a.qual.to_strings(), b.qual.to_strings()
        return Reads([y if f else x for x, y, f in zip(fs, bs, flipped)], [y if f else x for x, y, f in zip(fq, bq, flipped)],
                     encoding=reads.encoding)

    out = {}
    for key, sub, rd, md in (("adaptor1", subseq1, pick(front, back), md1), ("adaptor2", subseq2, pick(back, front), md2)):
        if sub is None or (len(sub["starts"]) == 0 and len(sub["ends"]) == 0):
            continue
        res = _align_and_extract(md["sequence"], rd, go, ge, np.asarray(sub["starts"]), np.asarray(sub["ends"]))
        if not np.allclose(res["score"], aligned[key]["score"], rtol=1.5e-8, atol=0):
            raise RuntimeError("score mismatch from 'aligned' for adaptor %s" % key[-1])
        out[key] = res["subseq"]
    return out


# ---------------------------------------------------------------------------
# alignment profiling (R/homopolymerFinder.R, R/homopolymerMatcher.R, R/errorFinder.R).  Pairwise alignments are given
# as the two gapped string sets the R code takes out of a PairwiseAlignmentsSingleSubject object:
# alignedSubject (reference) and alignedPattern (reads); ranges are (start, end) pairs, 1-based inclusive.
def homopolymerFinder(seq):
    """homopolymerFinder (R/homopolymerFinder.R:6-22): per sequence the list of (start, end, base)."""
    s = StringSet.from_strings(seq)
    idx, pos, size, base = calls.find_homopolymers(s)
    out = [[] for _ in range(len(s))]
    for i, p, w, b in zip(idx.tolist(), pos.tolist(), size.tolist(), base):
        out[i].append((p, p + w - 1, b))
    return out


def homopolymerMatcher(ref_aligned, read_aligned):
    """homopolymerMatcher (R/homopolymerMatcher.R:9-36): the homopolymers of the (single) reference and, for each,
    the sorted observed lengths over the alignments."""
    ref, reads = StringSet.from_strings(ref_aligned), StringSet.from_strings(read_aligned)
    if len(ref) == 0 or len({x.replace("-", "") for x in ref.to_strings()}) != 1:
        raise ValueError("alignments should be global and involve a single subject")
    _, positions, obs = calls.match_homopolymers(ref, reads)
    runs = homopolymerFinder(ref.slice(0, 1))[0]
    start_of = {r[0]: k for k, r in enumerate(runs)}
    by_pos = [[] for _ in runs]
    for p, l in zip(positions.tolist(), obs.tolist()):
        by_pos[start_of[p]].append(l)
    return [{"start": a, "end": b, "base": c, "observed": sorted(v)} for (a, b, c), v in zip(runs, by_pos)]


def errorFinder(ref_aligned, read_aligned):
    """errorFinder (R/errorFinder.R:9-49): per reference position (plus one past the end) the counts of A, C, G, T and
    deletions and the sorted insertion lengths (0 for alignments without one there); and the 4 x 4 transition matrix."""
    ref, reads = StringSet.from_strings(ref_aligned), StringSet.from_strings(read_aligned)
    if len(ref) == 0 or len({x.replace("-", "") for x in ref.to_strings()}) != 1:
        raise ValueError("alignments should be global and involve a single subject")
    bases, a, c, g, t, d, ipos, ilen = calls.find_errors(ref, reads)
    n = len(bases)
    ins = [[] for _ in range(n + 1)]
    for p, l in zip(ipos.tolist(), ilen.tolist()):
        ins[p].append(l)
    full = {"base": list(bases) + [None], "A": a.tolist() + [None], "C": c.tolist() + [None], "G": g.tolist() + [None],
            "T": t.tolist() + [None], "deletion": d.tolist() + [None],
            "insertion": [sorted([0] * (len(ref) - len(v)) + v) for v in ins]}
    cols = {"A": a, "C": c, "G": g, "T": t}
    barr = np.frombuffer(bases.encode(), dtype=np.uint8) if n else np.zeros(0, np.uint8)
    transition = np.array([[int(cols[y][barr == ord(x)].sum()) for y in "ACGT"] for x in "ACGT"], dtype=np.int64)
    return {"full": full, "transition": transition}
