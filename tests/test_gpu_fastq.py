"""Device FASTQ ingest: the parser on the GPU against the host reader, and
adaptorAlign on a FASTQ path (text parsed, windowed and aligned on the device) against
adaptorAlign on host Reads.  Byte work: everything must be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A1 = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"
A2 = "CACACTGAGCAGCGACTAGACA"


def fastq_text(rng, n, lo, hi, eol="\n", lower=False, final_eol=True):
    recs = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = "".join(rng.choice(list("ACGTN"), L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        if lower and i % 3 == 0:
            s = s.lower()
        q = "".join(chr(int(c)) for c in rng.integers(33, 127, L))
        recs.append("@read_%d some description/%d%s%s%s+%s%s" % (i + 1, i, eol, s, eol, eol, q))
    text = eol.join(recs)
    return (text + (eol if final_eol else "")).encode()


@pytest.mark.parametrize("eol,lower,final_eol,extra", [("\n", False, True, b""), ("\r\n", True, True, b""),
                                                      ("\n", True, False, b""), ("\n", False, True, b"\n\n\r\n")])
def test_device_parser_matches_host_reader(tmp_path, eol, lower, final_eol, extra):
    from sarlacc_amd.generics import read_fastq
    from sarlacc_amd.resident import DeviceReads
    rng = np.random.default_rng(len(eol) * 10 + lower + 2 * final_eol + len(extra))
    # lengths straddle the 8-KB tile and 32-byte thread granules of the line passes; includes empty reads
    text = fastq_text(rng, 300, 0, 700, eol, lower, final_eol) + extra
    path = tmp_path / "r.fastq"
    path.write_bytes(text)
    host = read_fastq(str(path))
    dev = DeviceReads.from_fastq(str(path))
    assert len(dev) == len(host) == 300
    seq, qual = dev.download()
    assert seq.to_strings() == host.seq.to_strings()
    assert qual.to_strings() == host.qual.to_strings()
    assert dev.names == host.names
    # same through bytes / arrays
    d2 = DeviceReads.from_fastq(text)
    assert d2.download()[0].to_strings() == host.seq.to_strings()


def test_device_parser_long_reads_and_empty_input():
    from sarlacc_amd.resident import DeviceReads
    rng = np.random.default_rng(5)
    text = fastq_text(rng, 40, 5000, 30000)
    dev = DeviceReads.from_fastq(text)
    lines = text.decode().split("\n")
    assert dev.download()[0].to_strings() == lines[1::4][:40]
    assert dev.download()[1].to_strings() == lines[3::4][:40]
    for empty in (b"", b"\n", b"\r\n\n"):
        d = DeviceReads.from_fastq(empty)
        assert len(d) == 0 and d.names == []
    # a last record with an empty sequence and quality, with and without its final newlines
    for tail in (b"", b"\n", b"\n\n", b"\r\n\r\n"):
        d = DeviceReads.from_fastq(b"@a\nAC\n+\nII\n@b x\n\n+" + tail)
        assert d.names == ["a", "b x"] and d.download()[0].to_strings() == ["AC", ""]


@pytest.mark.parametrize("text,msg", [
    (b"@a\nACGT\n+\nIIII\nb\nAC\n+\nII\n", "record 2 does not start with '@'"),
    (b"@a\nACGT\n-\nIIII\n", "record 1 has no '+' line"),
    (b"@a\nACGT\n+\nIII\n", "record 1: sequence and quality lengths differ"),
    (b"@a\nACGT\n+\nIIII\n@b\nAC\n", "ends inside a record"),
])
def test_device_parser_errors(text, msg):
    from sarlacc_amd._lib import SarlaccError
    from sarlacc_amd.resident import DeviceReads
    import re
    with pytest.raises(SarlaccError, match=re.escape(msg)):
        DeviceReads.from_fastq(text)


def test_adaptor_align_from_fastq_path(tmp_path):
    """adaptorAlign(filepath) as in the reference (R/adaptorAlign.R:7-37): identical to the
    host-Reads route in every column."""
    from sarlacc_amd import generics as G
    from sarlacc_amd.mock import mock_reads
    sim = mock_reads(A1, A2, nmolecules=20, nreads=6, seqlen=400, seed=1000)
    reads = G.Reads(sim["reads"], sim["quals"], ["READ_%d" % (i + 1) for i in range(len(sim["reads"]))])
    path = tmp_path / "mock.fastq"
    G.write_fastq(str(path), reads)
    a = G.adaptorAlign(A1, A2, reads, tolerance=120)
    b = G.adaptorAlign(A1, A2, str(path), tolerance=120)
    assert b["metadata"]["filepath"] == str(path) and b["names"] == a["names"]
    assert np.array_equal(a["read.width"], b["read.width"]) and np.array_equal(a["reversed"], b["reversed"])
    for key in ("adaptor1", "adaptor2"):
        assert np.array_equal(a[key]["score"].view(np.int64), b[key]["score"].view(np.int64))
        assert np.array_equal(a[key]["start"], b[key]["start"]) and np.array_equal(a[key]["end"], b[key]["end"])
        assert a[key]["subseq"] == b[key]["subseq"]


def test_strand_choice_on_the_device_equals_the_host_rule():
    """sarlacc_dev_choose_strand (.resolve_strand + the row selection of .align_AA_internal, R/adaptorAlign.R:112-122, :190-207)
    against the same rule in numpy on the four downloaded result blocks: mixed strands, ties (both orientations see the same
    windows: never reversed -- the comparison is strict), adaptors with different numbers of sections, an odd number of reads."""
    from sarlacc_amd import generics as G
    from sarlacc_amd.mock import mock_reads
    from sarlacc_amd.resident import DeviceReads
    sim = mock_reads(A1, A2, nmolecules=21, nreads=7, seqlen=300, seed=77)   # 147 reads
    dev = DeviceReads.upload(G.Reads(sim["reads"], sim["quals"]))
    front, back = dev.front_and_back(100)
    sub1, sub2 = G._setup_subseqs(A1), G._setup_subseqs(A2)
    assert len(sub1["starts"]) != len(sub2["starts"])

    def block(ad, d, sb):
        return d.align_block(ad, 5, 1, np.asarray(sb["starts"], dtype=np.int32) - 1, sb["ends"])

    for wins in ((front, back), (front, front)):
        cs, ce = block(A1, wins[0], sub1), block(A2, wins[1], sub2)
        rs, re = block(A1, wins[1], sub1), block(A2, wins[0], sub2)
        rows1, rows2, rev = DeviceReads.choose_strand(cs, ce, rs, re)
        h = {k: DeviceReads.block_to_host(b) for k, b in (("cs", cs), ("ce", ce), ("rs", rs), ("re", re))}
        want_rev, _ = G._resolve_strand(h["cs"][0], h["ce"][0], h["rs"][0], h["re"][0])
        assert rev.dtype == np.bool_ and np.array_equal(rev, want_rev)
        if wins[1] is front:
            assert not rev.any()
        else:
            assert rev.any() and not rev.all()
        for rows, cur, rc in ((rows1, h["cs"], h["rs"]), (rows2, h["ce"], h["re"])):
            assert np.array_equal(rows[0].view(np.int64), np.where(rev, rc[0], cur[0]).view(np.int64))
            assert np.array_equal(rows[1], np.where(rev, rc[1], cur[1])) and np.array_equal(rows[2], np.where(rev, rc[2], cur[2]))
            assert len(rows[3]) == len(cur[3]) == len(rows[4])
            for i in range(len(cur[3])):
                assert np.array_equal(rows[3][i], np.where(rev, rc[3][i], cur[3][i]))
                assert np.array_equal(rows[4][i], np.where(rev, rc[4][i], cur[4][i]))


def test_filter_and_realize_reads(tmp_path):
    """filterReads + realizeReads (R/filterReads.R, R/realizeReads.R) on the adaptorAlign(filepath)
    result: the device orientation + trimming against the same operations on host strings."""
    from sarlacc_amd import generics as G
    from sarlacc_amd.mock import _COMP, mock_reads
    sim = mock_reads(A1, A2, nmolecules=15, nreads=6, seqlen=300, seed=7)
    reads = G.Reads(sim["reads"], sim["quals"], ["READ_%d" % (i + 1) for i in range(len(sim["reads"]))])
    path = tmp_path / "mock.fastq"
    G.write_fastq(str(path), reads)
    aln = G.adaptorAlign(A1, A2, str(path), tolerance=120)
    filt = G.filterReads(aln, 6, 6)
    n0, n1 = len(aln["read.width"]), len(filt["read.width"])
    assert 0 < n1 <= n0
    # restatement of R/filterReads.R on the unfiltered table
    s1, s2 = aln["adaptor1"]["score"], aln["adaptor2"]["score"]
    keep = (s1 >= 6) & (s2 >= 6)
    start = np.where(s1 >= 6, aln["adaptor1"]["end"] + 1, 1)
    end = np.where(s2 >= 6, aln["adaptor2"]["end"] - 1, aln["read.width"])
    keep &= start < end
    assert filt["names"] == [nm for nm, k in zip(aln["names"], keep) if k]
    assert np.array_equal(filt["trim.start"], start[keep]) and np.array_equal(filt["trim.end"], end[keep])
    # realizeReads: reverse-complement + trim
    seqs, quals = reads.seq.to_strings(), reads.qual.to_strings()
    where = {nm: i for i, nm in enumerate(reads.names)}
    comp = {chr(i): chr(_COMP[i]) for i in range(256)}
    got = G.realizeReads(filt)
    assert got.names == filt["names"]
    for k, nm in enumerate(filt["names"]):
        s, q = seqs[where[nm]], quals[where[nm]]
        if filt["reversed"][k]:
            s, q = "".join(comp[c] for c in reversed(s)), q[::-1]
        a, b = int(filt["trim.start"][k]), int(filt["trim.end"][k])
        assert got.seq[k] == s[a - 1:b] and got.qual[k] == q[a - 1:b], (k, nm)
    # untrimmed, resident
    with pytest.warns(UserWarning, match="run 'filterReads' first"):
        whole = G.realizeReads(aln)
    assert [len(x) for x in whole.seq.to_strings()] == aln["read.width"].tolist()
    res = G.realizeReads(filt, resident=True)
    assert res.download()[0].to_strings() == got.seq.to_strings()
    bad = dict(filt, names=["nope"] + filt["names"][1:])
    with pytest.raises(ValueError, match="not present in FASTQ file"):
        G.realizeReads(bad)


@pytest.mark.parametrize("eol,final_eol,extra", [("\n", True, b""), ("\r\n", False, b""), ("\n", True, b"\n\r\n")])
def test_stream_fastq_chunks_equal_the_whole_file(tmp_path, eol, final_eol, extra):
    """FastqStreamer(filepath, n=number) (R/adaptorAlign.R:26): chunks of `number` records, file read in
    blocks that end anywhere (inside a header, a CRLF pair, the last line without newline)."""
    from sarlacc_amd.resident import DeviceReads
    rng = np.random.default_rng(31 + len(eol) + final_eol + len(extra))
    text = fastq_text(rng, 157, 0, 300, eol, False, final_eol) + extra
    path = tmp_path / "s.fastq"
    path.write_bytes(text)
    whole = DeviceReads.from_fastq(text)
    ws, wq = whole.download()
    for number, block in [(1, 1 << 20), (50, 997), (50, 1 << 20), (157, 4096), (1000, 313), (7, 64)]:
        sizes, seqs, quals, names = [], [], [], []
        for chunk in DeviceReads.stream_fastq(str(path), number, block_bytes=block):
            s, q = chunk.download()
            sizes.append(len(chunk))
            seqs += s.to_strings(); quals += q.to_strings(); names += list(chunk.names)
        assert sizes == [number] * (157 // number) + ([157 % number] if 157 % number else []), (number, block)
        assert seqs == ws.to_strings() and quals == wq.to_strings() and names == list(whole.names), (number, block)
    assert list(DeviceReads.stream_fastq(str(path), 10, block_bytes=100)) != []
    empty = tmp_path / "e.fastq"
    empty.write_bytes(b"\n\n")
    assert list(DeviceReads.stream_fastq(str(empty), 10)) == []
    with pytest.raises(ValueError):
        list(DeviceReads.stream_fastq(str(path), 0))


def test_adaptor_align_number_and_qual_type(tmp_path):
    """`number` only changes the chunking, never the table (R/adaptorAlign.R:26-60: per-chunk results are
    rbind-ed); `qual_type` selects the encoding vector the qualities are read with (:18-19)."""
    from sarlacc_amd import generics as G
    from sarlacc_amd.encoding import illumina_encoding, solexa_encoding
    from sarlacc_amd.mock import mock_reads
    sim = mock_reads(A1, A2, nmolecules=12, nreads=5, seqlen=300, seed=77)
    names = ["READ_%d" % (i + 1) for i in range(len(sim["reads"]))]
    reads = G.Reads(sim["reads"], sim["quals"], names)
    path = tmp_path / "mock.fastq"
    G.write_fastq(str(path), reads)
    one = G.adaptorAlign(A1, A2, str(path), tolerance=120)

    def same(a, b):
        assert list(a["names"]) == list(b["names"])
        assert np.array_equal(a["read.width"], b["read.width"]) and np.array_equal(a["reversed"], b["reversed"])
        for key in ("adaptor1", "adaptor2"):
            assert np.array_equal(a[key]["score"].view(np.int64), b[key]["score"].view(np.int64))
            assert np.array_equal(a[key]["start"], b[key]["start"]) and np.array_equal(a[key]["end"], b[key]["end"])
            for k in a[key]["subseq"]:
                assert list(a[key]["subseq"][k]) == list(b[key]["subseq"][k])

    for number in (1, 7, 59, 60, 1e5):
        same(one, G.adaptorAlign(A1, A2, str(path), tolerance=120, number=number))
    assert one["metadata"]["qual.type"] == "phred"
    # realizeReads streams with the same chunking rule and finds reads in whichever chunk holds them
    filt = G.filterReads(one, 6, 6)
    ref = G.realizeReads(filt)
    for number in (1, 13):
        got = G.realizeReads(filt, number=number)
        assert got.names == ref.names and got.seq.to_strings() == ref.seq.to_strings() and got.qual.to_strings() == ref.qual.to_strings()
    res = G.realizeReads(filt, number=13, resident=True)
    assert res.download()[0].to_strings() == ref.seq.to_strings()
    # the same error probabilities written as Illumina (+64) characters: identical table
    q64 = ["".join(chr(min(ord(c) - 33, 62) + 64) for c in q) for q in sim["quals"]]
    q33 = ["".join(chr(min(ord(c) - 33, 62) + 33) for c in q) for q in sim["quals"]]
    p64, p33 = tmp_path / "ill.fastq", tmp_path / "phr.fastq"
    G.write_fastq(str(p64), G.Reads(sim["reads"], q64, names))
    G.write_fastq(str(p33), G.Reads(sim["reads"], q33, names))
    ill = G.adaptorAlign(A1, A2, str(p64), tolerance=120, qual_type="illumina", number=25)
    assert ill["metadata"]["qual.type"] == "illumina"
    same(G.adaptorAlign(A1, A2, str(p33), tolerance=120), ill)
    # Solexa: against the host route with the Solexa encoding vector attached to the Reads
    sol = G.adaptorAlign(A1, A2, str(p64), tolerance=120, qual_type="sol")
    assert sol["metadata"]["qual.type"] == "solexa"
    same(G.adaptorAlign(A1, A2, G.Reads(sim["reads"], q64, names, encoding=solexa_encoding()), tolerance=120), sol)
    assert not np.array_equal(sol["adaptor1"]["score"], ill["adaptor1"]["score"])
    with pytest.raises(ValueError, match="should be one of"):
        G.adaptorAlign(A1, A2, str(path), qual_type="sanger")
    assert illumina_encoding().names[0] == ord("@")
