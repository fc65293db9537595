"""Literal inputs of the reference's profiling tests (tests/testthat/test-homopolymer.R:30-35,130-175;
test-error.R:39-75), upper-cased as DNAString does, and numpy-free restatements of the R checkers those tests
compare the package against (FINDCHECK :4-27, MATCHCHECK :44-126, CHECKFUN test-error.R:4-36).  Data + test-side
rules only; shared by the oracle tests and the GPU parity tests."""

FIND_SEQS = ["ATGG--GGCCGTTTAA", "ATG-GGGCCGTTTAA", "ATGGGGCCGT-TTAA", "ATGGGGCCGTTTAA-A-A-", "ATGGGGC-CGTTTAA", "---ATGGGGCCGTTTAA"]


def _u(xs):
    return [x.upper() for x in xs]


MATCH_CASES = [
    (_u(["acgtAAAAAtgca", "acgtAA-AAAtgca", "acgtAAAAAtgca", "acgtAAAAAtgca", "acgtAAAAAtgca", "acgt-AAAAAtgca", "acgtAAAAA-tgca"]),
     _u(["acgtAAAAAtgca", "acgtAAAAAAtgca", "acgtA-AAAtgca", "acgt-AAAAtgca", "acgtAAAA-tgca", "acgtAAAAAAtgca", "acgtAAAAAAtgca"])),
    (_u(["acgtAATAAtgca", "acgtTAAAAAtgca", "acgtAAACAtgca", "acgtAAAAGtgca"]),
     _u(["acgtAAAAAtgca", "acgtAAAAAAtgca", "acgtAAAAAtgca", "acgtAAAAAtgca"])),
    (_u(["CCCCCtgca", "--CCCCtgca", "CCCCCCtgca", "CCCC--tgca", "tgcaCCCCC", "tgcaCCCC--", "tgcaCCCCCC", "tgca--CCCC"]),
     _u(["CCCCCtgca", "CCCCCCtgca", "--CCCCtgca", "CCCCCCtgca", "tgcaCCCCC", "tgcaCCCCCC", "tgcaCCCC--", "tgcaCCCCCC"])),
    (_u(["CCCCaGGGaTT", "CCCCaCCCaTT", "CCCCaCCCaCC", "CCCCaGG-aT-", "CCGCaGAGaTC", "---CaGGGaTT", "CCCCaGGGaTTTTT"]),
     _u(["CCCCaGGGaTT", "CCCCaCCCaTT", "CCCCaCCCaCC", "CCCCaGGGaTT", "CCCCaGGGaTT", "CCCCaGGGaTT", "CCCCaGGGa--TT-"])),
    (_u(["actg--G--tt", "actg-----tt", "----tgca", "--A-tgca", "tgca----", "tgca-T--"]),
     _u(["actgGGGGGtt", "actgGGGGGtt", "AAAAtgca", "AAAAtgca", "tgcaTTTT", "tgcaTTTT"])),
    (_u(["actgAAACtttt", "actgAA-Ctttt", "actgA-A-tttt", "actg--A-tttt"]),
     _u(["actgCCCCtttt", "actgCCCCtttt", "actgCCCCtttt", "actgCCCCtttt"])),
]

ERROR_CASES = [
    (_u(["acgactagcacgtcagta", "acgactagcacTtcagta", "Gcgactagcacgtcagta", "acgactagcacgtcagtC", "GcgacCagcGGgtcaCCC"]),
     _u(["acgactagcacgtcagta"] * 5)),
    (_u(["acgactagcac-tcagta", "acgactagcac--cagta", "----ctagcacgtcagta", "acgactagcacgtca---", "-cgac--gc--gtca---"]),
     _u(["acgactagcacgtcagta"] * 5)),
    (_u(["acgactagcacgtcagta", "acgactagcacgtcagta", "acgacctagcacgtcagta", "acgactagcacgtcacga", "acgactggcttttcaatt"]),
     _u(["acgactagcac-tcagta", "acgactagcac--cagta", "----cctagcacgtcagta", "acgactagcacgtca---", "--gac--gca--tcag--"])),
    (_u(["ac--cAagcacgtcaCta", "--gaGGagcacgtcaCCa", "acgacctCCcacgtGa---", "--gacCCCcacgtTTcga", "acGactggcCtttc-att"]),
     _u(["acgactagcac-tcagta", "acgactagcac--cagta", "----cctagcacgtcagta", "acgactagcacgtca---", "--gac--gca--tcag--"])),
]


def _rle(chars):
    runs = []   # (value, start, end), 0-based inclusive, over the list given
    for i, c in enumerate(chars):
        if runs and runs[-1][0] == c:
            runs[-1][2] = i
        else:
            runs.append([c, i, i])
    return runs


def findcheck(seqs):
    """FINDCHECK: run-length encoding of the de-gapped string, runs longer than one base.
    -> (index 0-based, start 1-based, width, base) flattened over the sequences"""
    idx, pos, size, base = [], [], [], []
    for i, s in enumerate(seqs):
        for v, a, b in _rle([c for c in s if c != "-"]):
            if b - a + 1 > 1:
                idx.append(i); pos.append(a + 1); size.append(b - a + 1); base.append(v)
    return idx, pos, size, base


def _runs_with_gaps(seq, minlen):
    index = [i for i, c in enumerate(seq) if c != "-"]
    return index, [(v, a, b) for v, a, b in _rle([seq[i] for i in index]) if b - a + 1 > minlen]


def matchcheck(ref, read):
    """MATCHCHECK for one alignment: per homopolymer of the reference (start position in the de-gapped reference,
    1-based) the longest run of the same base in the read that overlaps the run proper, the read being looked at
    over the run extended by the gap characters around it."""
    ref, read = list(ref), list(read)
    idx, runs = _runs_with_gaps(ref, 1)
    pos, rlen = [], []
    for v, s0, e0 in runs:
        run_start, run_end = idx[s0], idx[e0]
        pre = idx[s0 - 1] + 1 if s0 != 0 else 0
        post = idx[e0 + 1] - 1 if e0 != len(idx) - 1 else len(ref) - 1
        sub = read[pre:post + 1]
        sidx, sruns = _runs_with_gaps(sub, 0)
        best = 0
        for sv, a, b in sruns:
            if sv != v:
                continue
            lo, hi = sidx[a] + pre, sidx[b] + pre
            if lo <= run_end and hi >= run_start:
                best = max(best, b - a + 1)
        pos.append(s0 + 1)
        rlen.append(best)
    return pos, rlen


def checkfun(refs, reads):
    """CHECKFUN: per reference base the read characters opposite it (A, C, G, T, deletion) counted over the
    alignments; insertions per alignment filed under the next reference base (0-based; len(ref) = past the end).
    -> (bases, A, C, G, T, deletion, {position: sorted insertion lengths > 0})"""
    bases = refs[0].replace("-", "")
    n = len(bases)
    cols = {c: [0] * n for c in "ACGT-"}
    ins = {}
    for rf, rd in zip(refs, reads):
        p = 0
        run = 0
        for a, b in zip(rf, rd):
            if a == "-":
                run += 1
                continue
            if run:
                ins.setdefault(p, []).append(run)
                run = 0
            cols[b][p] += 1
            p += 1
        if run:
            ins.setdefault(p, []).append(run)
    return bases, cols["A"], cols["C"], cols["G"], cols["T"], cols["-"], {p: sorted(v) for p, v in ins.items()}
