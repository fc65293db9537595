"""GPU end-to-end: the generic-level pipeline (adaptorAlign -> umiGroup -> multiReadAlign ->
consensusReadSeq, plus barcodeAlign / qualityAlign / expectedDist) run once on the HIP
library and once on the CPU oracle must give identical results at every stage
(scores bit-identical, positions / clusters / alignment rows / consensus + Phred strings
identical).  Includes the strand-flip case of tests/testthat/test-adaptor-align.R:186-212."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A1 = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"
A2 = "CACACTGAGCAGCGACTAGACA"


def run_pipeline(generics, sim):
    from sarlacc_amd.mock import revcomp
    rd = generics.Reads(sim["reads"], sim["quals"])
    aln = generics.adaptorAlign(A1, A2, rd)
    good = np.flatnonzero((aln["adaptor1"]["score"] > 10) & (aln["adaptor2"]["score"] > 5))
    umis = [aln["adaptor1"]["subseq"]["Sub1"][i] for i in good]
    groups = generics.umiGroup(umis, threshold1=2)
    big = [g for g in groups if len(g) >= 3]
    seqs, quals = sim["reads"].to_strings(), sim["quals"].to_strings()
    os_ = [revcomp(seqs[i]) if aln["reversed"][i] else seqs[i] for i in good]
    oq = [quals[i][::-1] if aln["reversed"][i] else quals[i] for i in good]
    msa = generics.multiReadAlign(generics.Reads(os_, oq), big)
    cons = generics.consensusReadSeq(msa)
    cons_basic = generics.consensusReadSeq({"alignments": msa["alignments"]})
    return aln, groups, msa, cons, cons_basic


def test_full_pipeline_identical_to_oracle(monkeypatch):
    from sarlacc_amd import generics
    from sarlacc_amd.mock import mock_reads
    from tests import oracle_calls
    sim = mock_reads(A1, A2, nmolecules=15, nreads=8, seqlen=600, seed=1000)
    got = run_pipeline(generics, sim)
    monkeypatch.setattr(generics, "calls", oracle_calls)
    want = run_pipeline(generics, sim)
    ga, wa = got[0], want[0]
    for ad in ("adaptor1", "adaptor2"):
        assert np.array_equal(ga[ad]["score"].view(np.int64), wa[ad]["score"].view(np.int64))
        assert np.array_equal(ga[ad]["start"], wa[ad]["start"]) and np.array_equal(ga[ad]["end"], wa[ad]["end"])
        assert ga[ad]["subseq"] == wa[ad]["subseq"]
    assert np.array_equal(ga["reversed"], wa["reversed"])
    assert [g.tolist() for g in got[1]] == [g.tolist() for g in want[1]]
    assert got[2]["alignments"] == want[2]["alignments"]
    assert got[3].seq.to_strings() == want[3].seq.to_strings()
    assert got[3].qual.to_strings() == want[3].qual.to_strings()
    assert got[4].seq.to_strings() == want[4].seq.to_strings()
    assert got[4].qual.to_strings() == want[4].qual.to_strings()


def test_strand_flip_case(tmp_path):
    # tests/testthat/test-adaptor-align.R:186-212
    from sarlacc_amd import generics
    from sarlacc_amd.mock import revcomp
    myread = "AACGTAACGTACGTACGTGGGGGGG"
    myqual = "1234567890ABCDEFGHIJKLMNO"
    rd = generics.Reads([myread, revcomp(myread)], [myqual, myqual[::-1]], ["X", "Y"])
    path = str(tmp_path / "x.fastq")
    generics.write_fastq(path, rd)
    out = generics.adaptorAlign("AANNNAA", "CCCCCCC", path)
    assert out["reversed"].tolist() == [False, True]
    for ad in ("adaptor1", "adaptor2"):
        assert out[ad]["score"][0] == out[ad]["score"][1]
        assert out[ad]["start"][0] == out[ad]["start"][1] and out[ad]["end"][0] == out[ad]["end"][1]
    assert out["adaptor1"]["start"][0] == 1 and out["adaptor1"]["end"][0] == 7
    assert out["adaptor2"]["start"][0] == len(myread) and out["adaptor2"]["end"][0] == len(myread) - 7 + 1
    assert out["names"] == ["X", "Y"]
    empty = generics.adaptorAlign("AAAAAAA", "CCCCCCC", generics.Reads([], []))
    assert len(empty["read.width"]) == 0


@pytest.mark.parametrize("tolerance", [250, 25])
def test_adaptor_align_generic_follows_the_reference_tests_rule(tmp_path, tolerance):
    """The check of the reference's own test of the generic (tests/testthat/test-adaptor-align.R:142-184), restated read by
    read and WITHOUT generics.py's code: for every read of adaptorAlign's table, the oriented read (reverse-complemented if
    `reversed`) aligned directly to adaptor 1 gives the row's score / start / end / UMI sub-sequence, its reverse complement
    aligned to adaptor 2 gives the second row with coordinates flipped as width - x + 1 (:180-181); the orientation itself
    follows .resolve_strand (R/adaptorAlign.R:112-122).  The direct alignments are single-read C-ABI calls on host strings
    -- windows, reverse complements and coordinate flips are this test's own (the reference test uses Biostrings'
    pairwiseAlignment there; tolerance 25 adds R/adaptorAlign.R:86-95's windows, which that test never reaches)."""
    import sarlacc_amd
    from sarlacc_amd import calls, generics
    from sarlacc_amd.mock import revcomp
    enc = sarlacc_amd.phred_encoding()
    a1, a2 = "ACGATCAGCTAGNNNNNCGACTAGCTAGCTAG", "CACACTGAGCAGCGACTAGA"
    rng = np.random.default_rng(141002)
    nuc = list("ACGT")
    reads = []
    for i in range(60):
        body = "".join(rng.choice(nuc, int(rng.integers(20, 81))))
        if i % 3 == 1:     # the adaptors where the protocol puts them ...
            body = a1.replace("N", "A")[:int(rng.integers(20, 33))] + body + revcomp(a2)
        if i % 6 == 4:     # ... and on the other strand
            body = revcomp(body)
        reads.append(body)
    quals = ["".join(chr(33 + int(q)) for q in np.round(10 * rng.uniform(1, 5, len(r)))) for r in reads]
    path = str(tmp_path / "r.fastq")
    generics.write_fastq(path, generics.Reads(reads, quals, ["READ_%d" % (i + 1) for i in range(len(reads))]))
    out = generics.adaptorAlign(a1, a2, path, tolerance=tolerance)
    ss, se = [12], [17]    # the run of N in adaptor 1: 0-based start, 1-based end (.setup_subseqs, subseq.starts - 1L)

    def direct(adaptor, seq, qual, sec_s=(), sec_e=()):
        r = calls.adaptor_align([seq], [qual], enc, 5, 1, adaptor, list(sec_s), list(sec_e))
        sub = [seq[int(s[0]) - 1:int(s[0]) - 1 + int(w[0])] for s, w in zip(r[3], r[4])]
        return float(r[0][0]), int(r[1][0]), int(r[2][0]), sub

    assert out["read.width"].tolist() == [len(r) for r in reads]
    n_rev = 0
    for i, (r, q) in enumerate(zip(reads, quals)):
        tol = min(tolerance, len(r))
        front, qfront = r[:tol], q[:tol]
        back, qback = revcomp(r[len(r) - tol:]), q[len(r) - tol:][::-1]
        f = max(direct(a1, front, qfront)[0], 0) + max(direct(a2, back, qback)[0], 0)
        b = max(direct(a1, back, qback)[0], 0) + max(direct(a2, front, qfront)[0], 0)
        assert bool(out["reversed"][i]) == (f < b)
        n_rev += f < b
        query, qquery = (revcomp(r), q[::-1]) if out["reversed"][i] else (r, q)
        s1, st1, en1, sub1 = direct(a1, query[:tol], qquery[:tol], ss, se)
        assert out["adaptor1"]["score"][i] == s1 and out["adaptor1"]["start"][i] == st1 and out["adaptor1"]["end"][i] == en1
        assert out["adaptor1"]["subseq"]["Sub1"][i] == sub1[0]
        s2, st2, en2, _ = direct(a2, revcomp(query[len(r) - tol:]), qquery[len(r) - tol:][::-1])
        assert out["adaptor2"]["score"][i] == s2
        assert out["adaptor2"]["start"][i] == len(r) - st2 + 1 and out["adaptor2"]["end"][i] == len(r) - en2 + 1
    assert 5 <= n_rev <= 55    # both orientations occur


def test_barcode_align_generic_read_by_read():
    """R/barcodeAlign.R:20-36 restated as a loop over single reads and single barcodes (the generic keeps the reads
    resident and reduces on arrays): best barcode = the first to reach the maximum (strict >), the gap is best minus
    next best, where next best starts at -Inf and a score equal to the best one counts as next best."""
    import sarlacc_amd
    from sarlacc_amd import calls, generics
    from sarlacc_amd.mock import random_reads
    enc = sarlacc_amd.phred_encoding()
    seqs, quals = random_reads(30, 8, 16, seed=11, qual_lo=40, qual_hi=80)
    barcodes = ["ACGTACGTACGT", "TTTTGGGGCCCC", "ACGTACGTACGT", "ACGTTGCAACGT"]   # (a duplicate: ties between barcodes)
    out = generics.barcodeAlign(generics.Reads(seqs, quals), barcodes)
    for i, (s, q) in enumerate(zip(seqs, quals)):
        cur, nxt, cid = -np.inf, -np.inf, None
        for b, bc in enumerate(barcodes):
            sc = float(calls.barcode_align([s], [q], enc, 5, 1, bc)[0])
            if sc > cur:
                cid, nxt, cur = b + 1, cur, sc
            elif sc > nxt:
                nxt = sc
        assert out["barcode"][i] == cid and out["score"][i] == cur and out["gap"][i] == cur - nxt
    single = generics.barcodeAlign(generics.Reads(seqs[:3], quals[:3]), barcodes[:1])
    assert np.all(np.isinf(single["gap"])) and single["barcode"].tolist() == [1, 1, 1]


def test_barcode_quality_expected(monkeypatch):
    from sarlacc_amd import generics
    from sarlacc_amd.mock import random_reads
    from tests import oracle_calls
    seqs, quals = random_reads(40, 8, 16, seed=4, qual_lo=40, qual_hi=80)
    rd = generics.Reads(seqs, quals)
    barcodes = ["ACGTACGTACGT", "TTTTGGGGCCCC", "ACGTTGCAACGT"]
    g1 = generics.barcodeAlign(rd, barcodes)
    g2 = generics.qualityAlign(rd, "ACGTACGTAC")
    g3 = generics.expectedDist(rd, max_err=0.01)
    grp = [i % 3 for i in range(40)]

    def grouped(max_err):
        try:
            return [x.tolist() for x in generics.umiGroup(rd, threshold1=2, max_err=max_err, groups=grp)]
        except Exception as e:   # masked UMIs with too many Ns are an error in the reference too (App.B Q10)
            return str(e)

    g4 = [grouped(0.001), grouped(0.01), grouped(None)]
    monkeypatch.setattr(generics, "calls", oracle_calls)
    w1 = generics.barcodeAlign(rd, barcodes)
    w2 = generics.qualityAlign(rd, "ACGTACGTAC")
    w3 = generics.expectedDist(rd, max_err=0.01)
    w4 = [grouped(0.001), grouped(0.01), grouped(None)]
    assert np.array_equal(g1["barcode"], w1["barcode"])
    assert np.array_equal(g1["score"].view(np.int64), w1["score"].view(np.int64))
    assert np.array_equal(g1["gap"].view(np.int64), w1["gap"].view(np.int64))
    assert np.array_equal(g2["score"].view(np.int64), w2["score"].view(np.int64))
    assert g2["reference"] == w2["reference"] and g2["query"] == w2["query"] and np.array_equal(g2["edit"], w2["edit"])
    assert g3.tolist() == w3.tolist()
    assert g4 == w4
    assert isinstance(g4[2], list)


def test_extract_subseq_and_resident_barcodes(monkeypatch):
    """extractSubseq (R/extractSubseq.R) re-aligns with known orientation and must agree with the
    stored scores; constant adaptor regions come back (nearly) as written."""
    from sarlacc_amd import generics
    from sarlacc_amd.mock import mock_reads
    from tests import oracle_calls
    sim = mock_reads(A1, A2, nmolecules=10, nreads=6, seqlen=300, seed=21)
    rd = generics.Reads(sim["reads"], sim["quals"])
    aligned = generics.adaptorAlign(A1, A2, rd)
    sub = generics.extractSubseq(aligned, rd, subseq1={"starts": [1, 22], "ends": [9, 30]}, subseq2={"starts": [5], "ends": [11]})
    assert set(sub) == {"adaptor1", "adaptor2"} and set(sub["adaptor1"]) == {"Sub1", "Sub2"}
    good = aligned["adaptor1"]["score"] > 15
    exact = np.mean([s == A1[:9] for s, g in zip(sub["adaptor1"]["Sub1"], good) if g])
    assert exact > 0.4
    only1 = generics.extractSubseq(aligned, rd, subseq1={"starts": [1], "ends": [9]})
    assert set(only1) == {"adaptor1"} and only1["adaptor1"]["Sub1"] == sub["adaptor1"]["Sub1"]
    with pytest.raises(ValueError):
        generics.extractSubseq(aligned, rd)
    # same through the oracle
    monkeypatch.setattr(generics, "calls", oracle_calls)
    want = generics.extractSubseq(aligned, rd, subseq1={"starts": [1, 22], "ends": [9, 30]}, subseq2={"starts": [5], "ends": [11]})
    assert want == sub


def test_page_locked_result_blocks_are_pooled_and_hold_what_the_device_wrote():
    """Large results come back into page-locked blocks of the library's pool (sarlacc_host_alloc, include/sarlacc_amd.h): a block
    lives as long as any view of it, returns to the pool afterwards and is handed out again; small results stay in ordinary
    memory; freeing a foreign pointer is an error, not a crash."""
    import ctypes as C
    import gc
    from sarlacc_amd import _lib
    from sarlacc_amd.resident import DevBuffer
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    dev = DevBuffer.from_numpy(data)
    a = dev.to_numpy(np.uint8, data.size)
    assert np.array_equal(a, data)
    blk = a.base
    while not isinstance(blk, _lib._HostBlock):
        blk = blk.base
    addr = blk.address
    tail = a[-1000:]            # a view keeps the block
    del a, blk
    gc.collect()
    b = _lib.host_array(3 << 20, np.uint8)   # same size class (4 MB): must be another block while `tail` lives
    assert b.ctypes.data != addr
    assert np.array_equal(tail, data[-1000:])
    del tail, b
    gc.collect()
    c = _lib.host_array(1 << 20, np.int32)   # 4 MB again: one of the two pooled blocks
    c[:] = 7
    first = c.ctypes.data
    del c
    gc.collect()
    d = _lib.host_array(1 << 20, np.int32)
    assert d.ctypes.data == first and d.dtype == np.int32 and d.size == 1 << 20
    small = _lib.host_array(100, np.float64)
    assert small.base is None and small.size == 100
    assert _lib.lib().sarlacc_host_free(C.c_void_p(small.ctypes.data)) != 0
    assert b"did not hand out" in _lib.lib().sarlacc_last_error()
    del d
    gc.collect()
    assert _lib.lib().sarlacc_host_release() == 0
    assert np.array_equal(dev.to_numpy(np.uint8, data.size), data)   # (a fresh block after the release)


def test_two_rank_pipeline_driver_matches_single_rank():
    """tools/run_pipeline.py (BASELINE config 5 in miniature) with two ranks sharing this GPU
    over gloo: read-range DP shards, tile-sharded UMI search + all-gather of neighbour pairs,
    clusters bin-packed over the ranks for MSA + consensus.  Rank 0 recomputes everything
    unsharded (--check): clusters identical, same consensus reads and bases."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SARLACC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "run_pipeline.py"), "--molecules", "400", "--copies", "6",
           "--read-len", "300", "--check"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [x for x in res.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["reads"] == 2400
    chk = out["check"]
    assert chk["clusters_identical"]
    assert out["consensus_reads"] == chk["consensus_reads"] and out["consensus_bases"] == chk["consensus_bases"]
    assert abs(out["score_checksum"] - chk["score_checksum"]) < 1e-6 * max(1.0, abs(chk["score_checksum"]))


def test_bench_two_ranks_prints_the_contract_line():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process; here both ranks
    share the one GPU and the collective runs over gloo): one JSON line from rank 0 with the whole-job value, the
    pipeline pass with its label all-gather, and every rank seen."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SARLACC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", SARLACC_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--reads", "20000", "--read-len", "500", "--molecules", "1500", "--copies", "6", "--no-cpu", "--no-host-pointer"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["reads_per_gpu"] == 20000 and "workload" in out["config"]
    p = out["pipeline"]
    assert p["reads"] == 2 * 1500 * 6 and p["n_ranks_seen"] == 2
    assert p["all_gather"]["backend"] == "gloo" and p["all_gather"]["bytes_received_total"] == 2 * 2 * 4 * 1500 * 6
    assert p["consensus_reads"] > 0 and p["reads_per_min"] > 0 and set(p["rooflines"]) == {"k_msa_pairwise", "k_m2_group", "k_consensus_code"}
    assert p["clusters_all_ranks"] >= p["consensus_reads"]
    assert p["rooflines"]["k_m2_group"]["gather_issue"]["gathers"] > 0 and 0 < p["rooflines"]["k_m2_group"]["frac"] < 1
    # BASELINE configs[4] as worded: the reads of both ranks as ONE pre-group -- tile-sharded search, all-gather of the neighbour
    # pairs, replicated clustering, clusters dealt to the ranks -- identical to what one rank computes
    g = p["giant_group"]
    assert g["reads"] == 2 * 1500 * 6 and g["n_ranks_seen"] == 2 and g["exchange"]["backend"] == "gloo"
    assert g["exchange"]["pairs"] > 0 and g["exchange"]["bytes_received_total"] > 0
    assert g["identical_to_single_rank"] == {"clusters": True, "consensus_of_sampled_clusters": True,
                                             "sampled_clusters": g["identical_to_single_rank"]["sampled_clusters"]}
    assert g["identical_to_single_rank"]["sampled_clusters"] > 0 and g["consensus_reads"] > 0 and g["reads_per_min"] > 0


def test_bench_starts_its_own_ranks_and_refuses_missing_gpus():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself as a
    child process (two ranks sharing this GPU over gloo here) and relays the one JSON line with n_gpus = 2;
    without the sharing switch the same command on a one-GPU box exits non-zero instead of printing n_gpus: 1."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        base.pop(k, None)
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "8000",
            "--read-len", "400", "--molecules", "600", "--copies", "6", "--no-cpu", "--no-host-pointer"]
    env = dict(base, SARLACC_DIST_BACKEND="gloo", SARLACC_BENCH_SHARE_GPU="1")
    res = subprocess.run(args, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["pipeline"]["n_ranks_seen"] == 2
    if torch.cuda.device_count() < 2:
        res = subprocess.run(args, env=base, capture_output=True, text=True, timeout=300)
        assert res.returncode != 0 and not [x for x in res.stdout.splitlines() if x.startswith("{")]
        assert "HIP device" in res.stderr


def test_rccl_label_all_gather_world_size_one():
    """The RCCL code path itself (backend "nccl" IS RCCL on ROCm), which the two-rank tests cannot take on a one-GPU
    box: a process group of one rank, the pipeline's all-gather of cluster labels on device tensors, and the
    gathered row rebuilding the clusters that were sent.  In a child process: a communicator is process-wide state."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from sarlacc_amd import pipeline
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
rng = np.random.default_rng(5)
n = 50000
sizes = rng.integers(1, 12, size=n)
sizes = sizes[np.cumsum(sizes) <= n]
coff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
cmem = (rng.permutation(n)[:coff[-1]] + 1).astype(np.int32)
label, pos = pipeline.labels_from_clusters(coff, cmem, n)
labs, poss, secs, nbytes = pipeline.all_gather_labels(label, pos, dist, torch.device("cuda", 0))
assert labs.shape == (1, n) and nbytes == 0 and secs >= 0
c2, m2 = pipeline.clusters_from_labels(labs[0], poss[0])
assert np.array_equal(c2, coff) and np.array_equal(m2, cmem)
# the giant pre-group's exchange (bench.py pipeline.giant_group): tile search, all-gather of the neighbour pairs on device
# tensors over RCCL, clustering on the gathered pairs -- the clusters of the plain call
from sarlacc_amd import calls, shard
from sarlacc_amd.mock import NUC, mutate
truth = NUC[rng.integers(0, 4, (300, 12))]
umis = [mutate(truth[k // 6], rng, 0.05, 0.01).tobytes().decode() for k in range(1800)]
st = {}
goff, gmem = shard.sharded_umi_group_tiles(umis, 1, calls, dist, torch.device("cuda", 0), flat=True, stats=st)
roff, rmem = calls.umi_group_flat(umis, 1, None, 1, np.array([0, len(umis)], np.int64), np.arange(1, len(umis) + 1, dtype=np.int32))
assert np.array_equal(goff, roff) and np.array_equal(gmem, rmem) and st["pairs_all"] == st["pairs_here"] > 0
t = torch.arange(8, device="cuda", dtype=torch.float64)
dist.all_reduce(t)
torch.cuda.synchronize()
assert t.cpu().tolist() == list(range(8))
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
""" % (root, port)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "rccl ok" in res.stdout, res.stderr[-2000:]
