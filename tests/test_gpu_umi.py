"""GPU parity: masked Levenshtein, neighbour search, greedy clustering and umi_group
against the CPU oracle, through the C ABI.  Everything here is integer work and
must be identical, including the order of neighbours inside each list and the
order of clusters / members in the output.

Mirrors /root/reference/tests/testthat/test-levenshtein.R and test-umicluster.R
(random sequences incl. empty strings and duplicates, limits 1/2/5, masked cases,
random symmetric graphs at several densities, one and two UMIs, pre-groups, solos)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def seqsim(rng, n, lo, hi, alphabet="ACGT"):
    return ["".join(rng.choice(list(alphabet), int(rng.integers(lo, hi + 1)))) for _ in range(n)]


def umisim(rng, n, length, rate=0.1, alphabet="ACGT"):
    ref = rng.choice(list("ACGT"), length)
    out = []
    for _ in range(n):
        t = ref.copy()
        ch = rng.random(length) < rate
        t[ch] = rng.choice(list(alphabet), int(ch.sum()))
        out.append("".join(t))
    return out


def same_lists(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for k, (x, y) in enumerate(zip(a, b)):
        assert list(x) == list(y), (k, list(x), list(y))


@pytest.mark.parametrize("lo,hi", [(1, 20), (5, 10), (0, 5), (28, 32)])
def test_lev_masked_dense(oracle, lo, hi):
    from sarlacc_amd import calls
    rng = np.random.default_rng(1000 + lo)
    seqs = seqsim(rng, 70, lo, hi, "ACGTN" if lo == 5 else "ACGT")
    assert calls.compute_lev_masked(seqs).tolist() == oracle.compute_lev_masked(seqs).tolist()
    ref = "ACAGCTAGC"
    for i in range(len(ref)):
        masked = ref[:i] + "N" + ref[i + 1:]
        assert calls.compute_lev_masked([ref, masked]).tolist() == [0.5]
        assert calls.compute_lev_masked([ref, ref]).tolist() == [0.0]
        assert calls.compute_lev_masked([masked, masked]).tolist() == [0.5]
    assert calls.compute_lev_masked(["ACGT"]).size == 0 and calls.compute_lev_masked([]).size == 0


@pytest.mark.parametrize("lo,hi,dup,alphabet", [(1, 20, False, "ACGT"), (5, 10, False, "ACGT"), (5, 10, True, "ACGT"),
                                               (0, 5, False, "ACGT"), (4, 9, False, "ACGTN"), (22, 32, False, "ACGT")])
def test_fast_levdist(oracle, lo, hi, dup, alphabet):
    from sarlacc_amd import calls
    rng = np.random.default_rng(lo * 100 + hi + dup)
    seqs = seqsim(rng, 50 if dup else 100, lo, hi, alphabet)
    if dup:
        seqs = [seqs[i] for i in rng.integers(0, 50, 100)]
    for limit in (0, 1, 2, 3, 5, 9):
        same_lists(calls.fast_levdist_test(seqs, limit, True), oracle.fast_levdist_test(seqs, limit))
    same_lists(calls.fast_levdist_test(seqs, 2, False), oracle.fast_levdist_test(seqs, 2))


def test_fast_levdist_masked_known_answers():
    from sarlacc_amd import calls
    ref = "ACAGCTAGC"
    for i in range(len(ref)):
        masked = ref[:i] + "N" + ref[i + 1:]
        out = calls.fast_levdist_test([ref, masked], 1)
        assert sorted(out[0].tolist()) == [1, 2] and sorted(out[1].tolist()) == [1, 2]
        out = calls.fast_levdist_test([ref, masked], 0)
        assert out[0].tolist() == [1] and out[1].tolist() == []
    assert calls.fast_levdist_test([], 1) == []


def test_fast_levdist_more_than_one_tile(oracle):
    from sarlacc_amd import calls
    rng = np.random.default_rng(8)
    seqs = []
    for _ in range(90):
        seqs += umisim(rng, 10, 12, rate=0.06)
    perm = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in perm]
    for limit in (1, 2):
        same_lists(calls.fast_levdist_test(seqs, limit), oracle.fast_levdist_test(seqs, limit))


def mockup(rng, n, density):
    a = rng.random((n, n)) < density
    a = np.triu(a, 1)
    a = a | a.T | np.eye(n, dtype=bool)
    return [(np.flatnonzero(a[:, j]) + 1).tolist() for j in range(n)]


@pytest.mark.parametrize("n,density", [(20, 0.05), (20, 0.1), (20, 0.2), (50, 0.2), (50, 0.4), (50, 0.1), (50, 0.0),
                                       (400, 0.01), (1000, 0.003),
                                       # >= 24 links per node: the rounds run on candidate sets (ClusterTop)
                                       (300, 0.3), (1000, 0.05), (2500, 0.03)])
def test_cluster_umis(oracle, n, density):
    from sarlacc_amd import calls
    rng = np.random.default_rng(int(n * 1000 + density * 1000))
    for _ in range(3):
        links = mockup(rng, n, density)
        # neighbour order inside a list matters: shuffle it
        links = [list(rng.permutation(l)) for l in links]
        same_lists(calls.cluster_umis_test(links), oracle.cluster_umis_test(links))


def test_cluster_errors():
    from sarlacc_amd import SarlaccError, calls
    with pytest.raises(SarlaccError, match="zero length read group"):
        calls.cluster_umis_test([[1], []])
    with pytest.raises(SarlaccError, match="single-read groups should contain only the read itself"):
        calls.cluster_umis_test([[2], [1, 2]])
    assert calls.cluster_umis_test([]) == []


def test_umi_group(oracle):
    from sarlacc_amd import calls
    rng = np.random.default_rng(77)
    seqs1, seqs2, pre = [], [], []
    for x in range(10):
        N = int(rng.integers(10, 21))
        seqs1 += umisim(rng, N, 10)
        seqs2 += umisim(rng, N, 5)
        pre += [x] * N
    o = rng.permutation(len(pre))
    seqs1 = [seqs1[i] for i in o]
    seqs2 = [seqs2[i] for i in o]
    pre = np.array(pre)[o]
    everything = [list(range(1, len(seqs1) + 1))]
    by_group = [(np.flatnonzero(pre == g) + 1).tolist() for g in range(10)]
    for groups in (everything, by_group):
        for t in (0, 1, 3):
            same_lists(calls.umi_group(seqs1, t, None, t, groups), oracle.umi_group(seqs1, t, None, t, groups))
        for t1, t2 in ((1, 1), (3, 1), (2, 0)):
            same_lists(calls.umi_group(seqs1, t1, seqs2, t2, groups), oracle.umi_group(seqs1, t1, seqs2, t2, groups))
    out = calls.umi_group(seqs1[:10], 1, None, 1, [[i] for i in range(1, 11)])
    assert [c.tolist() for c in out] == [[i] for i in range(1, 11)]
    # Rd examples (man/umiGroup.Rd:49-65), outputs recorded from the reference in SURVEY section 8c
    u1 = ["AACCGGTT", "AACGGTT", "ACCCGGTT", "AACCGGTTT"]
    u2 = ["AACCGGTT", "CACCGGTT", "AACCCGGTA", "AACGGGTT"]
    g = [[1, 2, 3, 4]]
    assert [c.tolist() for c in calls.umi_group(u1, 3, None, 3, g)] == [[1, 4, 2, 3]]
    assert [c.tolist() for c in calls.umi_group(u1, 0, None, 0, g)] == [[1], [2], [3], [4]]
    assert [c.tolist() for c in calls.umi_group(u1, 3, u2, 3, g)] == [[3, 1, 4, 2]]


def test_umi_group_masked_and_errors(oracle):
    from sarlacc_amd import SarlaccError, calls
    rng = np.random.default_rng(5)
    seqs = []
    for _ in range(12):
        seqs += umisim(rng, 8, 12, rate=0.08, alphabet="ACGTN")
    g = [list(range(1, len(seqs) + 1))]
    for t in (1, 2, 3):
        try:
            want = oracle.umi_group(seqs, t, None, t, g)
        except oracle.OracleError as e:
            with pytest.raises(SarlaccError, match=str(e)):
                calls.umi_group(seqs, t, None, t, g)
            continue
        same_lists(calls.umi_group(seqs, t, None, t, g), want)
    # a UMI with more Ns than 2*limit is not its own neighbour (App.B Q10)
    with pytest.raises(SarlaccError, match="zero length read group"):
        calls.umi_group(["ACNNNGT", "ACGTTGT"], 1, None, 1, [[1, 2]])
    with pytest.raises(SarlaccError, match="should have the same length"):
        calls.umi_group(["ACGT", "ACGA"], 1, ["ACGT"], 1, [[1, 2]])


def test_c3_shape_sample(oracle):
    # BASELINE config 3 shape on a sample the oracle finishes in seconds: 12-bp UMIs,
    # 10 reads per molecule with the mockReads error process, one pre-group
    from sarlacc_amd import calls
    from sarlacc_amd.mock import NUC, mutate
    rng = np.random.default_rng(1000)
    umis = []
    for _ in range(400):
        truth = NUC[rng.integers(0, 4, 12)]
        umis += [mutate(truth, rng).tobytes().decode() for _ in range(10)]
    g = [list(range(1, len(umis) + 1))]
    for t in (1, 2, 3):   # 3 is umiGroup's own default (R/umiGroup.R:3)
        got = calls.umi_group(umis, t, None, t, g)
        same_lists(got, oracle.umi_group(umis, t, None, t, g, fast=True))
        assert sorted(x for c in got for x in c.tolist()) == g[0]


def test_dense_neighbourhoods_both_round_schemes(oracle):
    # short UMIs at threshold 3: hundreds of neighbours per UMI, a few picks per round.  The candidate-set rounds and
    # the rounds that walk every list give the oracle's clusters, in its order.
    from sarlacc_amd import _lib, calls
    rng = np.random.default_rng(4242)
    umis = ["".join(rng.choice(list("ACGT"), 8)) for _ in range(900)]
    umis = [u for u in umis for _ in range(int(rng.integers(1, 8)))]
    umis = [umis[i] for i in rng.permutation(len(umis))]
    g = [list(range(1, len(umis) + 1))]
    want = oracle.umi_group(umis, 3, None, 3, g, fast=True)
    try:
        for full in (0, 1):
            calls.set_option("umi_full_rounds", full)
            same_lists(calls.umi_group(umis, 3, None, 3, g), want)
            assert _lib.stage_count("umi_links") >= 24 * len(umis)
            if not full:
                assert _lib.stage_count("umi_cluster_candidate_rounds") > 0
    finally:
        calls.set_option("umi_full_rounds", 0)


def test_candidate_rounds_cut_off_list_falls_back_to_a_full_round(oracle):
    # 2 500 copies of one UMI all carry the same count: the candidate list of the first round (at most max(1024, n / 8)
    # entries) is cut off, the device parks its control block and the host runs that round over every list; the
    # candidate rounds then go on with what is left (a second family of neighbours, some loners)
    from sarlacc_amd import _lib, calls
    rng = np.random.default_rng(77)
    umis = ["ACGTTGCAAC"] * 2500 + ["ACGTTGCAAG"] * 40 + ["TTTTGGGGCC"] * 300 + ["".join(rng.choice(list("ACGT"), 10)) for _ in range(200)]
    umis = [umis[i] for i in rng.permutation(len(umis))]
    g = [list(range(1, len(umis) + 1))]
    for t in (1, 2):
        want = oracle.umi_group(umis, t, None, t, g, fast=True)
        same_lists(calls.umi_group(umis, t, None, t, g), want)
        assert _lib.stage_count("umi_links") >= 24 * len(umis)
        assert _lib.stage_count("umi_cluster_full_rounds") >= 1 and _lib.stage_count("umi_cluster_candidate_rounds") >= 1


def noisy_umis(rng, molecules, lo, hi, n_rate=0.0, short=0):
    """Copies of random molecules with substitutions, insertions and deletions; optionally Ns and a few short strings."""
    out = []
    for _ in range(molecules):
        ref = list(rng.choice(list("ACGT"), int(rng.integers(lo, hi + 1))))
        for _ in range(int(rng.integers(1, 7))):
            t = list(ref)
            for _ in range(int(rng.integers(0, 3))):
                r = rng.random()
                if r < 0.4:
                    t[int(rng.integers(0, len(t)))] = str(rng.choice(list("ACGT")))
                elif r < 0.7 and len(t) > lo:
                    del t[int(rng.integers(0, len(t)))]
                elif len(t) < 32:
                    t.insert(int(rng.integers(0, len(t) + 1)), str(rng.choice(list("ACGT"))))
            t = [("N" if rng.random() < n_rate else c) for c in t]
            out.append("".join(t))
    for _ in range(short):
        out.append("".join(rng.choice(list("ACGT"), int(rng.integers(1, lo)))))
    return [out[i] for i in rng.permutation(len(out))]


@pytest.mark.parametrize("lo,hi,n_rate,short,split", [(12, 12, 0.0, 0, 1), (8, 11, 0.0, 0, 1), (9, 16, 0.01, 5, 1), (16, 20, 0.0, 3, 1),
                                                      (10, 10, 0.02, 0, 1), (24, 32, 0.005, 4, 1),
                                                      (10, 10, 0.06, 0, 0)])   # a quarter of the strings with an N: tile search
def test_split_key_search_small_sets(oracle, lo, hi, n_rate, short, split):
    # the candidate search of thresholds 2 and 3 (k_sk_scan), forced onto sets the oracle's trie walks in no time:
    # the neighbour lists themselves, then the groups with and without pre-groups
    from sarlacc_amd import SarlaccError, _lib, calls
    rng = np.random.default_rng(lo * 100 + hi)
    try:
        calls.set_option("umi_split_min", 64)
        for rep in range(3):
            umis = noisy_umis(rng, 150, lo, hi, n_rate, short)
            for t in (1, 2, 3):
                want = oracle.fast_levdist_test(umis, t)
                for single in (0, 1):   # two candidate columns per lane where every string is shorter than 16 bases, and one
                    calls.set_option("umi_scan_single", single)
                    got = calls.fast_levdist_test(umis, t)
                    assert _lib.stage_count("umi_split_search") == split
                    if split:
                        # (the rows of up to 15 - t bases scan two columns per lane)
                        assert (_lib.stage_count("umi_scan_two_columns") > 0) == (any(8 <= len(u) <= 15 - t for u in umis) and not single)
                    same_lists(got, want)
                calls.set_option("umi_scan_single", 0)
            n = len(umis)
            pre = rng.integers(0, 2, n)
            for groups in ([list(range(1, n + 1))], [(np.flatnonzero(pre == g) + 1).tolist() for g in range(2)]):
                for t in (1, 2, 3):
                    try:
                        want = oracle.umi_group(umis, t, None, t, groups, fast=True)
                    except oracle.OracleError as e:
                        with pytest.raises(SarlaccError, match=str(e)):
                            calls.umi_group(umis, t, None, t, groups)
                        continue
                    same_lists(calls.umi_group(umis, t, None, t, groups), want)
    finally:
        calls.set_option("umi_split_min", 0)
        calls.set_option("umi_scan_single", 0)


@pytest.mark.parametrize("threshold,n", [(1, 100000), (2, 100000), (3, 40000)])
def test_split_key_search_equals_tile_search_and_oracle(oracle, threshold, n):
    # BASELINE config 3's shape (12-base UMIs, 10 reads per molecule, mockReads errors) from the size on where the split-key
    # search takes over by itself: the same groups as the all-tile-pairs search and as the oracle
    from sarlacc_amd import _lib, calls
    from sarlacc_amd.mock import NUC, mutate
    rng = np.random.default_rng(31 + threshold)
    umis = []
    for _ in range(n // 10):
        truth = NUC[rng.integers(0, 4, 12)]
        umis += [mutate(truth, rng).tobytes().decode() for _ in range(10)]
    g = [list(range(1, len(umis) + 1))]
    try:
        got = calls.umi_group(umis, threshold, None, threshold, g)   # the split-key search by itself (>= 32 768 strings)
        assert _lib.stage_count("umi_split_search") == 1
        calls.set_option("umi_tile_search", 1)
        tiles = calls.umi_group(umis, threshold, None, threshold, g)
        assert _lib.stage_count("umi_split_search") == 0
    finally:
        calls.set_option("umi_tile_search", 0)
    same_lists(got, tiles)
    same_lists(got, oracle.umi_group(umis, threshold, None, threshold, g, fast=True))


def test_mask_bad_bases(oracle, oenc, enc):
    from sarlacc_amd import SarlaccError, calls
    from sarlacc_amd.mock import random_reads
    seqs, quals = random_reads(60, 0, 60, seed=3)
    for thr in (0.001, 0.01, 0.05, 0.1, 0.5):
        assert calls.mask_bad_bases(seqs, quals, enc, thr) == oracle.mask_bad_bases(seqs, quals, oenc, thr)
    assert calls.mask_bad_bases([], [], enc, 0.1) == []
    with pytest.raises(SarlaccError, match="same length"):
        calls.mask_bad_bases(["ACGT"], ["III"], enc, 0.1)
    with pytest.raises(SarlaccError, match="quality cannot be lower"):
        calls.mask_bad_bases(["ACGT"], ["II I"], enc, 0.1)


@pytest.mark.parametrize("split_min", [0, 64])
def test_tile_sharded_pairs_reproduce_umi_group(oracle, split_min):
    """The row-tile shards of the pair search (what each GPU of a node would compute) put together give exactly
    umi_group's result; shard boundaries balance the triangular work.  split_min 64: thresholds 2 and 3 through the
    split-key search, whose scans report only the rows of the shard."""
    from sarlacc_amd import _lib, calls
    rng = np.random.default_rng(21)
    umis = []
    for _ in range(300):
        umis += umisim(rng, 8, 12, rate=0.06)
    umis = [umis[i] for i in rng.permutation(len(umis))]
    g = [list(range(1, len(umis) + 1))]
    try:
        calls.set_option("umi_split_min", split_min)
        for limit in (1, 2, 3):
            want = calls.umi_group(umis, limit, None, limit, g)
            same_lists(want, oracle.umi_group(umis, limit, None, limit, g, fast=True))
            for world in (1, 3, 8):
                parts = [calls.umi_pairs_shard(umis, limit, r, world) for r in range(world)]
                assert _lib.stage_count("umi_split_search") == (1 if split_min else 0)
                allp = np.concatenate(parts)
                assert len(np.unique(allp)) == allp.size            # no pair is found twice
                same_lists(calls.umi_group_from_pairs(umis, limit, allp), want)
                if world == 8:
                    sizes = [p.size for p in parts]
                    assert max(sizes) > 0
    finally:
        calls.set_option("umi_split_min", 0)


@pytest.mark.parametrize("length", [50, 200])
def test_tile_sharded_pairs_of_long_strings(oracle, length):
    """The same for strings beyond one code word (4 words up to 128 bases, word planes from HBM beyond): the shards of the
    multi-GPU search put together give umi_group's result."""
    from sarlacc_amd import calls
    rng = np.random.default_rng(length)
    umis = []
    for _ in range(90):
        umis += umisim(rng, 8, length, rate=0.02)
    umis = [umis[i] for i in rng.permutation(len(umis))]
    g = [list(range(1, len(umis) + 1))]
    for limit in (1, 4):
        want = calls.umi_group(umis, limit, None, limit, g)
        same_lists(want, oracle.umi_group(umis, limit, None, limit, g))
        for world in (1, 3):
            parts = [calls.umi_pairs_shard(umis, limit, r, world) for r in range(world)]
            allp = np.concatenate(parts)
            assert len(np.unique(allp)) == allp.size
            same_lists(calls.umi_group_from_pairs(umis, limit, allp), want)


def test_pair_exchange_with_the_pairs_kept_on_the_device(oracle):
    """The device-resident form of the exchange (sarlacc_dev_umi_pairs_shard / _fetch / sarlacc_dev_umi_group_from_pairs: what
    shard.sharded_umi_group_tiles runs over RCCL): the shards' pairs, fetched into device tensors and put together on the
    device, are the host entry points' pairs and give umi_group's clusters; misuse is refused with a message."""
    import torch
    from sarlacc_amd import _lib, calls
    from sarlacc_amd._lib import SarlaccError
    rng = np.random.default_rng(33)
    umis = []
    for _ in range(260):
        umis += umisim(rng, 8, 12, rate=0.06)
    umis = [umis[i] for i in rng.permutation(len(umis))]
    g = [list(range(1, len(umis) + 1))]
    for limit in (1, 2):
        want = calls.umi_group(umis, limit, None, limit, g)
        same_lists(want, oracle.umi_group(umis, limit, None, limit, g, fast=True))
        for world in (1, 3):
            parts = []
            for r in range(world):
                m = calls.dev_umi_pairs_shard(umis, limit, r, world)
                t = torch.full((m + 3,), -1, dtype=torch.int64, device="cuda")
                calls.dev_umi_pairs_fetch(t, m + 3)
                assert bool((t[m:] == -1).all())                      # nothing written beyond the count
                assert np.array_equal(np.sort(t[:m].cpu().numpy().view(np.uint64)), np.sort(calls.umi_pairs_shard(umis, limit, r, world)))
                parts.append(t[:m])
            allp = torch.cat(parts)
            same_lists(calls.dev_umi_group_from_pairs(umis, limit, allp, allp.numel()), want)
            co, cm = calls.dev_umi_group_from_pairs(umis, limit, allp, allp.numel(), flat=True)
            assert [cm[co[k]:co[k + 1]].tolist() for k in range(co.size - 1)] == [list(c) for c in want]
    # a fetch that does not follow a shard search, a buffer that is too small, a malformed pair
    calls.umi_group(umis, 1, None, 1, g)
    with pytest.raises(SarlaccError, match="no neighbour pairs to fetch"):
        calls.dev_umi_pairs_fetch(torch.zeros(8, dtype=torch.int64, device="cuda"), 8)
    m = calls.dev_umi_pairs_shard(umis, 1, 0, 1)
    assert m > 8
    with pytest.raises(SarlaccError, match="pair buffer too small"):
        calls.dev_umi_pairs_fetch(torch.zeros(8, dtype=torch.int64, device="cuda"), 8)
    bad = torch.tensor([(5 << 32) | 5], dtype=torch.int64, device="cuda")   # i < j is required
    with pytest.raises(SarlaccError, match="malformed neighbour pair"):
        calls.dev_umi_group_from_pairs(umis, 1, bad, 1)
    with pytest.raises(SarlaccError, match="malformed neighbour pair"):
        calls.umi_group_from_pairs(umis, 1, np.array([(5 << 32) | len(umis)], np.uint64))
    # no pairs at all: every read alone (and a self link each, threshold 0 on distinct strings aside)
    solo = calls.dev_umi_group_from_pairs(["ACGTACGTAAAA", "TTTTGGGGCCCC", "GAGAGAGAGAGA"], 1, None, 0)
    same_lists(solo, calls.umi_group(["ACGTACGTAAAA", "TTTTGGGGCCCC", "GAGAGAGAGAGA"], 1, None, 1, [[1, 2, 3]]))
    m = calls.dev_umi_pairs_shard(umis, 1, 0, 1)
    assert _lib.release_umi_workspace() > 8 * m and _lib.release_umi_workspace() == 0   # (the stage's buffers only, and only once)
    with pytest.raises(SarlaccError, match="are gone"):
        calls.dev_umi_pairs_fetch(torch.zeros(m, dtype=torch.int64, device="cuda"), m)
    same_lists(calls.umi_group(umis, 1, None, 1, g), oracle.umi_group(umis, 1, None, 1, g, fast=True))   # and the stage simply allocates again
    _lib.lib().sarlacc_release_workspace()
    with pytest.raises(SarlaccError, match="no neighbour pairs to fetch|are gone"):
        calls.dev_umi_pairs_fetch(torch.zeros(8, dtype=torch.int64, device="cuda"), 8)


def test_umi_group_flat_matches_lists():
    """CSR in / CSR out variant used by the large-batch pipeline: same clusters, same order."""
    from sarlacc_amd import calls
    rng = np.random.default_rng(77)
    umis = umisim(rng, 400, 12, 0.15)
    pre = [list(range(1, 151)), [151], list(range(152, 401))]
    want = calls.umi_group(umis, 1, None, 1, pre)
    poff = np.cumsum([0] + [len(p) for p in pre])
    coff, mem = calls.umi_group_flat(umis, 1, None, 1, poff, np.concatenate(pre))
    got = [mem[coff[k]:coff[k + 1]] for k in range(len(coff) - 1)]
    same_lists(got, want)
    keep = np.diff(coff) >= 2
    soff, smem = calls.csr_select(coff, mem, keep)
    same_lists([smem[soff[k]:soff[k + 1]] for k in range(len(soff) - 1)], [w for w in want if len(w) >= 2])


# ---------------------------------------------------------------------------
# strings beyond one 64-bit code word (33..128 bases): UMIs can be up to 80 bases long in the reference's
# own tests (tests/testthat/test-umigroup.R), and expectedDist runs on whole adaptor sub-sequences

@pytest.mark.parametrize("lo,hi,alphabet", [(33, 48, "ACGT"), (60, 80, "ACGTN"), (100, 128, "ACGT"), (5, 70, "ACGT")])
def test_lev_masked_dense_long(oracle, lo, hi, alphabet):
    from sarlacc_amd import calls
    rng = np.random.default_rng(2000 + lo)
    seqs = seqsim(rng, 20, lo, hi, alphabet) + umisim(rng, 20, hi, 0.1, alphabet)
    assert calls.compute_lev_masked(seqs).tolist() == oracle.compute_lev_masked(seqs).tolist()


@pytest.mark.parametrize("length,alphabet", [(33, "ACGT"), (40, "ACGTN"), (64, "ACGT"), (65, "ACGT"), (80, "ACGTN"), (128, "ACGT")])
def test_fast_levdist_long(oracle, length, alphabet):
    """the neighbour lists (content and trie order) for strings of 33..128 bases, lengths varying around
    `length` so that the trie order is decided beyond base 32, 42, 63 ... (sort key boundaries)"""
    from sarlacc_amd import calls
    rng = np.random.default_rng(length)
    seqs = []
    for _ in range(12):
        fam = umisim(rng, 8, length, rate=0.04, alphabet=alphabet)
        for k, u in enumerate(fam):     # indels near the end, a few shorter relatives
            if k % 3 == 1:
                u = u[:-1]
            if k % 4 == 2:
                u = u[:length // 2] + u[length // 2 + 1:]
            seqs.append(u[:128])
    seqs += seqs[:5]                     # duplicates
    seqs += seqsim(rng, 10, 1, 20, "ACGT")   # short strings in the same call
    for limit in (0, 1, 2, 3, 5, 9, 20):
        same_lists(calls.fast_levdist_test(seqs, limit, True), oracle.fast_levdist_test(seqs, limit))


def test_umi_group_long(oracle):
    from sarlacc_amd import SarlaccError, calls
    rng = np.random.default_rng(4242)
    seqs1, seqs2, pre = [], [], []
    for x in range(8):
        N = int(rng.integers(10, 21))
        seqs1 += umisim(rng, N, 50, rate=0.03)
        seqs2 += umisim(rng, N, 36, rate=0.03)
        pre += [x % 3] * N
    o = rng.permutation(len(pre))
    seqs1 = [seqs1[i] for i in o]
    seqs2 = [seqs2[i] for i in o]
    pre = np.array(pre)[o]
    everything = [list(range(1, len(seqs1) + 1))]
    by_group = [(np.flatnonzero(pre == g) + 1).tolist() for g in range(3)]
    for groups in (everything, by_group):
        for t in (0, 1, 3):
            same_lists(calls.umi_group(seqs1, t, None, t, groups), oracle.umi_group(seqs1, t, None, t, groups))
        for t1, t2 in ((1, 1), (3, 1)):
            same_lists(calls.umi_group(seqs1, t1, seqs2, t2, groups), oracle.umi_group(seqs1, t1, seqs2, t2, groups))
    # more than one tile of long strings
    many = []
    for _ in range(60):
        many += umisim(rng, 10, 40, rate=0.03)
    g = [list(range(1, len(many) + 1))]
    same_lists(calls.umi_group(many, 2, None, 2, g), oracle.umi_group(many, 2, None, 2, g))
    with pytest.raises(SarlaccError, match="longer than 1024 bases"):
        calls.umi_group(["A" * 1025, "C" * 10], 1, None, 1, [[1, 2]])


@pytest.mark.parametrize("length,alphabet", [(129, "ACGT"), (160, "ACGTN"), (300, "ACGT"), (1024, "ACGT")])
def test_strings_beyond_128_bases(oracle, length, alphabet):
    """Strings of 129..1 024 bases (src/sorted_trie.cpp:39-71 takes any length): as many code words as the longest string
    needs, read from HBM (k_umi_pairs_long<K, true>).  Neighbour lists in trie order -- sort keys of 21 bases, so the order
    is decided up to 49 keys deep --, the dense distances and the groups, with shorter strings in the same call."""
    from sarlacc_amd import calls
    rng = np.random.default_rng(length)
    seqs = []
    for _ in range(10):
        fam = umisim(rng, 8, length, rate=0.01, alphabet=alphabet)
        for k, u in enumerate(fam):     # indels near the end and in the middle, a few shorter relatives
            if k % 3 == 1:
                u = u[:-1]
            if k % 4 == 2:
                u = u[:length // 2] + u[length // 2 + 1:]
            seqs.append(u[:length])
    seqs += seqs[:5]                     # duplicates
    seqs += seqsim(rng, 10, 1, 60, "ACGT")   # short strings in the same call
    for limit in (0, 1, 2, 3, 5, 9, 20):
        same_lists(calls.fast_levdist_test(seqs, limit, True), oracle.fast_levdist_test(seqs, limit))
    sub = seqs[:24] + seqs[-6:]
    assert calls.compute_lev_masked(sub).tolist() == oracle.compute_lev_masked(sub).tolist()
    second = [u[:40] for u in seqs]
    everything = [list(range(1, len(seqs) + 1))]
    halves = [list(range(1, len(seqs) // 2 + 1)), list(range(len(seqs) // 2 + 1, len(seqs) + 1))]
    for groups in (everything, halves):
        for t in (1, 4, 12):
            same_lists(calls.umi_group(seqs, t, None, t, groups), oracle.umi_group(seqs, t, None, t, groups))
        same_lists(calls.umi_group(seqs, 6, second, 2, groups), oracle.umi_group(seqs, 6, second, 2, groups))


def test_singleton_pregroups_pass_through_unchecked(oracle):
    """src/umi_group.cpp:39-42: a pre-group of one read is returned as it is, whatever its UMI holds
    (too long for the search, characters outside ACGTN, empty)."""
    from sarlacc_amd import calls
    umis = ["ACGT", "ACGA", "X" * 200, "", "ACNNNNNNGT", "ACGT"]
    groups = [[1, 2, 6], [3], [4], [5]]
    same_lists(calls.umi_group(umis, 1, None, 1, groups), oracle.umi_group(umis, 1, None, 1, groups))
    same_lists(calls.umi_group(umis, 1, umis, 1, groups), oracle.umi_group(umis, 1, umis, 1, groups))
