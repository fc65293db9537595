"""N > 1 path on CPU: two processes over gloo.  The compute module is the CPU oracle
(tests/oracle_calls.py) -- what is under test is the sharding logic of sarlacc_amd.shard:
reads split like .parallelize (R/adaptorAlign.R:126-134), pre-groups bin-packed over ranks,
and the all-gather of UMI cluster labels reproducing the unsharded output exactly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_umis(seed):
    rng = np.random.default_rng(seed)
    from tests.test_oracle_umi import umisim
    seqs, pre = [], []
    for x in range(9):
        N = int(rng.integers(6, 30))
        seqs += umisim(rng, N, 10)
        pre += [x] * N
    o = rng.permutation(len(pre))
    seqs = [seqs[i] for i in o]
    pre = np.array(pre)[o]
    groups = [(np.flatnonzero(pre == g) + 1).astype(np.int32) for g in range(9)]
    solo = groups[0][-1:]                       # a solo pre-group, split off group 0
    groups[0] = groups[0][:-1]
    groups.insert(3, solo)
    return seqs, groups


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sarlacc_amd import shard
    from tests import oracle_calls
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        seqs, groups = _make_umis(5)
        out = shard.sharded_umi_group(seqs, 1, None, 1, groups, oracle_calls, dist)
        # reads sharded contiguously: adaptor_align on the local range only
        from sarlacc_amd.mock import random_reads
        from oracle import oracle as O
        reads, quals = random_reads(41, 0, 60, seed=9)
        lo, hi, res = shard.sharded_over_reads(
            lambda a, b: O.adaptor_align(reads[a:b], quals[a:b], O.phred_encoding(), 5, 1, "ACGTNNACGT", [4], [6]), len(reads), dist)
        q.put((rank, [c.tolist() for c in out], lo, hi, res[0].tolist(), res[1].tolist(), res[3][0].tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    from oracle import oracle as O
    from tests import oracle_calls
    seqs, groups = _make_umis(5)
    want = [c.tolist() for c in oracle_calls.umi_group(seqs, 1, None, 1, groups)]
    assert results[0][1] == want and results[1][1] == want
    # read shards tile the batch and reproduce the unsharded result
    from sarlacc_amd.mock import random_reads
    reads, quals = random_reads(41, 0, 60, seed=9)
    full = O.adaptor_align(reads, quals, O.phred_encoding(), 5, 1, "ACGTNNACGT", [4], [6])
    assert results[0][2] == 0 and results[0][3] == results[1][2] and results[1][3] == len(reads)
    assert results[0][4] + results[1][4] == full[0].tolist()
    assert results[0][5] + results[1][5] == full[1].tolist()
    assert results[0][6] + results[1][6] == full[3][0].tolist()


def test_partition_helpers():
    from sarlacc_amd import shard
    # .parallelize: seq(1, n, length.out = parts + 1) boundaries
    assert shard.contiguous_bounds(10, 1).tolist() == [0, 10]
    b = shard.contiguous_bounds(10, 3)
    assert b[0] == 0 and b[-1] == 10 and (np.diff(b) > 0).all()
    assert b.tolist() == [0, 3, 6, 10]
    for n, parts in ((1000, 8), (7, 8), (0, 4), (100001, 8)):
        b = shard.contiguous_bounds(n, parts)
        assert b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all()
    own = shard.assign_groups([100, 1, 1, 1, 90, 50, 50], 2)
    loads = [sum(s * s for s, o in zip([100, 1, 1, 1, 90, 50, 50], own) if o == r) for r in range(2)]
    assert max(loads) <= 1.35 * min(loads)
    assert shard.assign_groups([], 4).size == 0


def _worker_tiles(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sarlacc_amd import shard
    from tests import oracle_calls
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from tests.test_oracle_umi import umisim
        rng = np.random.default_rng(3)
        umis = []
        for _ in range(70):
            umis += umisim(rng, 9, 10)
        out = shard.sharded_umi_group_tiles(umis, 1, oracle_calls, dist)
        q.put((rank, [c.tolist() for c in out]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_tile_sharding_of_one_giant_group():
    """Row tiles of the all-pairs matrix split over two ranks, neighbour pairs all-gathered:
    both ranks must reproduce the unsharded umi_group of the single pre-group."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tiles, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from tests import oracle_calls
    from tests.test_oracle_umi import umisim
    rng = np.random.default_rng(3)
    umis = []
    for _ in range(70):
        umis += umisim(rng, 9, 10)
    want = [c.tolist() for c in oracle_calls.umi_group(umis, 1, None, 1, [list(range(1, len(umis) + 1))])]
    assert results[0][1] == want and results[1][1] == want


def _worker_labels(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sarlacc_amd import pipeline
    from tests import oracle_calls
    from tests.test_oracle_umi import umisim
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        # every rank owns one pre-group (its own reads), as bench.py --gpus N does
        rng = np.random.default_rng(100 + rank)
        umis = []
        for _ in range(40):
            umis += umisim(rng, 5, 10)
        n = len(umis)
        clusters = oracle_calls.umi_group(umis, 1, None, 1, [list(range(1, n + 1))])
        coff = np.zeros(len(clusters) + 1, np.int64)
        np.cumsum([len(c) for c in clusters], out=coff[1:])
        cmem = np.concatenate(clusters).astype(np.int32)
        label, pos = pipeline.labels_from_clusters(coff, cmem, n)
        labs, poss, secs, nbytes = pipeline.all_gather_labels(label, pos, dist, None)
        q.put((rank, [c.tolist() for c in clusters], labs.tolist(), poss.tolist(), nbytes))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_label_all_gather():
    """The pipeline's one exchange step (bench.py --gpus N, BASELINE config 5): every rank clusters its own
    pre-group and all-gathers (label, position) per read; from the gathered vectors every rank can rebuild
    every rank's cluster list exactly."""
    import torch.multiprocessing as mp
    from sarlacc_amd import pipeline
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_labels, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][2] == results[1][2] and results[0][3] == results[1][3]   # both ranks hold the same gathered labels
    for r in range(2):
        coff, cmem = pipeline.clusters_from_labels(np.array(results[0][2][r]), np.array(results[0][3][r]))
        rebuilt = [cmem[coff[k]:coff[k + 1]].tolist() for k in range(coff.size - 1)]
        assert rebuilt == results[r][1]
        assert results[r][4] == 2 * 4 * len(results[0][2][r])   # bytes received from the one other rank


class _FailingCalls:
    """oracle_calls with one step failing on one rank: what a rank out of memory, or at the 2^32-link stop, looks like"""

    def __init__(self, base, step, on_rank, rank):
        self._base, self._step, self._fail = base, step, on_rank == rank

    def __getattr__(self, name):
        fn = getattr(self._base, name)
        if name == self._step and self._fail:
            def boom(*a, **k):
                raise RuntimeError("injected failure in " + name)
            return boom
        return fn


def _worker_tiles_failing(rank, world, port, q, step, on_rank):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sarlacc_amd import shard
    from tests import oracle_calls
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from tests.test_oracle_umi import umisim
        rng = np.random.default_rng(3)
        umis = []
        for _ in range(20):
            umis += umisim(rng, 9, 10)
        try:
            shard.sharded_umi_group_tiles(umis, 1, _FailingCalls(oracle_calls, step, on_rank, rank), dist)
            q.put((rank, "no error"))
        except shard.ShardError as e:
            q.put((rank, "ShardError: %s" % e))
        # the ranks left the call TOGETHER: the next collective still works, and so does a second, healthy call
        dist.barrier()
        out = shard.sharded_umi_group_tiles(umis, 1, oracle_calls, dist)
        q.put((rank, len(out)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("step,on_rank", [("umi_pairs_shard", 1), ("umi_group_from_pairs", 0)])
def test_a_failure_on_one_rank_is_raised_on_all_before_the_next_collective(step, on_rank):
    """bench.py's giant pre-group leg must never hang an N-GPU run: a local failure -- in the tile search of one rank, in the
    replicated clustering of another -- is agreed between the ranks (shard.agree) and raised on every one of them as
    ShardError; nobody is left waiting in the all-gather, and the process group stays usable."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tiles_failing, args=(r, 2, port, q, step, on_rank)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    msgs = {r: m for r, m in got if isinstance(m, str)}
    assert set(msgs) == {0, 1} and all(m.startswith("ShardError") for m in msgs.values()), msgs
    assert "injected failure in " + step in msgs[on_rank]            # the failing rank names its error,
    assert "injected failure" not in msgs[1 - on_rank]               # the other one only learns that somebody failed
    counts = [m for r, m in got if isinstance(m, int)]
    assert len(counts) == 2 and counts[0] == counts[1] > 0
