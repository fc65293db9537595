"""Sharding of the hot path over the GPUs of one node (one process per GPU).

The reference's only parallelism is BiocParallel chunking of reads / groups across worker
processes (/root/reference/R/adaptorAlign.R:126-134, R/multiReadAlign.R:29-31); every unit
(read, pre-group, group) is independent, so the path shards with no data-path collective.
The single real exchange step of the pipeline is making the UMI cluster assignment known
everywhere before reads are regrouped for the MSA: an all-gather of per-read cluster labels
(torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU tests) -- 2 x int32 per read.
"""
import numpy as np


def contiguous_bounds(n, parts):
    """Chunk boundaries of .parallelize (R/adaptorAlign.R:126-134): `parts` contiguous
    chunks whose starts are findInterval(seq_len(n), seq(1, n, length.out=parts+1)[-last])."""
    if parts <= 1 or n == 0:
        return np.array([0, n], dtype=np.int64)
    b = np.linspace(1, n, parts + 1)[:-1]
    ids = np.searchsorted(b, np.arange(1, n + 1), side="right")  # 1..parts
    starts = np.searchsorted(ids, np.arange(1, parts + 1), side="left")
    return np.concatenate([starts, [n]]).astype(np.int64)


def shard_range(n, rank, world):
    """[lo, hi) of the reads owned by `rank`."""
    b = contiguous_bounds(n, world)
    return int(b[rank]), int(b[rank + 1])


def assign_groups(sizes, world, power=2.0):
    """Longest-processing-time bin packing of pre-groups by cost size**power (the all-pairs
    neighbour search is quadratic in the pre-group size).  Deterministic; returns owner[g]."""
    sizes = np.asarray(sizes, dtype=np.float64)
    order = np.lexsort((np.arange(sizes.size), -sizes))  # big first, ties by index
    load = np.zeros(world)
    owner = np.zeros(sizes.size, dtype=np.int64)
    for g in order:
        r = int(np.argmin(load))
        owner[g] = r
        load[r] += sizes[g] ** power
    return owner


def assign_groups_snake(cost, world):
    """Vectorised near-balanced assignment for many groups of similar cost: groups sorted by cost
    (ties by index) are dealt to the ranks in snake order 0..w-1, w-1..0, ...  Deterministic."""
    cost = np.asarray(cost, dtype=np.float64)
    order = np.lexsort((np.arange(cost.size), -cost))
    pos = np.arange(cost.size)
    lap, k = pos // world, pos % world
    owner = np.empty(cost.size, dtype=np.int64)
    owner[order] = np.where(lap % 2 == 0, k, world - 1 - k)
    return owner


def _all_gather(vec, dist, device=None):
    """all_gather of a 1-D int32 numpy array of identical length on every rank."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(vec))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy() for o in out]


def sharded_umi_group(umi1, threshold1, umi2, threshold2, pregroups, calls, dist=None, device=None):
    """umi_group with the pre-groups spread over the ranks of `dist` (torch.distributed,
    already initialised; None = single process).  Every rank returns the complete list of
    clusters, identical -- order included -- to the unsharded call.

    Exchange: one all-gather of two int32 vectors of length n_reads (cluster label and
    position inside the cluster); cluster counts per pre-group ride in the same message."""
    ngroups = len(pregroups)
    if dist is None or dist.get_world_size() == 1:
        return calls.umi_group(umi1, threshold1, umi2, threshold2, pregroups)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = len(umi1)
    owner = assign_groups([len(g) for g in pregroups], world)
    mine = [g for g in range(ngroups) if owner[g] == rank]
    local = calls.umi_group(umi1, threshold1, umi2, threshold2, [pregroups[g] for g in mine]) if mine else []

    # which pre-group does each local cluster belong to: clusters come back group by group
    label = np.full(n, -1, dtype=np.int32)      # index of the cluster inside its pre-group
    pos = np.full(n, -1, dtype=np.int32)        # position of the read inside its cluster
    counts = np.zeros(ngroups, dtype=np.int32)  # clusters per pre-group
    group_of_read = {}
    for g in mine:
        for r in np.asarray(pregroups[g]).tolist():
            group_of_read[r] = g
    for clu in local:
        clu = np.asarray(clu)
        g = group_of_read[int(clu[0])]
        label[clu - 1] = counts[g]
        pos[clu - 1] = np.arange(clu.size, dtype=np.int32)
        counts[g] += 1

    msg = np.concatenate([label, pos, counts]).astype(np.int32)
    gathered = _all_gather(msg, dist, device)
    label_all = np.full(n, -1, dtype=np.int64)
    pos_all = np.full(n, -1, dtype=np.int64)
    counts_all = np.zeros(ngroups, dtype=np.int64)
    for m in gathered:
        lab, po, cnt = m[:n], m[n:2 * n], m[2 * n:]
        have = lab >= 0
        label_all[have] = lab[have]
        pos_all[have] = po[have]
        counts_all += cnt
    base = np.concatenate([[0], np.cumsum(counts_all)])
    # rebuild the flattened cluster list in the reference's order (pre-group by pre-group)
    out = []
    for g in range(ngroups):
        members = np.asarray(pregroups[g], dtype=np.int64)
        k = int(counts_all[g])
        lab = label_all[members - 1]
        po = pos_all[members - 1]
        order = np.lexsort((po, lab))
        sizes = np.bincount(lab, minlength=k)
        cuts = np.concatenate([[0], np.cumsum(sizes)])
        srt = members[order]
        for c in range(k):
            out.append(srt[cuts[c]:cuts[c + 1]].astype(np.int32))
    assert len(out) == int(base[-1])
    return out


def sharded_over_reads(fn, n, dist=None):
    """Run fn(lo, hi) on this rank's contiguous read range; returns (lo, hi, result)."""
    if dist is None or dist.get_world_size() == 1:
        return 0, n, fn(0, n)
    lo, hi = shard_range(n, dist.get_rank(), dist.get_world_size())
    return lo, hi, fn(lo, hi)


def sharded_umi_group_tiles(umi, threshold, calls, dist=None, device=None, flat=False, stats=None):
    """umi_group of ONE giant pre-group with the row tiles of the all-pairs matrix spread over the
    ranks.  Every rank holds all UMIs (they are 12 bytes each), searches its
    share of the tiles, all-gathers the neighbour pairs (counts first, then the padded lists) and
    runs the clustering on the concatenation -- replicated, deterministic, identical to the
    single-GPU result.  `stats` (a dict, optional) receives the seconds of the search, of the exchange and of the
    clustering, the pairs found here and overall, and the bytes this rank received."""
    import time
    if dist is None:   # (a process group of one rank goes through the collectives below: the RCCL path on a one-GPU box)
        return calls.umi_group_from_pairs(umi, threshold, calls.umi_pairs_shard(umi, threshold, 0, 1), **({"flat": True} if flat else {}))
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    t0 = time.perf_counter()
    mine = calls.umi_pairs_shard(umi, threshold, rank, world).astype(np.int64)  # values < 2^63: safe as int64
    t1 = time.perf_counter()
    counts = _all_gather(np.array([mine.size], dtype=np.int32), dist, device)
    counts = [int(c[0]) for c in counts]
    width = max(max(counts), 1)
    padded = np.zeros(width, dtype=np.int64)
    padded[:mine.size] = mine
    t = torch.from_numpy(padded)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    allpairs = np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)]).astype(np.uint64)
    t2 = time.perf_counter()
    out = calls.umi_group_from_pairs(umi, threshold, allpairs, **({"flat": True} if flat else {}))
    if stats is not None:
        stats.update({"search_s": t1 - t0, "exchange_s": t2 - t1, "clustering_s": time.perf_counter() - t2, "pairs_here": int(mine.size),
                      "pairs_all": int(allpairs.size), "bytes_received": int(8 * width * (world - 1) + 4 * (world - 1))})
    return out
