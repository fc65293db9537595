"""BASELINE configs[4] rehearsed on ONE GPU: umi_group on the UMIs of N x 10^6 reads as ONE pre-group (umiGroup without
`groups`, /root/reference/R/umiGroup.R:12-14, src/umi_group.cpp:35) -- what every rank's replicated clustering, and rank 0's
identity check, run at N = 8.  For every size: seconds, links, clustering rounds, clusters, device memory held afterwards,
and the size-independent properties (partition of the reads, second call identical); with --tiles the same set goes through
shard.sharded_umi_group_tiles with a world-size-1 process group (the pair exchange as it runs over RCCL, device tensors).
    python tools/perf_giant_group.py [sizes, e.g. 1e6,4e6,8e6,1e7] [threshold] [--tiles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sarlacc_amd
from sarlacc_amd import _lib, calls
from perf_umi import make_umis

args = [a for a in sys.argv[1:] if not a.startswith("--")]
sizes = [int(float(x)) for x in (args[0] if args else "1e6,2e6,4e6,8e6,1e7").split(",")]
thr = int(args[1]) if len(args) > 1 else 1
tiles = "--tiles" in sys.argv
dist = None
if tiles:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from sarlacc_amd import shard

for n in sizes:
    t0 = time.perf_counter()
    ss = make_umis(n // 10, 10, 1001)
    goff = np.array([0, len(ss)], np.int64)
    gflat = np.arange(1, len(ss) + 1, dtype=np.int32)
    print("n=%d UMIs generated in %.1f s" % (len(ss), time.perf_counter() - t0), flush=True)
    prev = None
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        coff, cmem = calls.umi_group_flat(ss, thr, None, thr, goff, gflat)
        dt = time.perf_counter() - t0
        free, total = torch.cuda.mem_get_info()
        sz = np.diff(coff)
        part = bool(cmem.size == len(ss) and np.array_equal(np.sort(cmem), gflat))
        same = None if prev is None else bool(np.array_equal(prev[0], coff) and np.array_equal(prev[1], cmem))
        prev = (coff, cmem)
        print("thr=%d n=%d umi_group %.3f s | pair kernels %.1f ms, search %.3f s in %d attempt(s) | links %.4g (%.1f per UMI) | adjacency %.3f s | clustering %.3f s in %d rounds "
              "(candidate sets %d, every list %d) | clusters %d (solo %d, largest %d, mean of the rest %.2f) | device memory held %.2f GB | partition %s | identical to first call %s" % (
                  thr, len(ss), dt, _lib.stage_ms("umi_pairs"), _lib.stage_count("umi_pair_search_s"), int(_lib.stage_count("umi_pair_attempts")),
                  _lib.stage_count("umi_links"), _lib.stage_count("umi_links") / len(ss), _lib.stage_count("umi_adjacency_s"),
                  _lib.stage_count("umi_cluster_s"), int(_lib.stage_count("umi_cluster_rounds")), int(_lib.stage_count("umi_cluster_candidate_rounds")),
                  int(_lib.stage_count("umi_cluster_full_rounds")), sz.size, int((sz == 1).sum()), int(sz.max()), float(sz[sz > 1].mean()) if (sz > 1).any() else 0.0,
                  (total - free) / 1e9, part, same), flush=True)
    bins = [2, 3, 5, 9, 13, 17, 25, 33, 49, 65, 10**9]
    big = sz[sz >= 2].astype(np.float64)
    cube = big ** 3
    print("      clusters of two and more by size (count, share of the reads in them, share of sum n^3 = the merge stage's work): " + "; ".join(
        "%d-%s: %d, %.3f, %.3f" % (lo, ("%d" % (hi - 1)) if hi < 10**9 else "", int(((big >= lo) & (big < hi)).sum()), big[(big >= lo) & (big < hi)].sum() / big.sum(),
                                   cube[(big >= lo) & (big < hi)].sum() / cube.sum()) for lo, hi in zip(bins[:-1], bins[1:])), flush=True)
    if tiles:
        st = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c2, m2 = shard.sharded_umi_group_tiles(ss, thr, calls, dist, torch.device("cuda:0"), flat=True, stats=st)
        dt = time.perf_counter() - t0
        free, total = torch.cuda.mem_get_info()
        print("thr=%d n=%d sharded_umi_group_tiles (RCCL, one rank) %.3f s: search %.3f, exchange %.3f, clustering %.3f | pairs %d (%.2f GB) | device memory held %.2f GB | identical to umi_group %s" % (
            thr, len(ss), dt, st["search_s"], st["exchange_s"], st["clustering_s"], st["pairs_all"], 8e-9 * st["pairs_all"], (total - free) / 1e9,
            bool(np.array_equal(c2, coff) and np.array_equal(m2, cmem))), flush=True)
    _lib.lib().sarlacc_release_workspace()
if tiles:
    dist.destroy_process_group()
