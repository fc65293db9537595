"""Where umi_group's time goes at a given threshold: neighbour search (pair kernel + sorts), clustering rounds, list building.
    python tools/perf_umi_stages.py [n] [thresholds, e.g. 2,3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import _lib, calls
from perf_umi import make_umis

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
thrs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]
ss = make_umis(n // 10, 10, 1001)
goff = np.array([0, len(ss)], np.int64)
gflat = np.arange(1, len(ss) + 1, dtype=np.int32)
for thr in thrs:
    for rep in range(2):
        t0 = time.perf_counter()
        coff, cmem = calls.umi_group_flat(ss, thr, None, thr, goff, gflat)
        dt = time.perf_counter() - t0
        print("thr=%d n=%d flat call %.3f s | pair kernel %.1f ms | adjacency %.3f s (links %.3g) | clustering %.3f s in %d rounds | clusters %d | encode+sort %.3f s, search+key sort %.3f s (pair search %.3f s in %d attempt(s)) | split-key search %d (length classes %d, work items %.3g, special strings %d) | rounds on candidate sets %d, over every list %d" % (
            thr, len(ss), dt, _lib.stage_ms("umi_pairs"), _lib.stage_count("umi_adjacency_s"), _lib.stage_count("umi_links"),
            _lib.stage_count("umi_cluster_s"), int(_lib.stage_count("umi_cluster_rounds")), coff.size - 1, _lib.stage_count("umi_encode_sort_s"),
            _lib.stage_count("umi_search_and_key_sort_s"), _lib.stage_count("umi_pair_search_s"), int(_lib.stage_count("umi_pair_attempts")),
            int(_lib.stage_count("umi_split_search")), int(_lib.stage_count("umi_split_classes")), _lib.stage_count("umi_split_items"), int(_lib.stage_count("umi_split_special")),
            int(_lib.stage_count("umi_cluster_candidate_rounds")), int(_lib.stage_count("umi_cluster_full_rounds"))), flush=True)
