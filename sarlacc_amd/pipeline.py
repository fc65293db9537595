"""umiGroup -> multiReadAlign -> consensusReadSeq on a device-resident read batch: the host logic
between the native calls (vignette order, /root/reference/vignettes/correction.Rmd:317-345; the R
callers are R/umiGroup.R:8-22, R/multiReadAlign.R:16-47, R/consensusReadSeq.R:14-21), shared by
bench.py (BASELINE configs C3 + C4, C5) and tools/.

Sharding: pre-groups are independent (src/umi_group.cpp:35), so every rank
owns whole pre-groups -- here one pre-group per rank, its own reads -- clusters them locally, and
the per-read cluster labels (cluster index inside the pre-group, position inside the cluster:
2 x int32 per read) are all-gathered over RCCL so that every rank knows the complete assignment,
which is what umiGroup returns.  Groups for the MSA stay on the rank that holds their reads.
"""
import time

import numpy as np

from . import _lib, calls, device


def labels_from_clusters(coff, cmem, n):
    """CSR clusters of 1-based read ids -> (label int32[n], pos int32[n]); -1 for reads in no cluster."""
    sizes = np.diff(coff)
    label = np.full(n, -1, np.int32)
    pos = np.full(n, -1, np.int32)
    idx = cmem.astype(np.int64) - 1
    label[idx] = np.repeat(np.arange(sizes.size, dtype=np.int32), sizes)
    pos[idx] = (np.arange(cmem.size, dtype=np.int64) - np.repeat(coff[:-1], sizes)).astype(np.int32)
    return label, pos


def clusters_from_labels(label, pos):
    """Inverse of labels_from_clusters for one pre-group: CSR (coff, cmem) in cluster order."""
    have = np.flatnonzero(label >= 0)
    lab = label[have].astype(np.int64)
    k = int(lab.max()) + 1 if have.size else 0
    sizes = np.bincount(lab, minlength=k)
    coff = np.zeros(k + 1, np.int64)
    np.cumsum(sizes, out=coff[1:])
    cmem = np.zeros(have.size, np.int32)
    cmem[coff[lab] + pos[have]] = (have + 1).astype(np.int32)   # (label, position) names the slot: a scatter, no sort
    return coff, cmem


def all_gather_labels(label, pos, dist, device_t=None):
    """All-gather of the per-read labels of every rank's pre-group (equal read counts per rank).
    With a device given the exchange runs on device tensors (RCCL); returns
    (labels int32[world, n], positions int32[world, n], seconds, bytes received per rank)."""
    import torch
    msg = torch.from_numpy(np.stack([label, pos]))
    if device_t is not None:
        msg = msg.to(device_t)
    world = dist.get_world_size()
    parts = [torch.empty_like(msg) for _ in range(world)]   # (list form: works on RCCL and on gloo)
    if device_t is not None:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.all_gather(parts, msg)
    if device_t is not None:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = np.stack([p.cpu().numpy() for p in parts])
    return res[:, 0, :], res[:, 1, :], dt, int(msg.numel() * msg.element_size() * (world - 1))


def run_resident(umis, d_seq, d_qual, off_host, encoding, threshold=1, bandwidth=100, min_cov=0.6, dist=None,
                 gather_device=None, min_cluster=2):
    """One pass over this rank's pre-group.  umis: StringSet (host) of the n UMIs; d_seq/d_qual: device
    tensors of the concatenated reads / qualities; off_host: numpy int64[n+1].
    Returns a dict with the consensus StringSets, the cluster CSR, the gathered labels and per-stage
    wall seconds and kernel milliseconds."""
    n = len(umis)
    t0 = time.perf_counter()
    coff, cmem = calls.umi_group_flat(umis, threshold, None, threshold, np.array([0, n], np.int64),
                                      np.arange(1, n + 1, dtype=np.int32))
    t1 = time.perf_counter()
    umi_ms = _lib.stage_ms("umi_pairs")
    umi_host = {k: _lib.stage_count("umi_%s_s" % k) for k in ("tables_in", "encode_sort", "search_and_key_sort", "adjacency", "cluster", "clusters_out")}
    gathered = None
    gather_s, gather_bytes = 0.0, 0
    clusters_all_ranks = int(coff.size - 1)
    if dist is not None and dist.get_world_size() > 1:
        label, pos = labels_from_clusters(coff, cmem, n)   # (the wire format of the exchange: nothing to build for a single rank)
        labs, poss, gather_s, gather_bytes = all_gather_labels(label, pos, dist, gather_device)
        gathered = (labs, poss)
        # the exchange's result is what the rest of the pass runs on: this rank's clusters are rebuilt from its
        # row of the gathered labels (and must be the ones it sent), the other rows give the global assignment
        coff_g, cmem_g = clusters_from_labels(labs[dist.get_rank()], poss[dist.get_rank()])
        if not (np.array_equal(coff_g, coff) and np.array_equal(cmem_g, cmem)):
            raise RuntimeError("all-gathered cluster labels do not rebuild this rank's clusters")
        coff, cmem = coff_g, cmem_g
        clusters_all_ranks = int((labs.max(axis=1) + 1).sum())
    t2 = time.perf_counter()
    goff, gflat = calls.csr_select(coff, cmem, np.diff(coff) >= min_cluster)
    t3 = time.perf_counter()
    cons, phred = device.dev_msa_consensus(goff, gflat, d_seq, d_qual, off_host, 0, -1, -5, -1, bandwidth, min_cov,
                                           encoding=encoding)
    t4 = time.perf_counter()
    return {
        "cons": cons, "phred": phred, "coff": coff, "cmem": cmem, "goff": goff, "gflat": gflat, "gathered": gathered,
        "umi_group_host_s": umi_host,
        "stage_s": {"umi_group": t1 - t0, "label_exchange": t2 - t1, "host_glue": t3 - t2, "msa_consensus": t4 - t3,
                    "total": t4 - t0},
        "all_gather_s": gather_s, "all_gather_bytes": gather_bytes, "clusters_all_ranks": clusters_all_ranks,
        "kernel_ms": {"umi_pairs": umi_ms, "msa_pairwise": _lib.stage_ms("msa_pairwise"),
                      "msa_merge": _lib.stage_ms("msa_merge"), "consensus": _lib.stage_ms("consensus")},
        "counts": {"msa_pairs": _lib.stage_count("msa_pairs"), "msa_cells": _lib.stage_count("msa_cells"),
                   "consensus_cells": _lib.stage_count("consensus_cells"), "msa_v1_fallback": _lib.stage_count("msa_v1_fallback"),
                   **{k: _lib.stage_count(k) for k in ("msa2_rows", "msa2_rows_capped", "msa2_rows_filtered", "msa2_entries_filtered",
                                                       "msa2_entries_kept", "msa2_joins", "msa2_joins_chain_in_hbm", "msa2_gathers",
                                                       "msa2_groups_second_pass", "msa2_batches", "msa2_cycles_rows", "msa2_cycles_chain", "msa2_cycles_walk",
                                                       "msa2_cycles_renumber", "msa_pairs_bitvector", "msa_bitvector_tile_bytes", "msa_bitvector_redone")}},
    }
