#!/usr/bin/env python3
"""BASELINE config 5 in miniature: the whole path on N GPUs of one node, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        tools/run_pipeline.py --molecules 100000 [--copies 10] [--read-len 2000]

Stages:
  1. adaptor_align on this rank's contiguous read range (.parallelize, R/adaptorAlign.R:126-134)
     -- reads resident in HBM, no exchange;
  2. umiGroup of ONE pre-group holding every read: row tiles of the all-pairs search per rank,
     all-gather of the neighbour pairs (RCCL over xGMI; gloo when SARLACC_DIST_BACKEND=gloo),
     exact clustering replicated on every rank -> identical cluster lists everywhere;
  3. clusters of >= 2 reads bin-packed over the ranks by their bases; each rank runs
     multiReadAlign + consensusReadSeq (sarlacc_msa_consensus) on its clusters -- no exchange;
  4. all-reduce of the counters, max-over-ranks timing, rank 0 prints one JSON line.

Every rank builds the same synthetic data set from the seed (on a real run every rank reads
the same FASTQ), so regrouping reads by cluster needs no read exchange: the label all-gather of
stage 2 is the only data-path collective, as in the reference's process-level chunking.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ADAPTOR1 = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=10000)
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--read-len", type=int, default=2000)
    ap.add_argument("--threshold", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--repeat", type=int, default=2, help="passes over the stages; the last one is reported, the first as first_pass_s")
    ap.add_argument("--check", action="store_true", help="rank 0 recomputes everything unsharded and compares")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("SARLACC_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("run_pipeline.py needs a HIP device")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    red_device = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    gather_device = red_device if backend == "nccl" else None

    import sarlacc_amd
    from sarlacc_amd import calls, shard
    from sarlacc_amd.generics import Reads
    from sarlacc_amd.resident import DeviceReads
    from perf_pipeline import NUC, noisy_copies
    sarlacc_amd.set_device(dev_index)
    enc = sarlacc_amd.phred_encoding()

    rng = np.random.default_rng(args.seed)   # same stream on every rank
    G = args.molecules
    umis, _ = noisy_copies(NUC[rng.integers(0, 4, (G, 12))], args.copies, rng)
    reads, quals = noisy_copies(NUC[rng.integers(0, 4, (G, args.read_len))], args.copies, rng)
    n = len(reads)
    D = dist if world > 1 else None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    calls.umi_group(["ACGT", "ACGA"], 1, None, 1, [[1, 2]])   # context and workspace warm-up, untimed
    def stages():
        fence()
        t0 = time.perf_counter()
        # 1. adaptor DP on the local read range
        lo, hi = shard.shard_range(n, rank, world)
        local = DeviceReads.upload(Reads(reads.slice(lo, hi), quals.slice(lo, hi), encoding=enc))
        scores = local.align_map(ADAPTOR1, 5, 1, [9], [21])[0]
        fence()
        t1 = time.perf_counter()
        # 2. one giant pre-group: tile shards + all-gather of neighbour pairs + replicated clustering
        coff, cmem = shard.sharded_umi_group_tiles(umis, args.threshold, calls, D, gather_device, flat=True)
        fence()
        t2 = time.perf_counter()
        # 3. clusters of >= 2 reads, dealt to the ranks by their bases; MSA + consensus on the owned ones
        sizes = np.diff(coff)
        big = np.flatnonzero(sizes >= 2)
        w = reads.widths()
        bases = np.add.reduceat(w[cmem.astype(np.int64) - 1], coff[:-1][sizes > 0]) if cmem.size else np.zeros(0)
        cost = np.zeros(sizes.size)
        cost[sizes > 0] = bases
        owner = shard.assign_groups_snake(cost[big], world)
        keep = np.zeros(sizes.size, bool)
        keep[big[owner == rank]] = True
        goff, gflat = calls.csr_select(coff, cmem, keep)
        cons, phred = calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
        fence()
        t3 = time.perf_counter()

        return (t1 - t0, t2 - t1, t3 - t2, t3 - t0), scores, coff, cmem, sizes, big, gflat, cons

    runs = [stages() for _ in range(max(1, args.repeat))]
    (d1, d2, d3, dall), scores, coff, cmem, sizes, big, gflat, cons = runs[-1]
    cold = runs[0][0]
    stats = torch.tensor([float(len(cons)), float(cons.total), float(gflat.size), float(scores.sum())],
                         dtype=torch.float64, device=red_device)
    times = torch.tensor([d1, d2, d3, dall, cold[0], cold[1], cold[2], cold[3]], dtype=torch.float64, device=red_device)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    if rank == 0:
        st, tm = stats.cpu().tolist(), times.cpu().tolist()
        out = {"metric": "reads/min through adaptor_align -> umiGroup -> multiReadAlign -> consensusReadSeq",
               "value": n / tm[3] * 60.0, "unit": "reads/min", "n_gpus": world, "reads": n, "clusters": int(sizes.size),
               "consensus_reads": int(st[0]), "consensus_bases": int(st[1]), "reads_in_clusters": int(st[2]),
               "score_checksum": st[3], "stage_s": {"adaptor_align": tm[0], "umi_group": tm[1], "msa_consensus": tm[2]},
               "first_pass_s": {"adaptor_align": tm[4], "umi_group": tm[5], "msa_consensus": tm[6], "total": tm[7]},
               "passes": max(1, args.repeat), "backend": backend, "scaling": "strong"}
        if args.check:
            ref_scores = DeviceReads.upload(Reads(reads, quals, encoding=enc)).align_map(ADAPTOR1, 5, 1, [9], [21])[0]
            ref_off, ref_mem = calls.umi_group_flat(umis, args.threshold, None, args.threshold, np.array([0, n], np.int64),
                                                    np.arange(1, n + 1, dtype=np.int32))
            same = np.array_equal(ref_off, coff) and np.array_equal(ref_mem, cmem)
            allbig = np.zeros(sizes.size, bool)
            allbig[big] = True
            goff_all, gflat_all = calls.csr_select(coff, cmem, allbig)
            ref_cons, _ = calls.msa_consensus_flat(goff_all, gflat_all, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
            out["check"] = {"clusters_identical": bool(same), "consensus_reads": len(ref_cons), "consensus_bases": ref_cons.total,
                            "score_checksum": float(ref_scores.sum())}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
