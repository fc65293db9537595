#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1]): adaptorAlign at the `.Call` level --
10^6 synthetic 2-kb Nanopore-like reads (mockReads recipe) against the 30-bp
adaptor (9 fixed + 12 N + 9 fixed), quality-aware local DP with traceback and one
section (the UMI), go=5, ge=1.  A "step" is one pass of adaptor_align over the
whole read batch, which is generated once and stays resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--read-len L]

For N > 1 launch with torch.distributed.run (one rank per GPU); reads shard
across ranks with no data-path collective (weak scaling: --reads is per GPU).
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ADAPTOR1 = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"  # 30 bp, UMI = the N run
ADAPTOR2 = "CACACTGAGCAGCGACTAGACA"              # 22 bp
UMI_SECTION = ([9], [21])                        # 0-based start, 1-based end
GAP_OPEN, GAP_EXT = 5.0, 1.0
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(sample_seq, sample_qual):
    """Oracle (bit-exact CPU port of the reference's loop) on a bounded sample, 1 core."""
    from oracle import oracle as O
    O.build()
    enc = O.phred_encoding()
    t0 = time.perf_counter()
    O.adaptor_align(sample_seq, sample_qual, enc, GAP_OPEN, GAP_EXT, ADAPTOR1, *UMI_SECTION)
    dt = time.perf_counter() - t0
    cells = sum(len(s) for s in sample_seq) * len(ADAPTOR1)
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
            "sample": "%d reads of the same batch (%.1f s of oracle adaptor_align, 1 thread)" % (len(sample_seq), dt)}


def pmc_traffic(n_reads):
    """HBM bytes per launch of the DP kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_1M_*.json; FETCH_SIZE and WRITE_SIZE are collected in separate passes and
    cannot be collected from inside this process).  Returned only for the configuration the
    passes were run on; FETCH_SIZE is taken at face value (the guide's possible 2x under-count
    for wide reads is noted in the file)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_1M_v*.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files or n_reads != 1_000_000:
        return None
    with open(files[-1]) as fh:
        d = json.load(fh)
    kb = sum(d["FETCH_SIZE_KB_per_launch"]) / len(d["FETCH_SIZE_KB_per_launch"]) + \
        sum(d["WRITE_SIZE_KB_per_launch"]) / len(d["WRITE_SIZE_KB_per_launch"])
    return kb * 1024.0, d.get("derived")


def pipeline_sample(groups=10000, read_len=2000, copies=10):
    """Second half of the headline metric on a bounded sample: reads/min through
    umi_group -> quick_msa -> create_consensus_quality_loop (host-pointer C ABI, PCIe included)."""
    import numpy as np
    import sarlacc_amd
    from sarlacc_amd import calls
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from perf_pipeline import NUC, noisy_copies
    rng = np.random.default_rng(1000)
    umis, _ = noisy_copies(NUC[rng.integers(0, 4, (groups, 12))], copies, rng)
    reads, quals = noisy_copies(NUC[rng.integers(0, 4, (groups, read_len))], copies, rng)
    n = len(reads)
    enc = sarlacc_amd.phred_encoding()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        coff, cmem = calls.umi_group_flat(umis, 1, None, 1, np.array([0, n], np.int64), np.arange(1, n + 1, dtype=np.int32))
        goff, gflat = calls.csr_select(coff, cmem, np.diff(coff) >= 2)
        cons, _ = calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
        dt = time.perf_counter() - t0   # everything between the UMI strings and the consensus strings
        best = dt if best is None else min(best, dt)
    return {"reads_per_min": n / best * 60.0, "reads": n, "consensus_reads": len(cons),
            "workload": "%d molecules x %d reads x %d bp, 12-bp UMIs: umi_group(threshold 1) -> msa_consensus (quick_msa bandwidth 100 + quality consensus, rows stay in HBM), "
                        "host-pointer C ABI incl. PCIe and the host glue between the calls" % (groups, copies, read_len)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=2000)
    ap.add_argument("--cpu-sample", type=int, default=25000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    backend = os.environ.get("SARLACC_DIST_BACKEND", "nccl")  # gloo only to exercise this path on one GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    red_device = device if backend == "nccl" else torch.device("cpu")

    import sarlacc_amd
    from sarlacc_amd import device as sdev
    from sarlacc_amd import devsynth
    sarlacc_amd.set_device(dev_index)
    enc = sarlacc_amd.phred_encoding()

    n = args.reads
    seq, qual, off, max_len = devsynth.make_reads(n, args.read_len, ADAPTOR1, ADAPTOR2,
                                                  seed=1000 + rank, device=device)
    total_bases = int(off[-1].item())
    R = len(ADAPTOR1)
    cells = total_bases * R
    scores = torch.empty(n, dtype=torch.float64, device=device)
    starts = torch.empty(n, dtype=torch.int32, device=device)
    ends = torch.empty(n, dtype=torch.int32, device=device)
    sso = torch.empty(n, dtype=torch.int32, device=device)
    swo = torch.empty(n, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    # resident format: 2-bit packed bases + exception bit-mask + 1 B quality per base
    packed = torch.empty(total_bases // 4 + 2, dtype=torch.uint8, device=device)
    nmask = torch.empty(total_bases // 8 + 1, dtype=torch.uint8, device=device)
    sdev.dev_pack_reads(seq, total_bases, packed, nmask, stream)
    torch.cuda.synchronize()

    def step():
        sdev.dev_align(packed, qual, off, n, max_len, enc, GAP_OPEN, GAP_EXT, ADAPTOR1, True,
                       UMI_SECTION[0], UMI_SECTION[1], scores, starts, ends, sso, swo, stream, d_nmask=nmask)
        return sarlacc_amd.last_kernel_ms()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        kern_ms.append(step())
    fence()
    elapsed = time.perf_counter() - t0

    el = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
    tot_cells = torch.tensor([float(cells)], dtype=torch.float64, device=red_device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_cells, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    all_cells = float(tot_cells.item())

    if rank == 0:
        gcups = all_cells * args.steps / elapsed / 1e9
        # algorithmic bytes per alignment: 2-bit packed bases + 1 B
        # quality per base in, 8 B score + 8 B start/end + 8 B per section out
        alg_bytes = total_bases / 4.0 + total_bases + n * 24.0
        k_ms = sum(kern_ms) / len(kern_ms)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "GCUPS (quality-aware pairwise DP cell updates/s), adaptor_align .Call level",
            "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "adaptorAlign: %d x %d bp mockReads-like reads per GPU vs 30 bp adaptor, "
                                   "local quality DP + traceback + 1 section, go=5 ge=1" % (n, args.read_len),
                       "reads_per_gpu": n, "read_len": args.read_len, "adaptor_len": R,
                       "resident_format": "2-bit packed bases + 1-bit non-ACGT mask + 1 B quality per base",
                       "sharding": "reads split across ranks, no collective"},
            "reads_per_s": n * world * args.steps / elapsed,
            "kernel_ms": k_ms,
            "kernel_gcups": cells / (k_ms * 1e-3) / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": (pmc_traffic(n) or (None, None))[0],
                         "algorithmic_bytes": alg_bytes,
                         "valu": (pmc_traffic(n) or (None, None))[1],
                         "note": "the DP kernel is VALU-issue bound (valu.valu_busy_frac from the PMC pass); compulsory traffic is 0.042 B/cell"},
        }
        if not args.no_pipeline:
            out["pipeline"] = pipeline_sample()
        if not args.no_cpu:
            m = min(args.cpu_sample, n)
            s_s, s_q = devsynth.to_host_strings(seq, qual, off, m)
            out["cpu_baseline"] = cpu_baseline(s_s, s_q)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
