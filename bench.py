#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

Headline workload (BASELINE.json configs[1]): adaptorAlign at the `.Call` level --
10^6 synthetic 2-kb Nanopore-like reads (mockReads recipe) against the 30-bp
adaptor (9 fixed + 12 N + 9 fixed), quality-aware local DP with traceback and one
section (the UMI), go=5, ge=1.  A "step" is one pass of adaptor_align over the
whole read batch, which is generated once and stays resident in HBM; `value` = GCUPS.

Second half of the headline metric (configs[2] + configs[3]), reported under "pipeline":
10^5 molecules x 10 reads x 2 kb generated in HBM, umiGroup (threshold 1, one pre-group
per GPU) -> multiReadAlign (bandwidth 100) -> consensusReadSeq (quality vote), reads/min
from the UMI strings to the consensus strings, with per-stage seconds, per-kernel
milliseconds (HIP events on the launch stream) and a roofline entry for the two dominant
kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--read-len L]

For N > 1: one rank per GPU under torch.distributed.run -- either started by the caller (RANK / WORLD_SIZE in
the environment) or, when `python bench.py --gpus N` is run bare, by bench.py itself as a child process; --gpus
larger than the visible device count is an error, never a silent one-GPU run.  Reads shard across ranks
(weak scaling: --reads and --molecules are per GPU); the DP has no data-path collective,
the pipeline all-gathers the UMI cluster labels over RCCL (configs[4] in miniature).
Rank 0 prints one JSON line.
"""
import argparse
import glob
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ADAPTOR1 = "ACGATCAGC" + "N" * 12 + "GTCAGTCAG"  # 30 bp, UMI = the N run
ADAPTOR2 = "CACACTGAGCAGCGACTAGACA"              # 22 bp
UMI_SECTION = ([9], [21])                        # 0-based start, 1-based end
GAP_OPEN, GAP_EXT = 5.0, 1.0
# MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); 256 CUs x 4 SIMD-32 at 2.4 GHz: a wave64 32-bit VALU
# instruction issues in 2 cycles (32 lanes/clk/SIMD), an fp64 one in 4 (16 lanes/clk/SIMD)
HBM_PEAK_GBS = 8000.0
VALU32_PEAK_TLOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 78.6 T lane-ops/s, 32-bit
VALU64_PEAK_TLOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # 39.3 T lane-ops/s, fp64
# algorithmic work per DP cell: quality DP = 10 fp64 add/sub/max/compare
# (Appendix A of SURVEY.md: H, LJ, V, UJ updates, M, three compares); integer Gotoh of the MSA
# pairwise stage = 5 add + 4 max + 2 for the match score = 11 int32 ops (traceback bits excluded); with the reference's
# default scores the first gap character costs no more than a further one, the recurrence is the linear-gap one and
# the kernel runs it as such: 3 add + 2 min + 2 for the match score = 7
ALIGN_OPS_PER_CELL = 10
MSA_OPS_PER_CELL = 7


def source_sha(names):
    h = hashlib.sha256()
    for nm in names:
        with open(os.path.join(ROOT, "sarlacc_amd", "csrc", nm), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_record(kernel, sources):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*_pmc_<kernel>*.json,
    written by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes; counters cannot be
    collected from inside this process).  Returned only while the kernel's source files still hash to
    what was profiled, so stale counters never sit next to fresh timings."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s*.json" % kernel)),
                   key=lambda f: (["_final_" in os.path.basename(f)], [int(x) for x in re.findall(r"\d+", os.path.basename(f))]))
    for f in reversed(files):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("kernel_substring", kernel) != kernel:   # ("k_align*" also matches the summaries of k_align_wide)
            continue
        if d.get("source_sha") != source_sha(sources):
            continue
        fe, wr = d.get("FETCH_SIZE_KB_per_launch") or [], d.get("WRITE_SIZE_KB_per_launch") or []
        if not fe or not wr:
            continue
        # gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM) -- reads are doubled
        return {"traffic": d.get("traffic_bytes_per_launch", (2.0 * sum(fe) / len(fe) + sum(wr) / len(wr)) * 1024.0),
                "traffic_source": {"file": os.path.relpath(f, ROOT), "box": d.get("box"), "same_box_bench": d.get("same_box_bench"),
                                   "profiled_kernel_avg_ms": (d.get("kernel_stats") or {}).get("avg_ms"),
                                   "note": "counters and timings of the profiled run (its box, its own bench line, the profiler's kernel average): "
                                           "compare with this line's kernel_ms before reading traffic against it"},
                "workload": d.get("workload"), "derived": d.get("derived")}
    return {"traffic": None, "traffic_source": None}


def cpu_baseline(seq_strings, qual_strings, cores):
    """Oracle (bit-exact CPU port of the reference's loop, oracle/align.c) on a bounded sample of the
    same batch: one thread, and all host cores (threads over read chunks = the analogue of the
    reference's BiocParallel chunking, R/adaptorAlign.R:126-134; ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    enc = O.phred_encoding()
    R = len(ADAPTOR1)

    def run(lo, hi):
        O.adaptor_align(seq_strings[lo:hi], qual_strings[lo:hi], enc, GAP_OPEN, GAP_EXT, ADAPTOR1, *UMI_SECTION)
        return sum(len(s) for s in seq_strings[lo:hi]) * R

    n = len(seq_strings)
    n1 = max(1, min(n, n // max(cores, 1)))
    t0 = time.perf_counter()
    cells1 = run(0, n1)
    dt1 = time.perf_counter() - t0
    chunks = [(n * k // (4 * cores), n * (k + 1) // (4 * cores)) for k in range(4 * cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        cells = sum(ex.map(lambda c: run(*c), chunks))
    dt = time.perf_counter() - t0
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": "%d reads of the same batch, %d threads over read chunks (%.1f s of oracle adaptor_align)" % (n, cores, dt),
            "single_core": {"value": cells1 / dt1 / 1e9, "unit": "GCUPS", "cores": 1,
                            "sample": "%d reads, 1 thread (%.1f s)" % (n1, dt1)},
            "note": "the reference's C++ needs Rcpp/Biostrings headers absent from the image, hence a port"}


def cpu_baseline_pipeline(umi_strings, threshold, groups, read_strings, qual_strings, cores):
    """The oracle beside the pipeline half: umi_group on the first 2*10^4 and 10^5 UMIs of the batch as one
    pre-group (one thread: the reference's loop over one pre-group is serial, src/umi_group.cpp:35, and its clustering
    is O(#clusters x n), so the 10^6 figure is an explicit extrapolation), quick_msa under the default spec and
    create_consensus_quality_loop on a sample of the batch's clusters (all cores over groups = the analogue of
    R/multiReadAlign.R:29-31, and one core)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    enc = O.phred_encoding()
    out = {"kind": "port", "cores": cores}
    um = {}
    for n in (20000, 100000):
        if n > len(umi_strings):
            continue
        t0 = time.perf_counter()
        O.umi_group(umi_strings[:n], threshold, None, threshold, [list(range(1, n + 1))])
        um[str(n)] = time.perf_counter() - t0
    if um:
        out["umi_group_seconds"] = um
        out["umi_group_note"] = "one pre-group, 1 thread; super-linear in n (greedy clustering scans all live nodes per cluster)"
        if "20000" in um and "100000" in um and um["20000"] > 0:
            import math
            expo = math.log(um["100000"] / um["20000"]) / math.log(5.0)
            out["umi_group_extrapolated_1M_seconds"] = um["100000"] * 10.0 ** expo
            out["umi_group_scaling_exponent"] = expo

    def run(gs):
        aln = O.quick_msa(gs, read_strings, 0, -1, -5, -1, 100)
        q = [[qual_strings[i - 1] for i in g] for g in gs]
        O.create_consensus_quality_loop(aln, 0.6, q, enc)
        return sum(len(g) for g in gs)

    t0 = time.perf_counter()
    n1 = run(groups[:max(1, len(groups) // max(cores, 1))])
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        nall = sum(ex.map(lambda g: run([g]), groups))
    dt = time.perf_counter() - t0
    out["msa_consensus"] = {"value": nall / dt * 60.0, "unit": "reads/min", "cores": cores,
                            "sample": "%d clusters (%d reads) of the same batch, quick_msa spec v2 + quality consensus, %d threads over "
                                      "clusters (%.1f s)" % (len(groups), nall, cores, dt),
                            "single_core": {"value": n1 / dt1 * 60.0, "unit": "reads/min", "cores": 1,
                                            "sample": "%d reads (%.1f s)" % (n1, dt1)}}
    return out


def giant_group_leg(args, rank, world, dist, device, gather_device, backend, enc, mol, fence, reduce):
    """BASELINE configs[4] as it is worded, in the bench line for N > 1: the reads of ALL ranks as ONE pre-group (R/umiGroup.R:12-14:
    no `groups` argument = one pre-group; src/umi_group.cpp:35) -- the one case of the path with a real exchange.  Row tiles of the
    all-pairs neighbour search per rank (sarlacc_dev_umi_pairs_shard), all-gather of the neighbour pairs over RCCL (device tensors,
    the pairs never leave HBM), the exact clustering replicated on every rank, the clusters dealt to the ranks by their bases for
    multiReadAlign + consensusReadSeq.  Every rank holds every rank's reads in HBM (on a real run every rank reads the same FASTQ
    files; 4 GB per 10^6 reads), so regrouping reads by cluster moves nothing.  Weak scaling like the rest of the line:
    N x --giant-molecules molecules.

    FAIL-SAFE: the leg never takes the bench line down.  Every local step runs under try/except and the ranks agree on its outcome
    (shard.agree: a 4-byte all-reduce) BEFORE the next collective, so a failure on one rank -- out of memory, the 2^32-link stop --
    is raised on all of them together instead of leaving the others waiting; whatever is raised ends up as {"error": ...} in the
    line, which still carries the headline and the pipeline legs.

    `--giant-virtual-ranks V` on ONE GPU (world == 1) rehearses the leg at the size an N = V run has: V shares of reads, the whole
    tile search, the exchange through a one-rank RCCL group, the replicated clustering at full size, the clusters dealt to V
    owners and the MSA + consensus of owner 0's share -- the single-GPU cost of every step of an N = V run but the xGMI transfer."""
    import torch
    from sarlacc_amd import shard
    virtual = world == 1
    own_group = False
    try:
        if virtual and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group(backend, rank=0, world_size=1)
            own_group = True
        return _giant_group_leg(args, rank, world, dist, device, gather_device, backend, enc, mol, fence, reduce)
    except Exception as e:   # noqa: BLE001 -- the leg reports, it never raises (see the docstring)
        import traceback
        free_b, total_b = torch.cuda.mem_get_info()
        mem = {"hbm_free_gb": free_b / 1e9, "hbm_total_gb": total_b / 1e9, "torch_allocated_gb": torch.cuda.memory_allocated() / 1e9,
               "torch_reserved_gb": torch.cuda.memory_reserved() / 1e9}
        torch.cuda.empty_cache()
        return {"error": "%s: %s" % (type(e).__name__, e), "where": traceback.format_exc(limit=3).strip().splitlines()[-3:], "memory_at_failure": mem,
                "note": "the giant pre-group leg failed (agreed by all ranks before the next collective); the rest of the line is unaffected"}
    finally:
        if own_group:
            dist.destroy_process_group()


def _giant_group_leg(args, rank, world, dist, device, gather_device, backend, enc, mol, fence, reduce):
    import numpy as np
    import torch
    from sarlacc_amd import _lib, calls, devsynth, shard
    from sarlacc_amd import device as sdev
    from sarlacc_amd.strset import StringSet
    shares = args.giant_virtual_ranks if world == 1 else world   # shares of reads = owners of clusters
    molecules = args.giant_molecules if args.giant_molecules > 0 else args.molecules
    # share r is what rank r's own pipeline pass ran on (seed 2000 + r) when the sizes are the pipeline's
    err = None
    seq = qual = off = umis = widths = None
    n_all = 0
    try:
        seqs, quals, umis_c, lens, ulens = [], [], [], [], []
        for r in range(shares):
            reuse = mol is not None and r == rank and molecules == args.molecules
            sh = mol if reuse else devsynth.make_molecule_reads(molecules, args.copies, args.read_len, seed=2000 + r, device=device)
            seqs.append(sh["seq"]); quals.append(sh["qual"])
            lens.append(torch.diff(sh["off"]).cpu().numpy())
            umis_c.append(sh["umi"].cpu().numpy()); ulens.append(torch.diff(sh["umi_off"]).cpu().numpy())
            del sh
        seq = torch.cat(seqs)
        del seqs
        qual = torch.cat(quals)
        del quals
        off = np.zeros(sum(x.size for x in lens) + 1, np.int64)
        np.cumsum(np.concatenate(lens), out=off[1:])
        uoff = np.zeros(off.size, np.int64)
        np.cumsum(np.concatenate(ulens), out=uoff[1:])
        umis = StringSet(np.concatenate(umis_c), uoff)
        n_all = off.size - 1
        widths = np.diff(off)
        torch.cuda.empty_cache()   # (the generator's temporaries and the shares' own copies go back to the device: the library allocates beside torch)
    except Exception as e:   # noqa: BLE001
        err = e
    shard.agree(dist, err, gather_device)

    def one_pass():
        st = {}
        fence()
        t0 = time.perf_counter()
        coff, cmem = shard.sharded_umi_group_tiles(umis, args.threshold, calls, dist, gather_device, flat=True, stats=st)   # (agrees by itself)
        t1 = time.perf_counter()
        e1, res = None, None
        try:
            umi_stats = {k: _lib.stage_count(k) for k in ("umi_links", "umi_cluster_rounds", "umi_adjacency_s", "umi_cluster_s")}
            umi_ms = _lib.stage_ms("umi_pairs")
            umi_ws = _lib.release_umi_workspace()   # (17 GB at 8 x 10^6 reads: the MSA stage's batches are sized from what is free)
            sizes = np.diff(coff)
            big = np.flatnonzero(sizes >= 2)
            bases = np.add.reduceat(widths[cmem.astype(np.int64) - 1], coff[:-1]) if cmem.size else np.zeros(0)
            owner = shard.assign_groups_snake(bases[big], shares)
            keep = np.zeros(sizes.size, bool)
            keep[big[owner == rank]] = True
            goff, gflat = calls.csr_select(coff, cmem, keep)
            t2 = time.perf_counter()
            cons, phred = sdev.dev_msa_consensus(goff, gflat, seq, qual, off, 0, -1, -5, -1, 100, 0.6, encoding=enc)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            free_b, total_b = torch.cuda.mem_get_info()
            ws_total, ws_top = _lib.workspace_report(10)
            res = {"t": (t1 - t0, t2 - t1, t3 - t2, t3 - t0), "ws": {"library_workspaces_gb": ws_total / 1e9, "largest_gb": {n: b / 1e9 for n, b in ws_top},
                                                                     "torch_allocated_gb": torch.cuda.memory_allocated() / 1e9, "torch_reserved_gb": torch.cuda.memory_reserved() / 1e9}, "st": st, "coff": coff, "cmem": cmem, "big": big, "owner": owner,
                   "cons": cons, "phred": phred, "gflat": gflat, "largest_cluster": int(sizes.max()) if sizes.size else 0,
                   "hbm_in_use_gb": (total_b - free_b) / 1e9, "v1_fallback": _lib.stage_count("msa_v1_fallback"),
                   "kernel_ms": dict({k: _lib.stage_ms(k) for k in ("msa_pairwise", "msa_merge", "consensus")}, umi_pairs=umi_ms),
                   "umi": umi_stats, "umi_workspace_gb": umi_ws / 1e9,
                   "msa": dict({k: _lib.stage_count("msa_host_%s_s" % k) for k in ("plan", "upload_alloc", "pairwise_launch", "rows", "total")},
                               **{k: _lib.stage_count("msa2_" + k) for k in ("batches", "groups_second_pass", "first_exit_s", "last_exit_s", "exit_s_1wave", "exit_s_8waves", "mem_all_gb", "mem_budget_gb")},
                               pairs=_lib.stage_count("msa_pairs"), pairs_run_again=_lib.stage_count("msa_bitvector_redone"))}
        except Exception as e:   # noqa: BLE001
            e1 = e
        shard.agree(dist, e1, gather_device)
        fence()
        return res

    first = one_pass()
    first_t = first["t"][3]
    del first
    r = one_pass()
    tm = reduce(list(r["t"]) + [first_t, r["st"].get("search_s", 0.0), r["st"].get("exchange_s", 0.0), r["st"].get("clustering_s", 0.0)]
                + [r["kernel_ms"][k] for k in ("umi_pairs", "msa_pairwise", "msa_merge", "consensus")] + [r["hbm_in_use_gb"]], dist.ReduceOp.MAX)
    sm = reduce([float(len(r["cons"])), float(r["cons"].total), float(r["gflat"].size), float(r["st"].get("bytes_received", 0)),
                 float(r["st"].get("pairs_here", 0)), 1.0, float(r["v1_fallback"])], dist.ReduceOp.SUM)
    # ---- identical to a single rank?  The clusters: rank 0 runs the unsharded umi_group on the same UMIs.  The consensus reads:
    # a sample of clusters spread over the owners, recomputed by rank 0 (it holds every read) and compared with the owners' strings.
    big, owner = r["big"], r["owner"]
    pick = np.unique(np.linspace(0, big.size - 1, num=min(big.size, 256 * shares)).astype(np.int64)) if big.size else np.zeros(0, np.int64)
    mine_rank = np.cumsum(owner == rank) - 1   # position of an owned cluster among this rank's clusters (csr_select keeps the order)
    sample = {}
    for k in pick.tolist():
        if owner[k] == rank:
            j = int(mine_rank[k])
            sample[int(k)] = (bytes(r["cons"].chars[r["cons"].off[j]:r["cons"].off[j + 1]]), bytes(r["phred"].chars[r["phred"].off[j]:r["phred"].off[j + 1]]))
    gathered = [None] * max(world, 1)
    dist.all_gather_object(gathered, sample)   # (all_gather underneath: the one object collective every backend here carries)
    res = None
    if rank == 0:   # (no collective below this line)
        t0 = time.perf_counter()
        ref_off, ref_mem = calls.umi_group_flat(umis, args.threshold, None, args.threshold, np.array([0, n_all], np.int64),
                                                np.arange(1, n_all + 1, dtype=np.int32))
        single_s = time.perf_counter() - t0
        clusters_same = bool(np.array_equal(ref_off, r["coff"]) and np.array_equal(ref_mem, r["cmem"]))
        if world == 1:   # (rehearsal: only owner 0's share was aligned here)
            pick = pick[owner[pick] == 0]
        keep = np.zeros(r["coff"].size - 1, bool)
        keep[big[pick]] = True
        goff_s, gflat_s = calls.csr_select(r["coff"], r["cmem"], keep)
        cons_s, phred_s = sdev.dev_msa_consensus(goff_s, gflat_s, seq, qual, off, 0, -1, -5, -1, 100, 0.6, encoding=enc)
        got = {}
        for d in gathered:
            got.update(d)
        same = len(got) == pick.size
        for j, k in enumerate(pick.tolist()):   # (csr_select keeps the order of the clusters, so sample j is cluster pick[j])
            a = (bytes(cons_s.chars[cons_s.off[j]:cons_s.off[j + 1]]), bytes(phred_s.chars[phred_s.off[j]:phred_s.off[j + 1]]))
            same = same and got.get(int(k)) == a
        wall = tm[3]
        res = {"reads": int(n_all), "reads_per_min": n_all / wall * 60.0, "seconds": wall, "first_pass_seconds": tm[4],
               "stage_s": {"umi_group_tiles_exchange_clustering": tm[0], "deal_clusters": tm[1], "msa_consensus": tm[2]},
               "umi_group_s": {"tile_search": tm[5], "pair_exchange": tm[6], "replicated_clustering": tm[7],
                               "replicated_clustering_kernels_only": {"adjacency_from_pairs": r["umi"]["umi_adjacency_s"], "greedy_rounds": r["umi"]["umi_cluster_s"],
                                                                      "rounds": int(r["umi"]["umi_cluster_rounds"]), "links": r["umi"]["umi_links"]},
                               "unsharded_umi_group_on_rank_0": single_s},
               "kernel_ms": dict(zip(("umi_pairs", "msa_pairwise", "msa_merge", "consensus"), tm[8:12])),
               "exchange": {"collective": "all_gather_into_tensor of the neighbour pairs (8 B each) + their counts", "backend": backend,
                            "pairs": int(sm[4]), "bytes_received_total": int(sm[3]), "seconds": tm[6],
                            "pairs_stay_in_hbm": bool(r["st"].get("pairs_on_device", False))},
               "n_ranks_seen": int(sm[5]), "clusters": int(r["coff"].size - 1), "clusters_of_two_and_more": int(big.size),
               "largest_cluster": r["largest_cluster"], "groups_aligned_by_spec_v1": int(sm[6]), "hbm_in_use_gb_max": tm[12],
               "umi_workspace_released_before_msa_gb": r["umi_workspace_gb"], "msa_stage_rank0": r["msa"], "memory_rank0": r["ws"],
               "consensus_reads": int(sm[0]), "consensus_bases": int(sm[1]), "reads_in_clusters": int(sm[2]),
               "identical_to_single_rank": {"clusters": clusters_same, "consensus_of_sampled_clusters": bool(same), "sampled_clusters": int(pick.size)},
               "workload": "BASELINE configs[4] as worded, weak scaling: the %d x %d reads of all ranks as ONE pre-group (umiGroup without "
                           "`groups`), threshold %d: tile-sharded neighbour search -> all-gather of the pairs -> replicated exact clustering -> "
                           "clusters dealt to the ranks by their bases -> multiReadAlign + consensusReadSeq; every rank holds every read in HBM"
                           % (shares, molecules * args.copies, args.threshold)}
        if world == 1:
            res["rehearsal"] = ("ONE GPU standing in for rank 0 of %d: the whole tile search (an N = %d rank does 1/%d of it), the exchange through a one-rank "
                                "RCCL group, the replicated clustering at full size, owner 0's share of the clusters aligned; reads_per_min is not a "
                                "throughput claim" % (shares, shares, shares))
    del seq, qual
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (DP workload)")
    ap.add_argument("--read-len", type=int, default=2000)
    ap.add_argument("--molecules", type=int, default=100_000, help="molecules per GPU (pipeline workload, x --copies reads)")
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--threshold", type=int, default=1, help="umiGroup threshold of the pipeline workload")
    ap.add_argument("--cpu-sample", type=int, default=12000, help="reads per host core in the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true")
    ap.add_argument("--no-host-pointer", action="store_true", help="skip the PCIe-inclusive host-pointer figures")
    ap.add_argument("--no-quality-align", action="store_true", help="skip the qualityAlign-shaped leg (2-kb reads against a 2-kb reference, k_align_wide_q)")
    ap.add_argument("--giant-molecules", type=int, default=0,
                    help="molecules per GPU of the giant pre-group leg (N > 1; 0 = --molecules).  Rehearsed on one GPU up to 10^6 "
                         "molecules = 10^7 reads in one pre-group (profiles/r05_giant_*): the default is inside that at N = 8")
    ap.add_argument("--giant-virtual-ranks", type=int, default=0,
                    help="on ONE GPU: rehearse the giant pre-group leg at the size of an N = V run (see giant_group_leg)")
    ap.add_argument("--no-giant", action="store_true", help="skip the giant pre-group leg of N > 1 runs")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    share = os.environ.get("SARLACC_BENCH_SHARE_GPU") == "1"   # rehearsal: several ranks on one card (never a measurement)
    ndev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device")
    if args.gpus > ndev and not share:
        raise SystemExit("--gpus %d, but only %d HIP device(s) are visible" % (args.gpus, ndev))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher: start the ranks ourselves -- as a fresh CHILD process, before anything in this
        # process has touched the GPU (never re-exec a process that has) -- relay its output and exit with its code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SARLACC_DIST_BACKEND", "nccl")  # gloo only to exercise this path on one GPU
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)   # before the process group: RCCL binds its communicator to the current device
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    red_device = device if backend == "nccl" else torch.device("cpu")
    D = dist if world > 1 else None

    import sarlacc_amd
    from sarlacc_amd import calls, devsynth, pipeline
    from sarlacc_amd import device as sdev
    from sarlacc_amd.strset import StringSet
    sarlacc_amd.set_device(dev_index)
    enc = sarlacc_amd.phred_encoding()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce(vals, op):
        t = torch.tensor(vals, dtype=torch.float64, device=red_device)
        if world > 1:
            dist.all_reduce(t, op=op)
        return t.cpu().tolist()

    # ------------------------------------------------------------------ DP (configs[1]) -------
    n = args.reads
    seq, qual, off, max_len = devsynth.make_reads(n, args.read_len, ADAPTOR1, ADAPTOR2, seed=1000 + rank, device=device)
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

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        kern_ms.append(step())
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = reduce([elapsed], dist.ReduceOp.MAX)[0]
    all_cells = reduce([float(cells)], dist.ReduceOp.SUM)[0]

    out = None
    if rank == 0:
        gcups = all_cells * args.steps / elapsed / 1e9
        # algorithmic bytes per alignment: 2-bit packed bases + 1 B
        # quality per base in, 8 B score + 8 B start/end + 8 B per section out
        alg_bytes = total_bases / 4.0 + total_bases + n * 24.0
        k_ms = sum(kern_ms) / len(kern_ms)
        hbm = alg_bytes / (k_ms * 1e-3) / 1e9
        ops = cells * ALIGN_OPS_PER_CELL / (k_ms * 1e-3) / 1e12
        pmc = pmc_record("k_align", ["align.hip"]) if n == 1_000_000 and args.read_len == 2000 else {"traffic": None, "traffic_source": None}
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
            "roofline": {"kernel": "k_align", "bound": "valu", "achieved": ops, "peak": VALU64_PEAK_TLOPS,
                         "unit": "T lane-op/s (fp64)", "frac": ops / VALU64_PEAK_TLOPS,
                         "algorithmic_ops_per_cell": ALIGN_OPS_PER_CELL,
                         "traffic": pmc["traffic"], "traffic_source": pmc["traffic_source"], "pmc": pmc.get("derived"),
                         "traffic_ratio": pmc["traffic"] / alg_bytes if pmc["traffic"] else None,
                         "hbm": {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": hbm / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes},
                         "note": "fp64 DP recurrence: vector-issue bound; compulsory traffic is 0.042 B/cell, so the HBM "
                                 "fraction is small by construction and reported as a second entry"},
        }

    # PCIe-inclusive `.Call`-level figure: host pointers in, host pointers out (never `value`)
    if rank == 0 and not args.no_host_pointer:
        h_off = off.cpu().numpy()
        hs = StringSet(seq.cpu().numpy(), h_off)
        hq = StringSet(qual.cpu().numpy(), h_off.copy())
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            calls.adaptor_align(hs, hq, enc, GAP_OPEN, GAP_EXT, ADAPTOR1, *UMI_SECTION)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["host_pointer"] = {"gcups": cells / best / 1e9, "seconds": best,
                               "note": "sarlacc_adaptor_align on host buffers: chunked upload overlapped with the kernels, results downloaded"}
        del hs, hq
    # generic level (SURVEY section 8d ii): adaptorAlign on the resident reads -- front / back windows of `tolerance` = 250 bases
    # (R/adaptorAlign.R:86-95), four alignments per read with traceback and sections (:186-189), strand choice
    if rank == 0 and not args.no_host_pointer:
        from sarlacc_amd import generics
        from sarlacc_amd.resident import DevBuffer, DeviceReads
        h_off = off.cpu().numpy()
        dev_reads = DeviceReads(DevBuffer.borrow(seq.data_ptr(), total_bases), DevBuffer.borrow(qual.data_ptr(), total_bases),
                                DevBuffer.borrow(off.data_ptr(), 8 * (n + 1)), h_off, enc)
        sub1, sub2 = generics._setup_subseqs(ADAPTOR1), generics._setup_subseqs(ADAPTOR2)
        acc = {"kernel_ms": 0.0, "windows_s": 0.0, "subseq_s": 0.0, "wbases": 0}
        orig = {k: getattr(DeviceReads, k) for k in ("align_block", "front_and_back", "subseq")}

        def timed_align(self, *a, **k):
            out = orig["align_block"](self, *a, **k)
            acc["kernel_ms"] += sarlacc_amd.last_kernel_ms()
            return out

        def timed_windows(self, *a, **k):
            t = time.perf_counter()
            out = orig["front_and_back"](self, *a, **k)
            torch.cuda.synchronize()
            acc["windows_s"] += time.perf_counter() - t
            acc["wbases"] = int(out[0].total)
            return out

        def timed_subseq(self, *a, **k):
            t = time.perf_counter()
            out = orig["subseq"](self, *a, **k)
            acc["subseq_s"] += time.perf_counter() - t
            return out

        DeviceReads.align_block, DeviceReads.front_and_back, DeviceReads.subseq = timed_align, timed_windows, timed_subseq
        best = None
        try:
            for _ in range(2):
                for k in acc:
                    acc[k] = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cs, ce, rev, _w = generics._adaptor_align_chunk(ADAPTOR1, ADAPTOR2, sub1, sub2, (GAP_OPEN, GAP_EXT), 250, dev=dev_reads)
                dt = time.perf_counter() - t0
                wcells = 2 * acc["wbases"] * (len(ADAPTOR1) + len(ADAPTOR2))
                cur = {"seconds": dt, "gcups": wcells / dt / 1e9, "kernel_gcups": wcells / (acc["kernel_ms"] * 1e-3) / 1e9,
                       "reads_per_s": n / dt, "reversed": int(rev.sum()),
                       "breakdown_s": {"windows_on_device": acc["windows_s"], "four_alignments_kernels": acc["kernel_ms"] * 1e-3,
                                       "subsequences_cut_on_device_and_downloaded": acc["subseq_s"],
                                       "rest_strand_choice_on_device_and_result_downloads": dt - acc["windows_s"] - acc["kernel_ms"] * 1e-3 - acc["subseq_s"]}}
                best = cur if best is None or cur["seconds"] < best["seconds"] else best
                del cs, ce
        finally:
            for k, f in orig.items():
                setattr(DeviceReads, k, f)
        best["note"] = ("adaptorAlign on the resident batch: tolerance 250, adaptor1 x front, adaptor2 x back, adaptor1 x back, adaptor2 x front "
                        "(30- and 22-base adaptors), traceback + sections, strand resolution, sub-sequences of the chosen strand cut on the device; "
                        "cells = window bases x adaptor length")
        out["generic_level"] = best
        del dev_reads
    # qualityAlign-shaped call (R/qualityAlign.R:13-15 -> src/general_align.cpp): the batch's first reads, globally, against a
    # reference as long as they are -- beyond the 1 024 columns one wavefront holds, so one workgroup per alignment: k_align_wide_q
    # (global mode, gapopen >= 0, one strip: wavefronts hand the row state on through LDS queues)
    if rank == 0 and not args.no_quality_align:
        # (reads of ONE molecule against that molecule -- what qualityAlign is called on: mockReads' error process, 5 % substitutions
        # and 1 % indel events, on host arrays; qualities of the DP batch)
        nq = min(n, 20000)
        rng_q = np.random.default_rng(2)
        nucq = np.frombuffer(b"ACGT", np.uint8)
        qmol = nucq[rng_q.integers(0, 4, args.read_len)]
        qref = qmol.tobytes().decode()
        body = np.repeat(qmol[None, :], nq, axis=0)
        sub = rng_q.random(body.shape) < 0.05
        body[sub] = nucq[rng_q.integers(0, 4, int(sub.sum()))]
        cnt = np.ones(body.shape, np.int64)
        ind = rng_q.random(body.shape) < 0.01
        cnt[ind] = np.array([0, 2, 3, 4, 5])[rng_q.integers(0, 5, int(ind.sum()))]
        flat = np.repeat(body.reshape(-1), cnt.reshape(-1))
        h_off = np.zeros(nq + 1, np.int64)
        np.cumsum(cnt.sum(1), out=h_off[1:])
        end = int(h_off[-1])
        hs = StringSet(flat, h_off)
        hq = StringSet(qual[:end].cpu().numpy(), h_off.copy())
        del body, sub, cnt, ind
        best = best0 = None
        for _ in range(2):
            t0 = time.perf_counter()
            calls.general_align(hs, hq, enc, GAP_OPEN, GAP_EXT, qref, True)
            dt = time.perf_counter() - t0
            cur = {"seconds": dt, "kernel_ms": sarlacc_amd.last_kernel_ms()}
            best = cur if best is None or cur["kernel_ms"] < best["kernel_ms"] else best
            t0 = time.perf_counter()
            calls.barcode_align(hs, hq, enc, GAP_OPEN, GAP_EXT, qref)   # (the same alignments, scores only)
            dt = time.perf_counter() - t0
            cur = {"seconds": dt, "kernel_ms": sarlacc_amd.last_kernel_ms()}
            best0 = cur if best0 is None or cur["kernel_ms"] < best0["kernel_ms"] else best0
        qcells = float(end) * args.read_len
        pw = pmc_record("k_align_wide", ["align.hip"])
        q_ops = qcells * ALIGN_OPS_PER_CELL / (best["kernel_ms"] * 1e-3) / 1e12
        # algorithmic bytes of a traceback call: the read (1 B base + 1 B quality per row) in, 4 bits of code per cell written once
        # and about (rows + columns) / 64 words of them read back by the walk, two gapped strings out
        q_alg = 2.0 * end + 0.5 * qcells + 2.0 * (end + nq * args.read_len)
        out["quality_align_2kb"] = {
            "reads": nq, "reference_len": args.read_len, "seconds": best["seconds"], "kernel_ms": best["kernel_ms"],
            "gcups": qcells / best["seconds"] / 1e9, "kernel_gcups": qcells / (best["kernel_ms"] * 1e-3) / 1e9,
            "scores_only": {"seconds": best0["seconds"], "kernel_ms": best0["kernel_ms"], "kernel_gcups": qcells / (best0["kernel_ms"] * 1e-3) / 1e9,
                            "frac_of_fp64_valu_peak": qcells * ALIGN_OPS_PER_CELL / (best0["kernel_ms"] * 1e-3) / 1e12 / VALU64_PEAK_TLOPS},
            "roofline": {"bound": "valu", "kernel": "k_align_wide_q<8, 2> (edit distances: the codes of every cell + the walk)",
                         "achieved": q_ops, "peak": VALU64_PEAK_TLOPS, "unit": "T lane-op/s (fp64)", "frac": q_ops / VALU64_PEAK_TLOPS,
                         "algorithmic_ops_per_cell": ALIGN_OPS_PER_CELL, "algorithmic_bytes": q_alg,
                         "hbm": {"achieved": q_alg / (best["kernel_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": q_alg / (best["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "traffic": pw["traffic"], "traffic_source": pw["traffic_source"], "pmc": pw.get("derived"),
                         "traffic_ratio": pw["traffic"] / q_alg if pw["traffic"] else None},
            "note": "sarlacc_general_align (edit distances, traceback of every alignment) and sarlacc_barcode_align (scores) on host buffers, "
                    "%d reads of %d bases against a %d-base reference, global mode; k_align_wide_q: one workgroup per alignment, thread = 8 "
                    "reference columns, DPP inside a wavefront and LDS queues between wavefronts" % (nq, args.read_len, args.read_len)}
        del hs, hq
    cpu_sample = None
    if rank == 0 and not args.no_cpu:
        # the GPU box gives a one-GPU job a share of 16 host cores (more threads than that only contend)
        cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
        # the sample stays in HBM until the CPU leg at the end: pulled to the host here (some 10^5 Python strings, 0.8 GB through
        # the allocator) the pipeline passes below measured 20-30 ms slower in their host-side copies
        k = min(n, args.cpu_sample * cores)
        end = int(off[k].item())
        cpu_sample = (seq[:end].clone(), qual[:end].clone(), off[:k + 1].clone(), k, cores)
    del seq, qual, off, packed, nmask, scores, starts, ends, sso, swo
    torch.cuda.empty_cache()
    sarlacc_amd._lib.lib().sarlacc_release_workspace()

    # ------------------------------------------------------------ pipeline (configs[2..4]) -----
    if not args.no_pipeline:
        mol = devsynth.make_molecule_reads(args.molecules, args.copies, args.read_len, seed=2000 + rank, device=device)
        nr = mol["off"].numel() - 1
        off_host = mol["off"].cpu().numpy()
        umis = StringSet(mol["umi"].cpu().numpy(), mol["umi_off"].cpu().numpy())
        gather_device = device if backend == "nccl" else None
        runs = []
        for _ in range(2):
            fence()
            t0 = time.perf_counter()
            r = pipeline.run_resident(umis, mol["seq"], mol["qual"], off_host, enc, threshold=args.threshold, dist=D,
                                      gather_device=gather_device)
            fence()
            r["wall"] = time.perf_counter() - t0
            runs.append(r)
            if len(runs) == 1:   # (only the first pass's wall time is reported: its strings go back before the timed pass)
                runs[0] = {"wall": r["wall"]}
                del r
        # the same pass with the centre-star alignment (spec v1, round 1's algorithm) for comparison, on every rank
        calls.set_msa_spec(1)
        try:
            r1 = None
            for _ in range(2):   # (the first pass sizes spec v1's own workspaces)
                r1 = None        # (its strings go back to the page-locked pool before the timed pass asks for its own)
                fence()
                t0 = time.perf_counter()
                r1 = pipeline.run_resident(umis, mol["seq"], mol["qual"], off_host, enc, threshold=args.threshold, dist=D,
                                           gather_device=gather_device)
                fence()
                dt1 = time.perf_counter() - t0
        finally:
            calls.set_msa_spec(0)
        v1 = reduce([dt1, r1["kernel_ms"]["msa_pairwise"], r1["kernel_ms"]["msa_merge"]], dist.ReduceOp.MAX)
        first, last = runs[0], runs[-1]
        names = ["umi_group", "label_exchange", "host_glue", "msa_consensus", "total"]
        kn = ["umi_pairs", "msa_pairwise", "msa_merge", "consensus"]
        mx = reduce([last["wall"], first["wall"], last["all_gather_s"]] + [last["stage_s"][k] for k in names]
                    + [last["kernel_ms"][k] for k in kn], dist.ReduceOp.MAX)
        sm = reduce([float(nr), float(len(last["cons"])), float(last["cons"].total), float(last["gflat"].size),
                     float(last["all_gather_bytes"])], dist.ReduceOp.SUM)
        if rank == 0:
            wall = mx[0]
            kms = dict(zip(kn, mx[3 + len(names):]))
            cnt = last["counts"]
            cons_cols = last["cons"].total
            msa_ops = cnt["msa_cells"] * MSA_OPS_PER_CELL / (kms["msa_pairwise"] * 1e-3) / 1e12
            cons_bytes = 2.0 * cnt["consensus_cells"] + 2.0 * cons_cols
            cons_gbs = cons_bytes / (kms["consensus"] * 1e-3) / 1e9
            # algorithmic bytes of the pairwise stage under spec v2: the clustered reads in (1 B per base) and both position maps of
            # every pair out (2 B per base of either read)
            msa_alg_bytes = float(args.read_len) * (float(last["gflat"].size) + 4.0 * cnt["msa_pairs"])
            pm = pmc_record("k_msa_pairwise", ["msa_pairwise.hip", "msa_common.hpp"])
            pw_s = kms["msa_pairwise"] * 1e-3
            if cnt.get("msa_pairs_bitvector", 0) >= 0.5 * cnt["msa_pairs"]:
                # the default scores are unit Levenshtein costs: k_msa_pairwise_bv (bit vectors, one pair per lane) + k_msa_moves_expand.
                # What it moves is its own traceback: 2 bits per cell of the band, written once, a quarter of it read back.
                tile = cnt["msa_bitvector_tile_bytes"]
                msa_roof = {"bound": "hbm", "kernel": "k_msa_pairwise_bv + k_msa_moves_expand (bit-vector edit distance, DESIGN.md section 4.5b)",
                            "achieved": msa_alg_bytes / pw_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": msa_alg_bytes / pw_s / 1e9 / HBM_PEAK_GBS,
                            "algorithmic_bytes": msa_alg_bytes,
                            "traceback_bytes_written": tile, "traceback_write_gbs": tile / pw_s / 1e9,
                            "traceback_write_frac_of_peak": tile / pw_s / 1e9 / HBM_PEAK_GBS,
                            "tcups": cnt["msa_cells"] / pw_s / 1e12, "pairs_per_s": cnt["msa_pairs"] / pw_s,
                            "pairs_on_bit_vectors": cnt["msa_pairs_bitvector"],
                            "pairs_run_again_with_whole_records": cnt.get("msa_bitvector_redone", 0.0),
                            "traffic": pm["traffic"], "traffic_source": pm["traffic_source"], "pmc": pm.get("derived"),
                            "traffic_ratio": pm["traffic"] / msa_alg_bytes if pm["traffic"] else None,
                            "note": "compulsory traffic (reads in, position maps out) is a few per cent of what the kernel writes as traceback records, "
                                    "so the fraction of peak of the records is given beside the contract's algorithmic fraction"}
            else:
                msa_roof = {"bound": "valu", "kernel": "k_msa_pairwise_pk (packed 16-bit DP)", "achieved": msa_ops, "peak": VALU32_PEAK_TLOPS,
                            "unit": "T lane-op/s (int32)", "frac": msa_ops / VALU32_PEAK_TLOPS, "algorithmic_ops_per_cell": MSA_OPS_PER_CELL,
                            "tcups": cnt["msa_cells"] / pw_s / 1e12,
                            "traffic": pm["traffic"], "traffic_source": pm["traffic_source"], "pmc": pm.get("derived"),
                            "algorithmic_bytes": msa_alg_bytes, "traffic_ratio": pm["traffic"] / msa_alg_bytes if pm["traffic"] else None}
            pc = pmc_record("k_consensus_code", ["consensus.hip", "msa_common.hpp"])
            # The merge stage of spec v2: k_m2_extend* (the extended library, once per group) + k_m2_group (the
            # progressive merging).  Algorithmic bytes: both position maps of every pair read once (2 B per base of either read), plane 0
            # of the library written once and read once (4 B per pair and base of its first-child read, twice; the further planes are
            # touched only where a base has several partners), the columns of every base written once (4 B), the profiles' positions
            # (2 B per cell of the final alignment) and the rows out as vote codes (2 B per cell).  The merging itself is bound by the
            # round trips of its small gathers, not the bytes: second entry, wave-wide gather instructions of its rows phase against
            # one per 16 cycles and CU (a wave64 dword gather occupies the address unit for 16 cycles).
            gsz = np.diff(last["goff"]).astype(np.float64)
            gbases = np.add.reduceat(np.diff(off_host)[last["gflat"].astype(np.int64) - 1], last["goff"][:-1]).astype(np.float64) if gsz.size else np.zeros(0)
            m2_alg_bytes = float((2.0 * (gsz - 1.0) * gbases * 2.0).sum() + (0.5 * (gsz - 1.0) * gbases * 4.0 * 2.0).sum()
                                 + 4.0 * gbases.sum() + 4.0 * cnt["consensus_cells"])
            pg = pmc_record("k_m2_group", ["msa2.hip", "msa_common.hpp"])
            mg_s = kms["msa_merge"] * 1e-3
            gather_peak = 256 * 2.4e9 / 16.0
            cyc = {k: cnt.get("msa2_cycles_" + k, 0.0) for k in ("rows", "chain", "walk", "renumber")}
            cyc_all = sum(cyc.values()) or 1.0
            m2_roof = {"bound": "hbm", "kernel": "k_m2_first + k_m2_extend / k_m2_extend_unit (extended library) + k_m2_group (1-, 4- and 8-wavefront instantiations side by side) + k_m2_tree / k_m2_init",
                       "achieved": m2_alg_bytes / mg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": m2_alg_bytes / mg_s / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes": m2_alg_bytes, "traffic": pg["traffic"], "traffic_source": pg["traffic_source"], "pmc": pg.get("derived"),
                       "traffic_ratio": pg["traffic"] / m2_alg_bytes if pg["traffic"] else None,
                       "gather_issue": {"bound": "gather request rate", "achieved": cnt.get("msa2_gathers", 0.0) / mg_s, "peak": gather_peak,
                                        "unit": "wave-wide gathers/s", "frac": cnt.get("msa2_gathers", 0.0) / mg_s / gather_peak,
                                        "gathers": cnt.get("msa2_gathers", 0.0),
                                        "note": "rows phase of k_m2_group only, a lower bound (records with several partners gather more); one wave64 gather per 16 cycles and CU x 256 CUs x 2.4 GHz"},
                       "phase_share_of_wavefront_cycles": {k: v / cyc_all for k, v in cyc.items()},
                       "groups_second_pass": cnt.get("msa2_groups_second_pass", 0.0), "batches": cnt.get("msa2_batches", 1.0)}
            if cnt.get("msa2_batches", 1.0) > 1.0:   # (pipelined batches: the alignments of batch k + 1 run under the merging of batch k)
                for rf in (m2_roof, msa_roof):
                    rf["note_batches"] = "the call was cut into %d pipelined batches: the stage timers overlap, the fractions are lower bounds" % int(cnt["msa2_batches"])
            n_seen = dist.get_world_size() if world > 1 else 1
            out["pipeline"] = {
                "reads": int(sm[0]), "reads_per_min": sm[0] / wall * 60.0, "seconds": wall, "first_pass_seconds": mx[1],
                "consensus_reads": int(sm[1]), "consensus_bases": int(sm[2]), "reads_in_clusters": int(sm[3]),
                "stage_s": dict(zip(names, mx[3:3 + len(names)])), "kernel_ms": kms,
                "umi_group_host_s": last["umi_group_host_s"],
                "all_gather": {"seconds": mx[2], "bytes_received_total": int(sm[4]), "backend": backend if world > 1 else None},
                "n_ranks_seen": n_seen,
                "clusters_all_ranks": last["clusters_all_ranks"],
                "workload": "%d molecules x %d reads x %d bp per GPU generated in HBM, %d-bp UMIs: umi_group(threshold %d, one "
                            "pre-group per GPU) -> label all-gather -> msa_consensus (quick_msa bandwidth 100 + quality "
                            "consensus, rows stay in HBM); reads and qualities resident, UMIs / group lists / consensus "
                            "strings cross PCIe" % (args.molecules, args.copies, args.read_len, 12, args.threshold),
                "msa_spec": "v2 (all-pairs + consistency library + guide tree + progressive merging; DESIGN.md section 5)",
                "msa_groups_aligned_by_spec_v1": cnt["msa_v1_fallback"],
                "msa_pairs": cnt["msa_pairs"], "msa_cells": cnt["msa_cells"], "consensus_cells": cnt["consensus_cells"],
                "rooflines": {
                    "k_msa_pairwise": msa_roof,
                    "k_m2_group": m2_roof,
                    "k_consensus_code": {"bound": "hbm", "achieved": cons_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": cons_gbs / HBM_PEAK_GBS, "algorithmic_bytes": cons_bytes,
                                       "traffic": pc["traffic"], "traffic_source": pc["traffic_source"], "pmc": pc.get("derived"),
                                       "traffic_ratio": pc["traffic"] / cons_bytes if pc["traffic"] else None},
                },
            }
            out["pipeline"]["msa2_rules"] = {
                "rows": cnt["msa2_rows"], "rows_capped": cnt["msa2_rows_capped"], "rows_filtered": cnt["msa2_rows_filtered"],
                "entries_filtered": cnt["msa2_entries_filtered"], "entries_kept": cnt["msa2_entries_kept"], "joins": cnt["msa2_joins"],
                "joins_chain_in_hbm": cnt["msa2_joins_chain_in_hbm"],
                "note": "how often spec v2's own rules act in this pass: rows (profile columns of a first child) that met a 17th "
                        "partner column, library entries dropped as lighter than half their row's heaviest; joins whose chain "
                        "left the LDS ring (DESIGN.md section 5)"}
            out["pipeline"]["spec_v1"] = {"reads_per_min": sm[0] / v1[0] * 60.0, "seconds": v1[0],
                                          "kernel_ms": {"msa_pairwise": v1[1], "msa_merge": v1[2]},
                                          "msa_pairs": r1["counts"]["msa_pairs"],
                                          "tcups": r1["counts"]["msa_cells"] / (r1["kernel_ms"]["msa_pairwise"] * 1e-3) / 1e12}
            if not args.no_host_pointer:
                hs = StringSet(mol["seq"].cpu().numpy(), off_host)
                hq = StringSet(mol["qual"].cpu().numpy(), off_host.copy())
                t0 = time.perf_counter()
                coff, cmem = calls.umi_group_flat(umis, args.threshold, None, args.threshold, np.array([0, nr], np.int64),
                                                  np.arange(1, nr + 1, dtype=np.int32))
                goff, gflat = calls.csr_select(coff, cmem, np.diff(coff) >= 2)
                hc, _ = calls.msa_consensus_flat(goff, gflat, hs, 0, -1, -5, -1, 100, 0.6, quals=hq, encoding=enc)
                dt = time.perf_counter() - t0
                same = np.array_equal(hc.off, last["cons"].off) and np.array_equal(hc.chars[:hc.total], last["cons"].chars[:hc.total])
                out["pipeline"]["host_pointer"] = {"reads_per_min": nr / dt * 60.0, "seconds": dt,
                                                   "identical_to_resident": bool(same),
                                                   "note": "same stages through the host-pointer C ABI (reads + qualities cross PCIe), rank 0 only"}
                del hs, hq
        # BASELINE configs[2] (umiGroup on the batch's UMIs as one pre-group) at umiGroup's own default threshold 3
        # (R/umiGroup.R:2) and at 2, beside the threshold the pipeline workload uses; second call timed, rank 0's UMIs
        if rank == 0:
            from sarlacc_amd import _lib as _L
            ug = {}
            for thr in sorted({1, 2, 3, args.threshold}):
                for _ in range(2):
                    torch.cuda.synchronize()   # rank 0 only: no barrier here
                    t0 = time.perf_counter()
                    coff_t, _cm = calls.umi_group_flat(umis, thr, None, thr, np.array([0, nr], np.int64), np.arange(1, nr + 1, dtype=np.int32))
                    dt = time.perf_counter() - t0
                ug[str(thr)] = {"seconds": dt, "umis_per_s": nr / dt, "clusters": int(coff_t.size - 1), "links": _L.stage_count("umi_links"),
                                "search_kernels_ms": _L.stage_ms("umi_pairs"), "clustering_s": _L.stage_count("umi_cluster_s"),
                                "rounds": int(_L.stage_count("umi_cluster_rounds")),
                                "split_key_search": bool(_L.stage_count("umi_split_search"))}
            out["pipeline"]["umi_group_thresholds"] = ug
        # BASELINE configs[3] as it is worded: 100k groups x 10 reads x 2 kb -- the molecules themselves as groups (no UMI
        # clustering in front, so no clusters of several molecules)
        goff_p = np.arange(0, nr + 1, args.copies, dtype=np.int64)
        gflat_p = np.arange(1, nr + 1, dtype=np.int32)
        c4 = cons_p = _ph = None
        for _ in range(2):
            cons_p = _ph = None
            fence()
            t0 = time.perf_counter()
            cons_p, _ph = sdev.dev_msa_consensus(goff_p, gflat_p, mol["seq"], mol["qual"], off_host, 0, -1, -5, -1, 100, 0.6, encoding=enc)
            fence()
            dt = time.perf_counter() - t0
            c4 = {"seconds": dt, "kernel_ms": {k: sarlacc_amd.stage_ms(k) for k in ("msa_pairwise", "msa_merge", "consensus")},
                  "consensus_reads": len(cons_p), "rows_capped": sarlacc_amd.stage_count("msa2_rows_capped"),
                  "entries_filtered": sarlacc_amd.stage_count("msa2_entries_filtered"), "rows": sarlacc_amd.stage_count("msa2_rows")}
        c4v = reduce([c4["seconds"]] + [c4["kernel_ms"][k] for k in ("msa_pairwise", "msa_merge", "consensus")], dist.ReduceOp.MAX)
        if rank == 0:
            out["pipeline"]["c4_pure_groups"] = {
                "reads_per_min": sm[0] / c4v[0] * 60.0, "seconds": c4v[0],
                "kernel_ms": dict(zip(("msa_pairwise", "msa_merge", "consensus"), c4v[1:])),
                "groups": int(goff_p.size - 1), "msa2_rows": c4["rows"], "msa2_rows_capped": c4["rows_capped"],
                "msa2_entries_filtered": c4["entries_filtered"],
                "workload": "multiReadAlign + consensusReadSeq on %d groups x %d reads x %d bp per GPU (the molecules as groups)" % (
                    args.molecules, args.copies, args.read_len)}
        if (world > 1 and not args.no_giant) or (world == 1 and args.giant_virtual_ranks > 0):
            del cons_p, _ph
            sarlacc_amd._lib.lib().sarlacc_release_workspace()
            gg = giant_group_leg(args, rank, world, dist, device, device if world == 1 else gather_device, backend, enc, mol, fence, reduce)
            if rank == 0:
                out["pipeline"]["giant_group"] = gg
        if rank == 0 and not args.no_cpu:
            cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
            h_seq, h_qual = mol["seq"].cpu().numpy(), mol["qual"].cpu().numpy()
            sizes = np.diff(last["goff"])
            pick = np.linspace(0, sizes.size - 1, num=min(12 * cores, sizes.size)).astype(np.int64)
            ids = sorted({int(i) for k in pick for i in last["gflat"][last["goff"][k]:last["goff"][k + 1]]})
            renum = {r: j + 1 for j, r in enumerate(ids)}
            rs = [h_seq[off_host[r - 1]:off_host[r]].tobytes().decode() for r in ids]
            qs = [h_qual[off_host[r - 1]:off_host[r]].tobytes().decode() for r in ids]
            groups_s = [[renum[int(i)] for i in last["gflat"][last["goff"][k]:last["goff"][k + 1]]] for k in pick]
            out["pipeline"]["cpu_baseline"] = cpu_baseline_pipeline(umis.to_strings()[:100000], args.threshold, groups_s, rs, qs, cores)
            del h_seq, h_qual
        del mol
        torch.cuda.empty_cache()

    if rank == 0:
        if cpu_sample is not None:
            cs, cq, co, k, cores = cpu_sample
            out["cpu_baseline"] = cpu_baseline(*(devsynth.to_host_strings(cs, cq, co, k) + (cores,)))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
