"""umi_group -> msa_consensus on device-generated molecules (bench.py's pipeline pass), stage and kernel-group times.
    python tools/perf_pipeline_resident.py [molecules] [spec] [reps] [pure|clusters] [UMI length]
A shorter UMI at fewer molecules reproduces the cluster sizes of a denser set: 50 000 molecules with 10-base UMIs collide like the
8 x 10^5 molecules of an 8-GPU giant pre-group with 12-base UMIs (molecules / 4^length = 0.048) at a sixteenth of the reads."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sarlacc_amd
from sarlacc_amd import _lib, calls, devsynth, pipeline
from sarlacc_amd.strset import StringSet

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
spec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pure = len(sys.argv) > 4 and sys.argv[4] == "pure"   # groups = the molecules themselves (no UMI clustering, no mixed clusters)
umi_len = int(sys.argv[5]) if len(sys.argv) > 5 else 12
dev = torch.device("cuda:0")
mol = devsynth.make_molecule_reads(G, 10, 2000, seed=2000, device=dev, umi_len=umi_len)
off = mol["off"].cpu().numpy()
umis = StringSet(mol["umi"].cpu().numpy(), mol["umi_off"].cpu().numpy())
enc = sarlacc_amd.phred_encoding()
calls.set_msa_spec(spec)
for kv in filter(None, os.environ.get("SARLACC_OPTS", "").split(",")):   # A/B switches, e.g. SARLACC_OPTS=msa_bitvector_tile_gb=12
    calls.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if pure:
    from sarlacc_amd import _lib, device
    n = off.size - 1
    goff = np.arange(0, n + 1, 10, dtype=np.int64)
    gflat = np.arange(1, n + 1, dtype=np.int32)
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cons, phred = device.dev_msa_consensus(goff, gflat, mol["seq"], mol["qual"], off, 0, -1, -5, -1, 100, 0.6, encoding=enc)
        dt = time.perf_counter() - t0
        print("pure rep %d spec %d: %.3f s  pairwise %.1f ms  merge %.1f ms  consensus %.1f ms" % (
            rep, spec, dt, _lib.stage_ms("msa_pairwise"), _lib.stage_ms("msa_merge"), _lib.stage_ms("consensus")), flush=True)
        print("      msa2 %s" % {k: _lib.stage_count("msa2_" + k) for k in ("rows", "rows_capped", "entries_filtered", "entries_kept", "joins",
              "joins_chain_in_hbm", "cycles_rows", "cycles_chain", "cycles_walk", "cycles_renumber", "first_exit_s", "last_exit_s")}, flush=True)
    sys.exit(0)
r = None
for rep in range(reps):
    r = None   # (the strings of the last pass go back to the page-locked pool before the next one asks for its own)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = pipeline.run_resident(umis, mol["seq"], mol["qual"], off, enc, threshold=1)
    dt = time.perf_counter() - t0
    sizes = np.diff(r["goff"])
    print("rep %d spec %d: %.3f s  (%.1f M reads/min)  kernel ms %s  groups %d (max size %d, >10 reads: %d)  v1 fallback %d" % (
        rep, spec, dt, (off.size - 1) / dt * 60 / 1e6, {k: round(v, 1) for k, v in r["kernel_ms"].items()}, sizes.size,
        sizes.max(), int((sizes > 10).sum()), int(r["counts"]["msa_v1_fallback"])), flush=True)
    print("      stage s %s  umi_group host s %s" % ({k: round(v, 4) for k, v in r["stage_s"].items()}, {k: round(v, 4) for k, v in r["umi_group_host_s"].items()}), flush=True)
    print("      msa host s %s" % {k: round(_lib.stage_count("msa_host_%s_s" % k), 4) for k in ("plan", "upload_alloc", "pairwise_launch", "rows", "total")}, flush=True)
    if rep == 0:
        h = np.bincount(sizes)
        print("      group sizes: %s" % {int(k): int(v) for k, v in enumerate(h) if v}, flush=True)
    from sarlacc_amd import _lib
    print("      msa2 %s" % {k: _lib.stage_count("msa2_" + k) for k in ("rows", "rows_capped", "entries_filtered", "rows_filtered", "entries_kept",
          "joins", "joins_chain_in_hbm", "groups_second_pass", "cycles_rows", "cycles_chain", "cycles_walk", "cycles_renumber", "launches", "first_exit_s", "last_exit_s", "exit_s_1wave", "exit_s_4waves", "exit_s_8waves")}, flush=True)
    print("      pairwise on bit vectors: %d pairs, %d run again with whole records" % (_lib.stage_count("msa_pairs_bitvector"), _lib.stage_count("msa_bitvector_redone")), flush=True)
