import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sarlacc_amd
from sarlacc_amd import _lib, calls, devsynth, pipeline, device
from sarlacc_amd.strset import StringSet
dev = torch.device("cuda:0")
mol = devsynth.make_molecule_reads(100000, 10, 2000, seed=2000, device=dev)
off = mol["off"].cpu().numpy()
umis = StringSet(mol["umi"].cpu().numpy(), mol["umi_off"].cpu().numpy())
enc = sarlacc_amd.phred_encoding()
import cProfile, pstats
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    r = pipeline.run_resident(umis, mol["seq"], mol["qual"], off, enc, threshold=1)
    if rep == 2:
        pr.disable()
    dt = time.perf_counter() - t0
    print("rep %d %.4f s stage %s kernels %s host %s" % (rep, dt, {k: round(v, 4) for k, v in r["stage_s"].items()}, {k: round(v, 1) for k, v in r["kernel_ms"].items()},
          {k: round(_lib.stage_count("msa_host_%s_s" % k), 4) for k in ("plan", "upload_alloc", "pairwise_launch", "rows", "select", "total")}), flush=True)
    r = None
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
