# SQ_INSTS_VALU of the pairwise kernel with / without the walk (SARLACC_MSA_DBG=1), 10^4 groups x 10 x 2 kb
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-exp_pmc}
mkdir -p $OUT
cat > /tmp/exp_msa_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.environ["REPO"] + "/tools")
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls, _lib
from perf_pipeline import NUC, noisy_copies
rng = np.random.default_rng(1000)
reads, quals = noisy_copies(NUC[rng.integers(0, 4, (10000, 2000))], 10, rng)
n = len(reads)
goff = np.arange(0, n + 1, 10, dtype=np.int64); gflat = np.arange(1, n + 1, dtype=np.int32)
try:
    calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=sarlacc_amd.phred_encoding())
except Exception as e:
    print("err", e)
print("pairwise %.2f ms" % _lib.stage_ms("msa_pairwise"))
PY
export REPO=$PWD
cd /tmp
for DBG in 7 0; do
  export SARLACC_MSA_DBG=$DBG
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/dbg$DBG -o pmc -- python3 /tmp/exp_msa_one.py > $OUT/dbg$DBG.log 2>&1
  python3 - <<PY
import csv, glob
tot = {}
for f in glob.glob("$OUT/dbg$DBG/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_msa_pairwise" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
print("dbg $DBG", {k: "%.4g" % v for k, v in tot.items()})
PY
done
