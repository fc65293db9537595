cd tools
for DBG in 0 1 2 3 7; do
  echo "== SARLACC_MSA2_DBG=$DBG"
  SARLACC_MSA2_DBG=$DBG python perf_msa2.py ${1:-5000} 2000 10 2>&1 | grep rep | tail -1
done
