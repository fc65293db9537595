for b in 1 2 4 6; do
  export SARLACC_MSA2_BATCHES=$b
  timeout -k 10 300 python tools/perf_pipeline_resident.py 100000 2 2 2>&1 | grep "rep 1" | sed "s/^/batches=$b /" | cut -c1-210
  timeout -k 10 300 python tools/perf_pipeline_resident.py 100000 2 2 pure 2>&1 | grep "pure rep 1" | sed "s/^/batches=$b /"
done
