set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_gpu_msa.py -x -q -m gpu > gpurun_out/t_msa.log 2>&1 || true
tail -4 gpurun_out/t_msa.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/bvp -o bvp -- python3 /root/repo/tools/perf_msa2.py 100000 > /root/repo/gpurun_out/bvp.log 2>&1
cd /root/repo
grep "^rep" gpurun_out/bvp.log
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/bvp/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:5]: print(r['Name'][:80], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
