set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_gpu_msa.py -x -q -m gpu > gpurun_out/t_msa.log 2>&1 || true
tail -4 gpurun_out/t_msa.log
timeout -k 10 900 python bench.py --no-cpu --no-host-pointer --steps 2 > gpurun_out/bench_bv.json 2> gpurun_out/bench_bv.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_bv.json') if l.startswith('{')][-1])
p=d['pipeline']
print('pipeline s', p['seconds'], 'reads/min', p['reads_per_min'])
print('kernel_ms', p['kernel_ms'])
print('c4', p['c4_pure_groups']['seconds'], p['c4_pure_groups'].get('kernel_ms'))
print(p['rooflines']['k_msa_pairwise'])
PY
