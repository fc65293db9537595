set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_msa.py -x -q -m gpu 2>&1 | tail -12
