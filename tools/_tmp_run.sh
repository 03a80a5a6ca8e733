set -u
cd /root/repo
python -m pytest tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|assert" | tail -5
