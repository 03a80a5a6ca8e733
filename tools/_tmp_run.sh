set -u
cd /root/repo
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|assert" | tail -5
bash tools/profile_round6_qt.sh 542000 24 "14" 2>&1 | grep -E "blocking|6 lanes|k_qt_|kernels per batch"
