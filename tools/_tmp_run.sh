set -u
cd /root/repo
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "tiled or fused or pipelined or shipped" 2>&1 | grep -E "passed|failed|error|assert" | tail -5
bash tools/profile_round6_qt.sh 542000 24 "13 14" 2>&1 | grep -E "FDGPU_QT32|blocking|6 lanes|phase|k_qt_|kernels per batch"
