set -u
cd /root/repo
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q -m gpu -k "glue or retriev or fused or pipelined or shipped" 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/profile_round6_qt.sh 542000 24 "14" 2>&1 | grep -E "blocking|6 lanes|k_rs_|kernels per batch"
