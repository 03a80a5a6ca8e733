set -u
cd /root/repo
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q -m gpu -k "glue or retriev or fused or pipelined or shipped or tiled" 2>&1 | grep -E "passed|failed|error" | tail -3
for T in 512 256 128; do
  export FDGPU_QT_ROWS_T=$T
  echo "== FDGPU_QT_ROWS_T=$T"
  bash tools/profile_round6_qt.sh 542000 24 "14" 2>&1 | grep -E "blocking|6 lanes|k_qt_rows|k_rs_pack|k_rs_bases|k_rs_scan|kernels per batch"
done
