#!/bin/bash
# tools/r4_qtile.sh — parity of the tiled scoring, then the kernel trace of the batched query at 542,000 structures for both tile sizes
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "tiled or shipped or batch_of" 2>&1 | tail -5
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "count_query" 2>&1 | tail -3
for T in ${TILES:-13 14}; do
  FDGPU_QT_TILE=$T bash tools/profile_query_batch.sh
  echo "== tile 2^$T"; grep "k_qt_\|k_cq_\|k_pl_\|k_topn" gpurun_out/r2qb_kernels.txt | head -24
  cp gpurun_out/r2qb_kernels.txt gpurun_out/r4_qb_kernels_t$T.txt
  FDGPU_QT_TILE=$T FDGPU_QT_DBG=1 python tools/profile_query_host.py --structures 542000 --reps 2 --chunk 32 --no-profile 2>&1 | grep "^\[qt\]" | tail -3
done
