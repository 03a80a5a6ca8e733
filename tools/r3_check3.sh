#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
bash tools/r3_gloo2.sh
