#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -3
python tools/profile_whole_query.py --structures 203250 > gpurun_out/whole_query_profile_r3d.txt 2>&1
grep -v amdgpu.ids gpurun_out/whole_query_profile_r3d.txt | grep "fdgpu_retrieve\|query map" | tail -8
