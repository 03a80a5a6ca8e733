#!/bin/bash
# quick check of a query-side change: the query-map / retrieval tests, then the stage split of a whole-structure query
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -m gpu -x -q --durations=5 -k "query_map or two_pass or retrieve or retrieval or glue or whole" > gpurun_out/t_query.log 2>&1; tail -12 gpurun_out/t_query.log
python tools/profile_whole_query.py --structures 203250 > gpurun_out/whole_q.log 2>&1; grep -v "amdgpu.ids" gpurun_out/whole_q.log | tail -40
