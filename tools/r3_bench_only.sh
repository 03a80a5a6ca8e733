#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_e2e.py -m gpu -x -q > gpurun_out/t_e2e.log 2>&1; tail -2 gpurun_out/t_e2e.log
python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err; tail -c 300 gpurun_out/bench_r3_final.err
sed -n '/^python - <<.PY.$/,/^PY$/p' tools/r3_final.sh | sed '1d;$d' > /tmp/_sum.py; python /tmp/_sum.py
