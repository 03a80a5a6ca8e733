#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/t_r3c.log 2>&1; tail -14 gpurun_out/t_r3c.log
python bench.py > gpurun_out/bench_r3c.json 2> gpurun_out/bench_r3c.err; tail -c 400 gpurun_out/bench_r3c.err
# two ranks on ONE GPU over gloo: the sharded query leg through the fused entry points + the replica leg (plumbing; not xGMI)
FD_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-export > gpurun_out/bench_r3c_g2.json 2> gpurun_out/bench_r3c_g2.err; tail -c 600 gpurun_out/bench_r3c_g2.err
python - <<'PY'
import json
for f in ("bench_r3c", "bench_r3c_g2"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        q = d["query"]
        print(f, "build", round(d["value"]), "ms", round(d["ms_per_step"], 1), "| query", q.get("error") or {k: (round(q[k]["value"]) if isinstance(q.get(k), dict) and "value" in q[k] else None) for k in ("batched_with_matching", "batched_with_matching_mt", "batched", "single", "with_matching", "replicas")}, q.get("exchange"))
    except Exception as e:
        print(f, "failed", repr(e))
PY
