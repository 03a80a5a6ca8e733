#!/bin/bash
# two ranks on ONE GPU over gloo (plumbing check of `bench.py --gpus 2` with the automatic call plan; the numbers are not a scaling result)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
FD_BENCH_TRACE=1 FD_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-cli-index --no-replicas > gpurun_out/bench_r3_g2.json 2> gpurun_out/bench_r3_g2.err
grep "^\[bench\|Error\|error" gpurun_out/bench_r3_g2.err | tail -15
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_r3_g2.json").read().strip().splitlines()[-1]); q = d["query"]
    print("g2 build", round(d["value"]), round(d["ms_per_step"], 1), d["config"]["call_plan"], d["config"]["build_calls_per_rank"])
    print("g2 query", q.get("error") or {k: (round(q[k]["value"]) if isinstance(q.get(k), dict) and "value" in q[k] else None) for k in ("batched_with_matching", "batched", "single")}, q.get("exchange"))
except Exception as e:
    print("g2 failed", repr(e)); print(open("gpurun_out/bench_r3_g2.err").read()[-3000:])
PY
