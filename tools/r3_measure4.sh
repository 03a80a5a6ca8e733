#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python - > gpurun_out/qmap_trace_r3.txt 2>&1 <<'PY'
import os, time, numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth
from folddisco_amd.api import PackedStructures, length_penalty
from folddisco_amd.query import make_query_map
dev = torch.device("cuda", 0)
S = 67750
d = synth.generate(S, seed=7, device=dev)
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ro = d["res_off"].contiguous()
batch = ctx.wrap_device(S, int(ro[-1].item()), ro.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=(ro, d))
ix = fd.FolddiscoIndex.build(ctx, batch)
roh = ro.cpu().numpy(); nres = np.diff(roh)
s = int(np.nonzero((nres >= 295) & (nres <= 305))[0][0]); x, y = int(roh[s]), int(roh[s + 1])
item = dict(n_xyz=d["n_xyz"][x:y].cpu().numpy(), ca_xyz=d["ca_xyz"][x:y].cpu().numpy(), cb_xyz=d["cb_xyz"][x:y].cpu().numpy(), aa=d["aa"][x:y].cpu().numpy())
qb = ctx.upload(PackedStructures.concat([item]))
allres = np.arange(y - x, dtype=np.uint32)
os.environ["FDGPU_TRACE"] = "1"
for rep in range(4):
    for index in (ix, None):
        t0 = time.perf_counter(); qm = make_query_map(ctx, qb, allres, None, index, float(S)); t1 = time.perf_counter()
        print("rep %d index=%s: %.2f ms, %d entries" % (rep, index is not None, (t1 - t0) * 1e3, len(qm.hash)), flush=True)
ctx2 = fd.Context(0)
for rep in range(2):
    t0 = time.perf_counter(); qm = make_query_map(ctx2, ctx2.upload(PackedStructures.concat([item])), allres, None, None, float(S)); t1 = time.perf_counter()
    print("own-stream context rep %d: %.2f ms" % (rep, (t1 - t0) * 1e3), flush=True)
PY
cat gpurun_out/qmap_trace_r3.txt | grep -v amdgpu.ids | tail -50
FD_BENCH_TRACE=1 FD_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-export --build-chunk-blocks 1 > gpurun_out/bench_r3c_g2.json 2> gpurun_out/bench_r3c_g2.err; grep "^\[bench\|^\[querybench\|Error\|error" gpurun_out/bench_r3c_g2.err | tail -40
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_r3c_g2.json").read().strip().splitlines()[-1]); q = d["query"]
    print("g2 build", round(d["value"]), "| query", q.get("error") or {k: (round(q[k]["value"]) if isinstance(q.get(k), dict) and "value" in q[k] else None) for k in ("batched_with_matching", "batched_with_matching_128", "batched", "single", "with_matching", "replicas")}, q.get("exchange"))
except Exception as e:
    print("g2 failed", repr(e))
PY
