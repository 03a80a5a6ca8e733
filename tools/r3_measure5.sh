#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for nt in 0 1; do
  FDGPU_EMIT_NT=$nt python bench.py --no-query --no-cpu-baseline --no-export --no-cli-index --steps 3 > gpurun_out/ab_nt_$nt.json 2> gpurun_out/ab_nt_$nt.err
done
python - <<'PY'
import json
for g in (0, 1):
    try:
        d = json.loads(open("gpurun_out/ab_nt_%d.json" % g).read().strip().splitlines()[-1])
        print("NT=%d ms_per_step %.1f value %.0f" % (g, d["ms_per_step"], d["value"]), {k: round(v, 1) for k, v in d["roofline"]["stages_ms"].items()})
    except Exception as e:
        print("NT=%d failed: %r" % (g, e)); print(open("gpurun_out/ab_nt_%d.err" % g).read()[-1500:])
PY
python bench.py > gpurun_out/bench_r3d.json 2> gpurun_out/bench_r3d.err; tail -c 300 gpurun_out/bench_r3d.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r3d.json").read().strip().splitlines()[-1]); q = d["query"]
print("build", round(d["value"]), round(d["ms_per_step"], 1), "| query", q.get("error") or {k: (round(q[k]["value"]) if isinstance(q.get(k), dict) and "value" in q[k] else None) for k in ("batched_with_matching", "batched_with_matching_128", "batched_with_matching_mt", "batched", "single", "with_matching")})
print("whole", {k: q["whole_structure"].get(k) for k in ("prefilter_ms", "full_ms")}, "roofline", {k: q["roofline"][k] for k in ("avg_ms", "frac", "traffic")})
c = d["cpu_baseline"]; print("cpu", round(c["value"]), c["hashing_structures_per_s"], c["stages_s"], c["t64"]["stages_s"], c["extrapolated_to_metric_size"]["value"])
print("qcpu", q["cpu_baseline"] and {k: q["cpu_baseline"][k] for k in ("value", "cores")})
PY
