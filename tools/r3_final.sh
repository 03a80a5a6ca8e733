#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/t_r3_final.log 2>&1; tail -10 gpurun_out/t_r3_final.log
python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err; tail -c 300 gpurun_out/bench_r3_final.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r3_final.json").read().strip().splitlines()[-1]); q = d["query"]
print("build", round(d["value"]), round(d["ms_per_step"], 1), {k: round(v, 1) for k, v in d["roofline"]["stages_ms"].items()})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"]["traffic"], d["roofline"]["end_to_end"])
print("export", d["export_inclusive"]["value"], d["export_inclusive"]["ms_per_step"])
print("query", q.get("error") or {k: (round(q[k]["value"]) if isinstance(q.get(k), dict) and "value" in q[k] else None) for k in ("batched_with_matching", "batched_with_matching_128", "batched_with_matching_512", "batched_with_matching_mt", "batched", "single", "with_matching")}, q["batched_with_matching_mt"])
print("whole", {k: q["whole_structure"].get(k) for k in ("prefilter_ms", "full_ms", "stages_ms")}, "qroofline", {k: q["roofline"][k] for k in ("avg_ms", "achieved", "frac", "traffic")})
c = d["cpu_baseline"]; print("cpu", round(c["value"]), c["cores"], c["cpu_model"], c["cpu_budget"], c["hashing_structures_per_s"]["scaling"], c["stages_s"], c["t64"]["value"], c["extrapolated_to_metric_size"]["value"])
print("qcpu", q["cpu_baseline"] and {k: q["cpu_baseline"][k] for k in ("value", "cores")}, q["cpu_baseline"] and q["cpu_baseline"]["t64"]["value"])
print("cli", d["cli_index"])
PY
