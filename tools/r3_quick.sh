#!/bin/bash
# quick check of a build-kernel change: the MSD parity test, the scale tests, and the build-only bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import torch; f,t=torch.cuda.mem_get_info(); print(\"HBM free/total bytes\", f, t)"
python -m pytest tests/test_gpu_parity.py::test_msd_build_equals_structure_major_build tests/test_gpu_scale.py tests/test_gpu_fullsize.py::test_swissprot_scale_542000_index_and_planted_motifs -m gpu -x -q --durations=5 > gpurun_out/t_quick.log 2>&1; tail -5 gpurun_out/t_quick.log
python bench.py --no-query --no-cpu-baseline --no-export --no-cli-index --steps 3 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
print("build", round(d["value"]), round(d["ms_per_step"], 1), d["config"].get("call_plan"), {k: round(v, 1) for k, v in d["roofline"]["stages_ms"].items()})
PY
python tools/profile_whole_query.py --structures 203250 > gpurun_out/whole_q.log 2>&1; grep -v "amdgpu.ids" gpurun_out/whole_q.log | tail -40
