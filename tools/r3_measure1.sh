#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
for g in 3 1; do
  python bench.py --no-query --no-cpu-baseline --no-export --no-cli-index --steps 3 --build-chunk-blocks $g > gpurun_out/ab_chunk_$g.json 2> gpurun_out/ab_chunk_$g.err
done
python - <<'PY'
import json
for g in (3, 1):
    try:
        d = json.loads(open("gpurun_out/ab_chunk_%d.json" % g).read().strip().splitlines()[-1])
        print("blocks/call=%d ms_per_step %.1f value %.0f" % (g, d["ms_per_step"], d["value"]), {k: round(v, 1) for k, v in d["roofline"]["stages_ms"].items()})
    except Exception as e:
        print("blocks/call=%d failed: %r" % (g, e)); print(open("gpurun_out/ab_chunk_%d.err" % g).read()[-1500:])
PY
python tools/profile_query_host.py --structures 542000 --chunk 32,128 > gpurun_out/query_host_profile_r3.txt 2>&1
head -40 gpurun_out/query_host_profile_r3.txt
python tools/profile_whole_query.py --structures 542000 > gpurun_out/whole_query_profile_r3.txt 2>&1
tail -25 gpurun_out/whole_query_profile_r3.txt
