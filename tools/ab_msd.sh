#!/bin/bash
# A/B of the MSD index build on the bench's database (FDGPU_MSD=1: bucketed emit + three segmented passes, with and without the amino-acid
# order of the residues; 0: structure-major stream + four passes): the bench line of each mode, and the sha256 of the index files of a
# 20,000-structure database built every way.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in "1 1" "1 0" "0 1"; do
  set -- $cfg
  tag="$1$2"
  FDGPU_MSD=$1 FDGPU_MSD_PERM=$2 python bench.py --no-query --no-cpu-baseline --no-export --steps 3 > gpurun_out/ab_msd_$tag.json 2> gpurun_out/ab_msd_$tag.err
  FDGPU_MSD=$1 FDGPU_MSD_PERM=$2 python - > gpurun_out/ab_msd_sha_$tag.txt 2>&1 <<'PY'
import hashlib, torch, numpy as np
import folddisco_amd as fd
from folddisco_amd import synth
ctx = fd.Context(0)
ps = synth.to_packed(synth.generate(20000, seed=5, device="cuda"))
ix = fd.FolddiscoIndex.build(ctx, ctx.upload(ps), first_id=7)
v, h, o = ix.export()
print(hashlib.sha256(v.tobytes()).hexdigest()[:16], hashlib.sha256(h.tobytes()).hexdigest()[:16], hashlib.sha256(o.tobytes()).hexdigest()[:16], len(v), len(h))
PY
done
tail -n 1 gpurun_out/ab_msd_sha_*.txt
python - <<'PY'
import json
for m in ("11", "10", "01"):
    try:
        d = json.loads(open("gpurun_out/ab_msd_%s.json" % m).read().strip().splitlines()[-1])
        print("MSD,PERM=%s ms_per_step %.1f value %.0f" % (m, d["ms_per_step"], d["value"]), {k: round(v, 1) for k, v in d["roofline"]["stages_ms"].items()})
    except Exception as e:
        print("MSD,PERM=%s failed: %r" % (m, e)); print(open("gpurun_out/ab_msd_%s.err" % m).read()[-2000:])
PY
