#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python bench.py --no-query --no-cpu-baseline --no-export --no-cli-index --steps 3 > gpurun_out/ab_final.json 2> gpurun_out/ab_final.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/ab_final.json").read().strip().splitlines()[-1])
print("build ms_per_step %.1f value %.0f" % (d["ms_per_step"], d["value"]), {k: round(v, 1) for k, v in d["roofline"]["stages_ms"].items()})
PY
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wq -o wq -- python $GRAFT_REPO_ROOT/tools/profile_whole_query.py --structures 203250 > $GRAFT_REPO_ROOT/gpurun_out/whole_query_profile_r3b.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids gpurun_out/whole_query_profile_r3b.txt | tail -28
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/wq/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if r["Name"].replace("void ", "").startswith("k_")]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print("%-70s calls=%-5s total_ms=%9.3f avg_us=%10.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
