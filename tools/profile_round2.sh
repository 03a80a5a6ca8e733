#!/bin/bash
# tools/profile_round2.sh — the rocprofv3 evidence of round 2, one call on the GPU box (summaries land in gpurun_out/, copy to profiles/):
#  1. kernel trace + stats of the DEFAULT bench command (542,000 structures, query leg included): every kernel, not only k_*
#  2. FETCH_SIZE / WRITE_SIZE passes of the build at 542,000 (bench.py reads profiles/*pmc_traffic_S542000.json)
#  3. SQ / TCC / TCP counter sets of k_rs_scatter4 and k_pair_emit2 at 67,750 structures (one build call)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof2
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/r2_trace.log 2>&1
CMDB="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-export --no-query"
INC='--kernel-include-regex k_.*'
timeout 900 rocprofv3 --output-format csv $INC --pmc FETCH_SIZE -d $RAW/pmc_fetch -o pmc_fetch -- $CMDB > $OUT/r2_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --output-format csv $INC --pmc WRITE_SIZE -d $RAW/pmc_write -o pmc_write -- $CMDB > $OUT/r2_pmc_write.log 2>&1
cd $REPO
python tools/summarize_prof.py $RAW 542000 > $OUT/r2_prof_summary.txt 2>&1
cp $RAW/prof_traffic.json $OUT/r2_prof_traffic.json 2>/dev/null
python - "$RAW" > $OUT/r2_all_kernels.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats: python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export (542,000 structures, query leg included) ==")
for r in rows[:60]:
    print("%-90s calls=%-7s total_ms=%10.3f avg_us=%11.2f pct=%s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
bash tools/pmc_scatter2.sh 'k_rs_scatter4.*' r2_scatter4 67750 > /dev/null 2>&1
bash tools/pmc_scatter2.sh 'k_pair_emit2.*' r2_emit2 67750 > /dev/null 2>&1
head -50 $OUT/r2_all_kernels.txt
