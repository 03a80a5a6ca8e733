#!/bin/bash
# tools/profile_round6.sh — the rocprofv3 evidence of round 6, one call on the GPU box (summaries land in gpurun_out/, copy to profiles/):
#  1. kernel trace + stats of the DEFAULT bench command (542,000 structures, query + whole-structure legs included)
#  2. FETCH_SIZE / WRITE_SIZE passes of the build at 542,000 (bench.py reads profiles/*pmc_traffic_S542000.json)
# (the prefilter's PMC passes: tools/pmc_query_traffic.sh 128 / 32; its SQ counters: tools/pmc_round6_qt.sh)
# Every --pmc pass is its own run with no tracing domain.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof6
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-cli-index"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/r6_trace.log 2>&1
if [ -z "${R6_NO_PMC:-}" ]; then
CMDB="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-export --no-query --no-cli-index"
INC='--kernel-include-regex k_.*'
timeout 900 rocprofv3 --output-format csv $INC --pmc FETCH_SIZE -d $RAW/pmc_fetch -o pmc_fetch -- $CMDB > $OUT/r6_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --output-format csv $INC --pmc WRITE_SIZE -d $RAW/pmc_write -o pmc_write -- $CMDB > $OUT/r6_pmc_write.log 2>&1
fi
cd $REPO
python tools/summarize_prof.py $RAW 542000 > $OUT/r6_prof_summary.txt 2>&1
cp $RAW/prof_traffic.json $OUT/r6_prof_traffic.json 2>/dev/null
python - "$RAW" > $OUT/r6_all_kernels.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats: python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-cli-index (542,000 structures, query and whole-structure legs included) ==")
for r in rows[:90]:
    print("%-90s calls=%-7s total_ms=%10.3f avg_us=%11.2f pct=%s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -3 $OUT/r6_trace.log | cut -c1-400
head -40 $OUT/r6_prof_summary.txt
