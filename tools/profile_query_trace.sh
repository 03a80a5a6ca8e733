#!/bin/bash
# tools/profile_query_trace.sh — rocprofv3 kernel trace of the default bench command restricted to the library's kernels (k_*), so that
# the query leg's kernels (k_cq_*, k_topn_*, k_match_pairs, k_rs_*, k_superpose, k_metrics) are visible next to the build's.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprofq
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv --kernel-include-regex 'k_.*' -d $RAW/trace -o trace -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-export > $OUT/r2q_trace.log 2>&1
cd $REPO
python - "$RAW" > $OUT/r2q_kernels.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats --kernel-include-regex 'k_.*': python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-export (542,000 structures, query leg included) ==")
for r in rows[:80]:
    print("%-100s calls=%-7s total_ms=%10.3f avg_us=%11.2f pct=%s" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -3 $OUT/r2q_trace.log
