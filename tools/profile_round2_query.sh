set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof2
mkdir -p $OUT
timeout 900 python bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
bash tools/profile_query_batch.sh > $OUT/final_qb.log 2>&1
rm -rf $RAW; mkdir -p $RAW
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/r2_trace.log 2>&1
cd $REPO
python - "$RAW" > $OUT/r2_all_kernels.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats: python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export (542,000 structures, query leg included) ==")
for r in rows[:70]:
    print("%-90s calls=%-7s total_ms=%10.3f avg_us=%11.2f pct=%s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -2 $OUT/final_qb.log
