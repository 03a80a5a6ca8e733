#!/bin/bash
# tools/profile_query_batch.sh — rocprofv3 kernel trace of the BATCHED full query alone (30 x 2 batches of 32 queries on the resident
# 542,000-structure index): per-kernel averages of the query leg without the single-query and whole-structure legs mixed in.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprofqb
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python $REPO/tools/profile_query_host.py --structures 542000 --reps ${REPS:-30} --chunk ${CHUNK:-32} --no-profile > $OUT/r2qb_trace.log 2>&1
cd $REPO
python - "$RAW" > $OUT/r2qb_kernels.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if r["Name"].startswith(("k_", "void k_", "__amd_rocclr"))]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats: tools/profile_query_host.py --structures 542000 --reps 30 (62 batches of 32 full queries; build kernels of the 8 chunks included) ==")
for r in rows[:70]:
    print("%-100s calls=%-7s total_ms=%10.3f avg_us=%11.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
grep "full batched" $OUT/r2qb_trace.log
