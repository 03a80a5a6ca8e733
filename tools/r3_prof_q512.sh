#!/bin/bash
# where a 512-query batch of full motif queries spends its time: wall per stage (one host thread) and the kernels' own time (rocprofv3)
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprofq; rm -rf $RAW; mkdir -p $OUT $RAW; export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/tools/profile_query_host.py --structures 542000 --queries 512 --chunk 512 --reps 4 --no-profile"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/q512_trace.log 2>&1
grep "full batched" $OUT/q512_trace.log
cd $REPO
python - "$RAW" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
own = [r for r in rows if r["Name"].replace("void ", "").startswith(("k_cq", "k_topn", "k_rs_", "k_match", "k_pl_", "k_superpose", "k_metrics", "k_pair_features", "k_hash_features", "k_scan", "k_gather", "k_qm"))]
own.sort(key=lambda r: -float(r["TotalDurationNs"]))
# 5 batches of 512 ran (1 warm-up + 4 timed) besides the index build's kernels
tot = 0.0
for r in own[:22]:
    per = float(r["TotalDurationNs"]) / 1e6 / 5
    tot += per
    print("%-60s calls=%-6s per 512-batch %.3f ms" % (r["Name"].replace("void ", "")[:60], r["Calls"], per))
print("sum of the listed kernels per 512-query batch: %.3f ms = %.1f us per query" % (tot, tot * 1e3 / 512))
PY
