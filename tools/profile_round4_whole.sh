#!/bin/bash
# tools/profile_round4_whole.sh — rocprofv3 kernel trace of the whole-structure query (tools/profile_whole_query.py) at S structures
S=${1:-542000}
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprofw4
rm -rf $RAW; mkdir -p $OUT $RAW; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python $REPO/tools/profile_whole_query.py --structures $S > $OUT/r4_whole_trace.log 2>&1
cd $REPO
python - "$RAW" $S > $OUT/r4_whole_kernels_S$S.txt 2>&1 <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats: tools/profile_whole_query.py --structures %s (4 whole-structure queries; the index build's kernels left out) ==" % sys.argv[2])
skip = ("at::native", "rocclr", "elementwise", "k_mg_", "k_pair_", "k_rs_scatter", "k_rs_hist", "k_rs_scan", "k_rs_seg", "k_enc_", "k_frames", "k_scan", "k_set_u64", "k_hash_ok", "k_selfcheck")
for r in [x for x in rows if not any(k in x["Name"] for k in skip)][:40]:
    print("%-90s calls=%-6s total_ms=%9.3f avg_us=%10.2f" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
grep "query map\|kernel stages" $OUT/r4_whole_trace.log
head -36 $OUT/r4_whole_kernels_S$S.txt
