#!/bin/bash
# tools/profile_round4_host.sh — host share of a batch of 128 full motif queries driven by ONE host thread at 542,000 structures:
#   1. wall time per stage of the loop (tools/profile_query_host.py --chunk 128), FDGPU_TRACE=1 stage split of the library, cProfile of the Python side
#   2. rocprofv3 --kernel-trace --stats of the same loop: kernel time per batch; host share = 1 - kernel time / wall time
# -> gpurun_out/r4_host_share.txt (copy to profiles/round4_query_host_share_S542000.txt)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof4h
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
S=${1:-542000}; REPS=${2:-20}; FUSED=${3:-}      # third argument --fused: one fdgpu_query_batch call per batch
cd /tmp
timeout 900 python $REPO/tools/profile_query_host.py --structures $S --queries 128 --chunk 128 --reps $REPS $FUSED > $OUT/r4_host_wall.txt 2> $OUT/r4_host_wall.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python $REPO/tools/profile_query_host.py --structures $S --queries 128 --chunk 128 --reps $REPS --no-profile $FUSED > $OUT/r4_host_trace.log 2>&1
cd $REPO
python - "$RAW" "$OUT" "$REPS" > $OUT/r4_host_share.txt <<'PY'
import csv, glob, re, sys
raw, out, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
wall = open(out + "/r4_host_wall.txt").read()
traced = open(out + "/r4_host_trace.log").read()
m = re.search(r"full batched: (\d+) q/s; per 128-query batch: (.*)", wall)
mt = re.search(r"full batched: (\d+) q/s; per 128-query batch: (.*)", traced)
print("== one host thread, batches of 128 full motif queries, 542,000 structures: tools/profile_round4_host.sh ==")
print("untraced run :", m.group(0) if m else "?")
print("traced run   :", mt.group(0) if mt else "?")
rows = []
for f in glob.glob(raw + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
QUERY = re.compile(r"k_qt_|k_qd_|k_cq_|k_pl_|k_qm_|k_match|k_rs_(slots|count|scan\(|scatter\(|records)|k_superpose|k_metrics|k_topn|k_pair_features|k_hash_features|k_vote|k_found|k_lms")
batches = reps + 1          # one warm-up round + reps timed rounds of one 128-query batch each
tot = 0.0
lines = []
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].replace("void ", "")
    calls = int(r["Calls"])
    # kernels of the query loop run once (or a small multiple) per batch; the index build's kernels run once per build call
    if calls % batches or not QUERY.search(name):
        continue
    per = float(r["TotalDurationNs"]) / batches / 1e3
    tot += per
    lines.append("  %-70s launches/batch %3d  us/batch %8.1f" % (name.split("(")[0][:70], calls // batches, per))
print("kernels of the query loop, per batch of 128 (rocprofv3 --kernel-trace --stats):")
print("\n".join(lines[:40]))
print("kernel time per batch: %.3f ms" % (tot / 1e3))
if m:
    qps = float(m.group(1)); wall_ms = 128.0 / qps * 1e3
    print("wall per batch (untraced): %.3f ms -> host share (wall - kernels) / wall = %.1f %%" % (wall_ms, 100.0 * (wall_ms - tot / 1e3) / wall_ms))
print()
print("-- library stage split (FDGPU_TRACE=1) and cProfile of the Python side, same loop --")
print(open(out + "/r4_host_wall.err").read()[-6000:])
print(wall[-7000:])
PY
cat $OUT/r4_host_share.txt | head -70
