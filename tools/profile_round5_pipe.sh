#!/bin/bash
# tools/profile_round5_pipe.sh — where a batch of 128 full motif queries goes at 542,000 structures, ONE host thread:
#   1. the blocking fdgpu_query_batch loop: per-kernel time per batch (rocprofv3 --kernel-trace), wall per batch, host share = 1 - kernels / wall
#   2. fdgpu_query_batch_submit / _wait, 4 lanes, 6 batches in flight: wall per batch and the GPU's BUSY share of it (union of the kernels' intervals
#      over the timed loop — kernels of different lanes overlap, so their summed durations exceed the busy time)
# -> gpurun_out/r5_pipe_share.txt (copy to profiles/round5_query_host_share_S542000.txt)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof5
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
S=${1:-542000}; REPS=${2:-48}; SHAPE=${3:-4x6}
cd /tmp
timeout 600 python $REPO/tools/query_pipe.py --structures $S --reps $REPS --profile blocking > $OUT/r5_pp_block_plain.log 2>&1
timeout 600 python $REPO/tools/query_pipe.py --structures $S --reps $REPS --profile $SHAPE > $OUT/r5_pp_pipe_plain.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $RAW/block -o t -- python $REPO/tools/query_pipe.py --structures $S --reps $REPS --profile blocking > $OUT/r5_pp_block_traced.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $RAW/pipe -o t -- python $REPO/tools/query_pipe.py --structures $S --reps $REPS --profile $SHAPE > $OUT/r5_pp_pipe_traced.log 2>&1
cd $REPO
python - "$RAW" "$OUT" "$REPS" "$SHAPE" "$S" > $OUT/r5_pipe_share.txt <<'PY'
import csv, glob, re, sys
raw, out, reps, shape, S = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
def line(f):
    m = re.search(r"PROFILE .*", open(f).read())
    return m.group(0) if m else "?"
def load(d):
    rows = []
    for f in glob.glob(raw + "/" + d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]))
    rows.sort()
    return rows
def timed_region(rows):
    """kernels between the two 1 s sleeps: the last cluster that is preceded by a gap > 0.5 s and holds more than a handful of kernels"""
    cuts = [0] + [k for k in range(1, len(rows)) if rows[k][0] - max(r[1] for r in rows[max(0, k - 64):k]) > 400_000_000] + [len(rows)]
    segs = [rows[a:b] for a, b in zip(cuts, cuts[1:]) if b - a > 50]
    return segs[-1] if segs else rows
def union(rows):
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    return busy
print("== ONE host thread, batches of 128 full motif queries, %s structures: tools/profile_round5_pipe.sh ==" % S)
for tag, name in (("block", "blocking fdgpu_query_batch"), ("pipe", "fdgpu_query_batch_submit / _wait, lanes x in flight = " + shape)):
    plain, traced = line(out + "/r5_pp_%s_plain.log" % tag), line(out + "/r5_pp_%s_traced.log" % tag)
    rows = timed_region(load(tag))
    n_b = sum(1 for r in rows if r[2].startswith("k_mp_scan")) or 1
    span = rows[-1][1] - rows[0][0] if rows else 0
    busy = union(rows)
    print("\n-- %s --" % name)
    print("untraced:", plain)
    print("traced  :", traced)
    print("timed loop in the trace: %d kernels, %d batches; first kernel start -> last kernel end %.3f ms = %.3f ms per batch" % (len(rows), n_b, span / 1e6, span / 1e6 / n_b))
    print("GPU busy (union of kernel intervals) %.3f ms = %.3f ms per batch = %.1f %% of the span; summed kernel durations %.3f ms per batch" %
          (busy / 1e6, busy / 1e6 / n_b, 100.0 * busy / max(span, 1), sum(e - s for s, e, _ in rows) / 1e6 / n_b))
    m = re.search(r"\(([\d.]+) ms per batch\)", plain)
    if m:
        wall = float(m.group(1))
        print("untraced wall per batch %.3f ms -> share of the wall NOT covered by kernels = %.1f %% (busy time per batch of the traced run)" % (wall, 100.0 * max(0.0, wall - busy / 1e6 / n_b) / wall))
    agg = {}
    for s, e, k in rows:
        a = agg.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
    print("kernels per batch (duration while running %s):" % ("alone" if tag == "block" else "beside the other lanes' kernels"))
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
        print("  %-64s launches/batch %5.2f  us/batch %8.1f" % (k[:64], n / n_b, t / 1e3 / n_b))
PY
cat $OUT/r5_pipe_share.txt
