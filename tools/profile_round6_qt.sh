#!/bin/bash
# tools/profile_round6_qt.sh — the motif prefilter's pass A in its three forms at S structures, batches of 128 full motif queries, one host thread:
#   FDGPU_QT32=0  the 64-bit kernel of rounds 4-5 (k_qt_plan + k_qt_score<A>: first-touch lists, one workgroup per CU)
#   FDGPU_QT32=14 k_qt_layout + k_qt_score32<14, 512>  (64 KB of accumulators: two workgroups per CU)
#   FDGPU_QT32=15 k_qt_layout + k_qt_score32<15, 1024> (tiles of 32,768 structures, one workgroup per CU)
# per form: rocprofv3 --kernel-trace of the blocking loop -> per-kernel time per batch; the untraced blocking and 6x10 pipelined queries/s; the
# workgroup phase clocks (FDGPU_QT_DBG=1).  -> gpurun_out/r6_qt_modes.txt (copy to profiles/round6_qt_modes_S542000.txt)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof6
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
S=${1:-542000}; REPS=${2:-24}; MODES=${3:-"0 14 15"}
cd /tmp
for M in $MODES; do
  export FDGPU_QT32=$M
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $RAW/m$M -o t -- python $REPO/tools/query_pipe.py --structures $S --reps $REPS --profile blocking > $OUT/r6_qt_m${M}_traced.log 2>&1
  timeout 900 python $REPO/tools/query_pipe.py --structures $S --reps 48 --profile 6x10 > $OUT/r6_qt_m${M}_pipe.log 2>&1
  FDGPU_QT_DBG=1 timeout 900 python $REPO/tools/query_pipe.py --structures $S --reps 2 --profile blocking > $OUT/r6_qt_m${M}_dbg.log 2>&1
done
unset FDGPU_QT32
cd $REPO
python - "$RAW" "$OUT" "$S" $MODES > $OUT/r6_qt_modes.txt <<'PY'
import csv, glob, re, sys
raw, out, S, modes = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
def line(f, pat=r"PROFILE .*"):
    try:
        m = re.findall(pat, open(f).read())
    except OSError:
        return "?"
    return m[-1] if m else "?"
def load(d):
    rows = []
    for f in glob.glob(raw + "/" + d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]))
    rows.sort()
    return rows
def timed_region(rows):
    cuts = [0] + [k for k in range(1, len(rows)) if rows[k][0] - max(r[1] for r in rows[max(0, k - 64):k]) > 400_000_000] + [len(rows)]
    segs = [rows[a:b] for a, b in zip(cuts, cuts[1:]) if b - a > 50]
    return segs[-1] if segs else rows
print("== motif prefilter pass A, three forms; %s structures, batches of 128 full motif queries (tools/profile_round6_qt.sh) ==" % S)
for m in modes:
    rows = timed_region(load("m" + m))
    n_b = sum(1 for r in rows if r[2].startswith("k_mp_scan")) or 1
    print("\n-- FDGPU_QT32=%s --" % m)
    print("blocking, traced :", line(out + "/r6_qt_m%s_traced.log" % m))
    print("6 lanes x 10     :", line(out + "/r6_qt_m%s_pipe.log" % m))
    print("phase clocks     :", line(out + "/r6_qt_m%s_dbg.log" % m, r"\[qt(?:32)?\] .*"))
    agg = {}
    for s, e, k in rows:
        a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print("kernels per batch (%d batches in the timed loop): %.1f us in all" % (n_b, tot / 1e3 / n_b))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print("  %-64s %5.1f launches  %8.1f us  %5.1f %%" % (k[:64], v[0] / n_b, v[1] / 1e3 / n_b, 100.0 * v[1] / tot))
PY
cat $OUT/r6_qt_modes.txt
