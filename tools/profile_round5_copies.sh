#!/bin/bash
# tools/profile_round5_copies.sh — the memory copies of one batch of 128 full motif queries (blocking fdgpu_query_batch): rocprofv3 --memory-copy-trace
# -> gpurun_out/r5_copies.txt: per copy of a batch its direction and bytes, in stream order
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof5c
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $RAW/t -o t -- python $REPO/tools/query_pipe.py --structures ${1:-542000} --reps 6 --profile blocking > $OUT/r5_copies.log 2>&1
cd $REPO
python - "$RAW" > $OUT/r5_copies.txt <<'PY'
import csv, glob, sys
raw = sys.argv[1]
cp, kn = [], []
for f in glob.glob(raw + "/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "?")), int(r.get("Bytes", r.get("Size", 0) or 0))))
for f in glob.glob(raw + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:40]))
ev = sorted([(s, e, "COPY %s %d B" % (d, b)) for s, e, d, b in cp] + [(s, e, "kernel " + n) for s, e, n in kn])
# the last batch: everything after the last k_qm_expand_hash... find starts of batches by k_pair_features12
starts = [i for i, x in enumerate(ev) if x[2].startswith("kernel k_pair_features12")]
if len(starts) >= 2:
    a, b = starts[-2], starts[-1]
    t0 = ev[a][0]
    print("== one batch of 128 full motif queries (blocking fdgpu_query_batch), stream order: start us, duration us, what ==")
    for s, e, w in ev[a:b]:
        print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, w))
    print("copies in the batch: %d, bytes %d" % (sum(1 for x in ev[a:b] if x[2].startswith("COPY")), sum(int(x[2].split()[-2]) for x in ev[a:b] if x[2].startswith("COPY"))))
else:
    print("no batch boundaries found", len(ev), len(cp), len(kn))
    import os
    for f in glob.glob(raw + "/t/**/*.csv", recursive=True): print(f, open(f).readline().strip())
PY
cat $OUT/r5_copies.txt | head -90
