#!/bin/bash
# kernel trace + a few SQ counters of the whole-structure query tool (203,250 structures)
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprofw
rm -rf $RAW; mkdir -p $OUT $RAW; export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/tools/profile_whole_query.py --structures 203250"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/pw_trace.log 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-include-regex 'k_match_pairs.*' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $RAW/pmc1 -o pmc1 -- $CMD > $OUT/pw_pmc1.log 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-include-regex 'k_match_pairs.*' --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_FLAT GRBM_GUI_ACTIVE -d $RAW/pmc2 -o pmc2 -- $CMD > $OUT/pw_pmc2.log 2>&1
cd $REPO
python - "$RAW" > $OUT/pw_summary.txt 2>&1 <<'PY'
import csv, glob, sys, collections
raw = sys.argv[1]
rows = []
for f in glob.glob(raw + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
own = ("k_match_pairs", "k_cq_", "k_vote_rows", "k_qm_", "k_found_", "k_topn", "k_pl_", "k_pair_features", "k_gather_xyz", "k_superpose", "k_metrics")
for r in [x for x in rows if any(k in x["Name"] for k in own)] + rows[:10]:
    print("%-80s calls=%-6s total_ms=%9.3f avg_us=%10.2f" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
# per-dispatch durations of the pair scan
for f in glob.glob(raw + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_match_pairs" in r["Kernel_Name"]:
            print("dispatch", r["Kernel_Name"][:40], "grid", r.get("Grid_Size"), "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for d in ("pmc1", "pmc2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{raw}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Dispatch_Id"], r["Kernel_Name"][:30], r.get("Grid_Size"))].append((r["Counter_Name"], float(r["Counter_Value"])))
    for k, v in sorted(acc.items(), key=lambda kv: int(kv[0][0])):
        print(d, k, {n: "%.4g" % x for n, x in v})
PY
tail -60 $OUT/pw_summary.txt
