#!/bin/bash
# tools/profile_round5.sh — the rocprofv3 evidence of round 5, one call on the GPU box (summaries land in gpurun_out/, copy to profiles/):
#  1. kernel trace + stats of the DEFAULT bench command (542,000 structures, query + whole-structure legs included)
#  2. FETCH_SIZE / WRITE_SIZE passes of the build at 542,000 (bench.py reads profiles/*pmc_traffic_S542000.json)
#  3. FETCH_SIZE / WRITE_SIZE / TCC request passes of the tiled motif prefilter (k_qt_*) at 542,000, 8 batches of 32 queries
#     (querybench reads profiles/*pmc_query_traffic_S542000.json); the request counters calibrate the FETCH_SIZE correction per kernel
# Every --pmc pass is its own run with no tracing domain.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof5
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
if [ -z "${R5_ONLY_QUERY:-}" ]; then      # R5_ONLY_QUERY=1: part 3 alone (the query prefilter's PMC passes)
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-cli-index"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/r5_trace.log 2>&1
CMDB="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-export --no-query --no-cli-index"
INC='--kernel-include-regex k_.*'
timeout 900 rocprofv3 --output-format csv $INC --pmc FETCH_SIZE -d $RAW/pmc_fetch -o pmc_fetch -- $CMDB > $OUT/r5_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --output-format csv $INC --pmc WRITE_SIZE -d $RAW/pmc_write -o pmc_write -- $CMDB > $OUT/r5_pmc_write.log 2>&1
cd $REPO
python tools/summarize_prof.py $RAW 542000 > $OUT/r5_prof_summary.txt 2>&1
cp $RAW/prof_traffic.json $OUT/r5_prof_traffic.json 2>/dev/null
python - "$RAW" > $OUT/r5_all_kernels.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats: python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-cli-index (542,000 structures, query and whole-structure legs included) ==")
for r in rows[:80]:
    print("%-90s calls=%-7s total_ms=%10.3f avg_us=%11.2f pct=%s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
fi
# ---- 3. query prefilter traffic: 8 batches of 32 full queries (tools/profile_query_host.py: 1 warm-up + 3 timed rounds of 64 queries)
cd /tmp
CMDQ="python $REPO/tools/profile_query_host.py --structures 542000 --reps 3 --no-profile"
QINC='--kernel-include-regex k_qt_.*|k_cq_plan.*|k_pl_.*'
timeout 900 rocprofv3 --output-format csv $QINC --pmc FETCH_SIZE -d $RAW/q_fetch -o q_fetch -- $CMDQ > $OUT/r5_q_fetch.log 2>&1
timeout 900 rocprofv3 --output-format csv $QINC --pmc WRITE_SIZE -d $RAW/q_write -o q_write -- $CMDQ > $OUT/r5_q_write.log 2>&1
timeout 900 rocprofv3 --output-format csv $QINC --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $RAW/q_req -o q_req -- $CMDQ > $OUT/r5_q_req.log 2>&1
cd $REPO
python - "$RAW" > $OUT/r5_q_traffic_summary.txt <<'PY'
import csv, glob, json, sys, collections
raw = sys.argv[1]
N_BATCH = 8
def pmc(d, counter):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(f"{raw}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return acc, cnt
fa, fc = pmc("q_fetch", "FETCH_SIZE"); wa, wc = pmc("q_write", "WRITE_SIZE")
rq, rc = pmc("q_req", "TCC_EA0_RDREQ_sum"); r32, _ = pmc("q_req", "TCC_EA0_RDREQ_32B_sum"); wq, _ = pmc("q_req", "TCC_EA0_WRREQ_sum"); w64, _ = pmc("q_req", "TCC_EA0_WRREQ_64B_sum")
# FETCH_SIZE on gfx950 counts a read request at 64 B (32 B for the _32B kind); a request made for a 16-byte-per-lane access moves 128 B (the microarch
# guide's factor 2).  Per kernel: `wide` = the share of its read bytes that 16-byte-per-lane loads fetch (from the source: posting bytes, range
# table entries and the decoded stream are dwordx4 / dwordx2x2 accesses, key lists, penalties, metadata are 4- or 8-byte accesses) — the
# corrected figure is raw x (1 + wide).
WIDE = {"k_qt_score<false, 14, 1024, 512, 1, false>": 0.9, "k_qt_rows<14, 512, 6144, 512>": 0.5, "k_qt_plan": 0.0, "k_qt_thr": 0.0, "k_qt_sort": 0.0,
        "k_cq_plan": 0.0, "k_pl_lookup": 0.0}
out = {}
print("== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_*REQ (separate passes): tools/profile_query_host.py --structures 542000 --reps 3 --no-profile = %d batches of 32 full motif queries ==" % N_BATCH)
for k in sorted(set(fa) | set(wa)):
    n = max(fc.get(k, 0), wc.get(k, 0), 1)
    raw_f = fa.get(k, 0.0) * 1024 / max(fc.get(k, 1), 1)
    wide = WIDE.get(k, 0.0)
    out[k] = {"fetch_bytes_per_launch": raw_f, "fetch_correction": 1.0 + wide, "wide_read_share": wide,
              "write_bytes_per_launch": wa.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1), "launches_per_batch": n / N_BATCH, "launches_profiled": n,
              "read_requests_per_launch": rq.get(k, 0.0) / max(rc.get(k, 1), 1), "read_requests_32B_per_launch": r32.get(k, 0.0) / max(rc.get(k, 1), 1),
              "write_requests_per_launch": wq.get(k, 0.0) / max(rc.get(k, 1), 1), "write_requests_64B_per_launch": w64.get(k, 0.0) / max(rc.get(k, 1), 1)}
    print("%-48s launches/batch %5.2f fetch/launch raw %.4g B x %.2f, write/launch %.4g B, read requests %.4g (32 B: %.4g), write requests %.4g (64 B: %.4g)" %
          (k[:48], n / N_BATCH, raw_f, 1 + wide, out[k]["write_bytes_per_launch"], out[k]["read_requests_per_launch"], out[k]["read_requests_32B_per_launch"],
           out[k]["write_requests_per_launch"], out[k]["write_requests_64B_per_launch"]))
lo = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_per_batch"] for v in out.values())
mid = sum((v["fetch_bytes_per_launch"] * v["fetch_correction"] + v["write_bytes_per_launch"]) * v["launches_per_batch"] for v in out.values())
hi = sum((2 * v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_per_batch"] for v in out.values())
print("HBM bytes per batch of 32 queries: raw %.4g, calibrated per kernel %.4g, every read doubled %.4g" % (lo, mid, hi))
json.dump({"structures": 542000, "batches": N_BATCH, "kernels": out, "bytes_per_batch": {"raw": lo, "calibrated": mid, "all_reads_doubled": hi},
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ,WRREQ (separate passes) over tools/profile_query_host.py --structures 542000 --reps 3 "
                   "--no-profile; fetch_correction = 1 + share of the kernel's read bytes fetched by 16-byte-per-lane loads (gfx950 counts those requests at half their size)"},
          open(raw + "/q_traffic.json", "w"), indent=1)
PY
cp $RAW/q_traffic.json $OUT/r5_q_traffic.json 2>/dev/null
[ -z "${R5_ONLY_QUERY:-}" ] && head -60 $OUT/r5_all_kernels.txt; cat $OUT/r5_q_traffic_summary.txt; [ -z "${R5_ONLY_QUERY:-}" ] && tail -30 $OUT/r5_prof_summary.txt
