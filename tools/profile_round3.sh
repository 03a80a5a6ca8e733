#!/bin/bash
# tools/profile_round3.sh — the rocprofv3 evidence of round 3, one call on the GPU box (summaries land in gpurun_out/, copy to profiles/):
#  1. kernel trace + stats of the DEFAULT bench command (542,000 structures, query leg included)
#  2. FETCH_SIZE / WRITE_SIZE passes of the build at 542,000 (bench.py reads profiles/*pmc_traffic_S542000.json)
#  3. FETCH_SIZE / WRITE_SIZE passes of the batched motif query's prefilter kernels at 542,000 (querybench reads profiles/*pmc_query_traffic_S542000.json)
#  4. SQ / LDS / TCC counter sets of the MSD pair kernel and the segmented scatter at 203,250 structures (one build call)
# Every --pmc pass is its own run with no tracing domain.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof3
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-cli-index"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/r3_trace.log 2>&1
CMDB="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-export --no-query --no-cli-index"
INC='--kernel-include-regex k_.*'
timeout 900 rocprofv3 --output-format csv $INC --pmc FETCH_SIZE -d $RAW/pmc_fetch -o pmc_fetch -- $CMDB > $OUT/r3_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --output-format csv $INC --pmc WRITE_SIZE -d $RAW/pmc_write -o pmc_write -- $CMDB > $OUT/r3_pmc_write.log 2>&1
cd $REPO
python tools/summarize_prof.py $RAW 542000 > $OUT/r3_prof_summary.txt 2>&1
cp $RAW/prof_traffic.json $OUT/r3_prof_traffic.json 2>/dev/null
python - "$RAW" > $OUT/r3_all_kernels.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --kernel-trace --stats: python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-cli-index (542,000 structures, query leg included) ==")
for r in rows[:70]:
    print("%-90s calls=%-7s total_ms=%10.3f avg_us=%11.2f pct=%s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
# ---- 3. query prefilter traffic: 8 batches of 32 full queries (tools/profile_query_host.py: 1 warm-up + 3 timed rounds of 64 queries)
cd /tmp
CMDQ="python $REPO/tools/profile_query_host.py --structures 542000 --reps 3 --no-profile"
QINC='--kernel-include-regex k_cq_.*|k_topn_.*|k_scan_.*|k_pl_.*'
timeout 900 rocprofv3 --output-format csv $QINC --pmc FETCH_SIZE -d $RAW/q_fetch -o q_fetch -- $CMDQ > $OUT/r3_q_fetch.log 2>&1
timeout 900 rocprofv3 --output-format csv $QINC --pmc WRITE_SIZE -d $RAW/q_write -o q_write -- $CMDQ > $OUT/r3_q_write.log 2>&1
cd $REPO
python - "$RAW" > $OUT/r3_q_traffic_summary.txt <<'PY'
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
out = {}
print("== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes): tools/profile_query_host.py --structures 542000 --reps 3 --no-profile = %d batches of 32 full motif queries ==" % N_BATCH)
for k in sorted(set(fa) | set(wa)):
    n = max(fc.get(k, 0), wc.get(k, 0), 1)
    out[k] = {"fetch_bytes_per_launch": fa.get(k, 0.0) * 1024 / max(fc.get(k, 1), 1), "write_bytes_per_launch": wa.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1),
              "launches_per_batch": n / N_BATCH, "launches_profiled": n}
    print("%-40s launches/batch %6.2f fetch/launch %.4g B write/launch %.4g B" % (k[:40], n / N_BATCH, out[k]["fetch_bytes_per_launch"], out[k]["write_bytes_per_launch"]))
tot = sum((2 * v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_per_batch"] for v in out.values())
print("HBM bytes per batch of 32 queries (2 x FETCH_SIZE + WRITE_SIZE, all prefilter kernels): %.4g" % tot)
json.dump({"structures": 542000, "batches": N_BATCH, "kernels": out}, open(raw + "/q_traffic.json", "w"), indent=1)
PY
cp $RAW/q_traffic.json $OUT/r3_q_traffic.json 2>/dev/null
# ---- 4. counter sets of the two dominant build kernels, one build call of 203,250 structures
cd /tmp
CMDK="python $REPO/bench.py --structures 203250 --steps 1 --warmup 0 --no-query --no-cpu-baseline --no-export --no-cli-index"
for K in "k_pair_emit2.*:emit_msd" "k_rs_scatter4_seg.*:scatter_seg"; do
  RX=${K%%:*}; TAG=${K##*:}
  run() { name=$1; shift; timeout 300 rocprofv3 --output-format csv --kernel-include-regex "$RX" --pmc "$@" -d $RAW/k_$TAG/$name -o $name -- $CMDK > $OUT/r3_pmc_${TAG}_$name.log 2>&1; }
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH
  run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
  run tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  run mem FETCH_SIZE
  run mem2 WRITE_SIZE
done
cd $REPO
python - "$RAW" > $OUT/r3_pmc_kernels_summary.txt <<'PY'
import csv, glob, collections, sys
print("== rocprofv3 --pmc (one pass per counter set): python bench.py --structures 203250 --steps 1 --warmup 0 --no-query ... = 2 build calls of 203,250 structures (6.64e9 keys each); sums over the launches ==")
for tag in ("emit_msd", "scatter_seg"):
    for d in ("sq", "sq2", "sq3", "tcc", "mem", "mem2"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for f in glob.glob(f"{sys.argv[1]}/k_{tag}/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"][:56]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:56], r["Counter_Name"])] += 1
        for k, v in sorted(acc.items()):
            print(tag, d, "%-56s" % k, {a: f"{b:.4g} (n={n[(k, a)]})" for a, b in v.items()})
PY
head -45 $OUT/r3_all_kernels.txt; cat $OUT/r3_q_traffic_summary.txt; cat $OUT/r3_pmc_kernels_summary.txt | head -30
