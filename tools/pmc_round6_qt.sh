#!/bin/bash
# tools/pmc_round6_qt.sh — SQ / TCP / TCC counters (separate --pmc passes, no tracing domains) of the prefilter kernels of the batched full query
# (k_qt_layout, k_qt_score32, k_qt_rows) at S structures, batches of 128 through fdgpu_query_batch -> gpurun_out/r6_pmc_qt.txt
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdpmc_q6; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp; cd /tmp
S=${1:-542000}
CMD="python $REPO/tools/profile_query_host.py --structures $S --queries 128 --chunk 128 --reps 3 --no-profile --fused"
REGEX=${2:-'k_qt_layout.*|k_qt_score32.*|k_qt_rows.*'}
run() { name=$1; shift; timeout 300 rocprofv3 --output-format csv --kernel-include-regex "$REGEX" --pmc "$@" -d $RAW/$name -o $name -- $CMD > $OUT/r6_pmc_qt_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT
run tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run occ SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM
cd $REPO
python - "$RAW" <<'PY' > $OUT/r6_pmc_qt.txt 2>&1
import csv, glob, collections, sys
print("== rocprofv3 --pmc (separate passes): tools/profile_query_host.py --structures 542000 --queries 128 --chunk 128 --reps 3 --fused: counters summed over the launches of each kernel ==")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k][r["Counter_Name"]] += 1
for k, v in acc.items():
    print(k, "(launches per counter: %d)" % max(nl[k].values()))
    for a, b in sorted(v.items()):
        print("    %-36s %.4g" % (a, b))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("    -> waiting %.0f %% of wave cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES), waiting on LDS %.0f %%, VALU share of issue-active cycles %.0f %%, VALU per wave %.0f" % (
            100 * v["SQ_WAIT_ANY"] / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_ACTIVE_INST_ANY"], 1), v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1)))
PY
cat $OUT/r6_pmc_qt.txt
