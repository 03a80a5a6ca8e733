#!/bin/bash
# tools/pmc_query_kernels.sh — SQ counters (separate --pmc passes, no tracing domains) of the three largest kernels of the batched full query
# (k_match_pairs, k_qt_score, k_rs_slots) at 542,000 structures, batches of 128 through fdgpu_query_batch -> gpurun_out/r5_pmc_query_kernels.txt
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdpmc_qk; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp; cd /tmp
S=${1:-542000}
CMD="python $REPO/tools/profile_query_host.py --structures $S --queries 128 --chunk 128 --reps 3 --no-profile --fused"
REGEX='k_match_pairs.*|k_rs_slots.*|k_qt_score<false, 14.*'
run() { name=$1; shift; timeout 240 rocprofv3 --output-format csv --kernel-include-regex "$REGEX" --pmc "$@" -d $RAW/$name -o $name -- $CMD > $OUT/r5_pmc_qk_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT
cd $REPO
python - "$RAW" <<'PY' > $OUT/r5_pmc_query_kernels.txt 2>&1
import csv, glob, collections, sys
print("== rocprofv3 --pmc (three passes): tools/profile_query_host.py --structures 542000 --queries 128 --chunk 128 --reps 3 --fused: counters summed over 4 launches per kernel ==")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("sq", "sq2", "sq3"):
    for f in glob.glob(f"{sys.argv[1]}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("void ", "").split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k)
    for a, b in sorted(v.items()):
        print("    %-24s %.4g" % (a, b))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("    -> waiting %.0f %% of wave cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES), waiting on LDS %.0f %%, VALU share of issue-active cycles %.0f %%, VALU per wave %.0f" % (
            100 * v["SQ_WAIT_ANY"] / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_ACTIVE_INST_ANY"], 1), v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1)))
PY
cat $OUT/r5_pmc_query_kernels.txt
