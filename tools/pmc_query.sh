#!/bin/bash
# tools/pmc_query.sh — SQ / LDS / TCC counter sets (separate --pmc passes, no tracing) of the scoring and selection kernels of the
# batched motif query on the resident 542,000-structure index: k_cq_seg, k_cq_bounds, k_cq_rows_keys, k_topn_*.
REGEX=${1:-k_cq_.*|k_topn_.*}; TAG=${2:-r2_query}; S=${3:-542000}
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdpmc_$TAG; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/tools/profile_query_host.py --structures $S --reps 3 --no-profile"
run() { name=$1; shift; timeout 300 rocprofv3 --output-format csv --kernel-include-regex "$REGEX" --pmc "$@" -d $RAW/$name -o $name -- $CMD > $OUT/pmc_${TAG}_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_WAVES
run tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
cd $REPO
python - "$RAW" <<'PY' > $OUT/pmc_${TAG}_summary.txt 2>&1
import csv, glob, collections, sys
print("== rocprofv3 --pmc (one pass per counter set): tools/profile_query_host.py --structures 542000 --reps 3 (8 batches of 32 full motif queries); sums over all launches ==")
for d in ("sq","sq2","sq3","tcc2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(f"{sys.argv[1]}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", d)
    for k,v in sorted(acc.items()): print("%-48s" % k, {a: f"{b:.4g}" for a,b in v.items()})
PY
cat $OUT/pmc_${TAG}_summary.txt
