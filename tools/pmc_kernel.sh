#!/bin/bash
# PMC counters for one kernel family (own runs, no tracing domains). usage: tools/pmc_kernel.sh <kernel-regex> <tag> [structures]
REGEX=${1:-k_pair_emit2.*}; TAG=${2:-emit}; S=${3:-16384}
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdpmc_$TAG; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/bench.py --structures $S --steps 1 --warmup 0 --no-query --no-cpu-baseline"
run() { name=$1; shift; timeout 180 rocprofv3 --output-format csv --kernel-include-regex "$REGEX" --pmc "$@" -d $RAW/$name -o $name -- $CMD > $OUT/pmc_${TAG}_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $REPO
python - "$RAW" <<'PY' > $OUT/pmc_${TAG}_summary.txt 2>&1
import csv, glob, collections, sys
for d in ("sq","sq2","sq3","grbm"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{sys.argv[1]}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", d)
    for k,v in acc.items(): print(k, {a: f"{b:.4g}" for a,b in v.items()})
PY
cat $OUT/pmc_${TAG}_summary.txt
