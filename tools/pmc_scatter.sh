#!/bin/bash
# PMC counters for the radix scatter / hist kernels (own runs, no tracing domains)
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdpmc; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/bench.py --structures 16384 --steps 1 --warmup 0 --no-query --no-cpu-baseline"
INC='--kernel-include-regex k_rs_scatter.*'
run() { name=$1; shift; timeout 180 rocprofv3 --output-format csv $INC --pmc "$@" -d $RAW/$name -o $name -- $CMD > $OUT/pmc_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS
run tcc1 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum
run tcc2 TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_WRITEBACK_sum TCC_WRITE_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $REPO
python - <<'PY' > $OUT/pmc_scatter_summary2.txt 2>&1
import csv, glob, collections
for d in ("sq","sq2","tcc1","tcc2","tcp","grbm"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"/tmp/fdpmc/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", d)
    for k,v in acc.items(): print(k, {a: f"{b:.4g}" for a,b in v.items()})
PY
cat $OUT/pmc_scatter_summary2.txt
