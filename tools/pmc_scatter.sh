#!/bin/bash
# PMC counters for the radix scatter / hist kernels (own runs, no tracing domains)
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdpmc; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC_EA0?_[A-Z0-9_]+|SQ_[A-Z_]*LDS[A-Z_]*|TCP_[A-Z_]*STALL[A-Z_]*|TCC_[A-Z_]*STALL[A-Z_]*|SQ_INSTS_VMEM[A-Z_]*|TCC_REQ[A-Z_]*|TCC_HIT[A-Z_]*|TCC_MISS[A-Z_]*|TCC_WRITE[A-Z_]*|TCC_READ[A-Z_]*)\b" | sort -u | tr '\n' ' ' > $OUT/pmc_available.txt
CMD="python $REPO/bench.py --structures 16384 --steps 1 --warmup 0 --no-query --no-cpu-baseline"
INC='--kernel-include-regex k_rs_.*'
run() { name=$1; shift; rocprofv3 --output-format csv $INC --pmc "$@" -d $RAW/$name -o $name -- $CMD > $OUT/pmc_$name.log 2>&1; }
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run tccw TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run tccr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcch TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cd $REPO
python - <<'PY' > $OUT/pmc_scatter_summary.txt 2>&1
import csv, glob, collections
for d in ("lds","tccw","tccr","tcch"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for f in glob.glob(f"/tmp/fdpmc/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", d)
    for k,v in acc.items(): print(k, {a: f"{b:.4g}" for a,b in v.items()})
PY
cat $OUT/pmc_available.txt | head -c 3000; echo; cat $OUT/pmc_scatter_summary.txt
