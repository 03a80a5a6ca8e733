#!/bin/bash
# PMC of the pair kernel with and without the MSD buckets (one build call of 67,750 structures each): HBM traffic, L2 write requests,
# instruction mix.  Separate --pmc passes, no tracing domains.
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdpmc_emit_ab; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/bench.py --structures 67750 --steps 1 --warmup 0 --no-query --no-cpu-baseline --no-export"
for m in 1 0; do
  run() { name=$1; shift; FDGPU_MSD=$m timeout 240 rocprofv3 --output-format csv --kernel-include-regex 'k_pair_emit2.*|k_pair_count.*' --pmc "$@" -d $RAW/m$m/$name -o $name -- $CMD > $OUT/pmc_emitab_${m}_$name.log 2>&1; }
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run tcc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_REQ TCC_HIT TCC_MISS
  run sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
  run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
done
cd $REPO
python - "$RAW" <<'PY' > $OUT/pmc_emit_ab_summary.txt 2>&1
import csv, glob, collections, sys
for m in ("m1", "m0"):
    print("======== FDGPU_MSD=%s" % m[1])
    for d in ("fetch", "write", "tcc", "sq", "sq2"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(int)
        for f in glob.glob(f"{sys.argv[1]}/{m}/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            print(d, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
cat $OUT/pmc_emit_ab_summary.txt
