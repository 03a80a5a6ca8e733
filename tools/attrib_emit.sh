#!/bin/bash
# tools/attrib_emit.sh — where the VALU instructions of the index build's pair kernel go: SQ counters PER LAUNCH of
# k_pair_emit2 / k_pair_count_msd for ONE build call of 67,750 structures (tools/one_build.py; bench.py's counters sum over its legs), in variants:
#   default | FDGPU_MSD=0 (no bucket bookkeeping: structure-major stream) | FDGPU_EXACT=1 (no speculation) | -DFD_EMIT_STUB=1 (the drain without
#   the descriptor: queue, claims, stores only — rebuilt on the box) -> gpurun_out/r6_attrib_emit.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdattrib; rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
S=${1:-67750}
REGEX='k_pair_emit2.*|k_pair_count_msd.*|k_pair_count2.*'
run() { tag=$1; shift
  ( cd /tmp; timeout 300 rocprofv3 --output-format csv --kernel-include-regex "$REGEX" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $RAW/$tag -o $tag -- python $REPO/tools/one_build.py --structures $S > $OUT/r6_attrib_$tag.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $RAW/${tag}_t -o t -- python $REPO/tools/one_build.py --structures $S --calls 3 > $OUT/r6_attrib_${tag}_t.log 2>&1 ); }
run default
FDGPU_MSD=0 run msd0
FDGPU_EXACT=1 run exact
( cd $REPO; FD_KHASH_FLAGS="-fno-slp-vectorize -DFD_EMIT_STUB=1" python -c "from folddisco_amd import build as fb; fb.build(force=True)" > $OUT/r6_attrib_rebuild.log 2>&1 )
run stub1
( cd $REPO; python -c "from folddisco_amd import build as fb; fb.build(force=True)" >> $OUT/r6_attrib_rebuild.log 2>&1 )
python - "$RAW" "$S" > $OUT/r6_attrib_emit.txt <<'PY'
import csv, glob, collections, sys
raw, S = sys.argv[1], int(sys.argv[2])
print("== index build pair kernels, ONE build call of %d structures (tools/attrib_emit.sh): SQ counters per launch; durations from a separate --kernel-trace run (median of calls 2-3) ==" % S)
for tag in ("default", "msd0", "exact", "stub1"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.Counter()
    for f in glob.glob(f"{raw}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": nl[k] += 1
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{raw}/{tag}_t/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if k.startswith("k_pair"): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("\n-- variant %s --" % tag)
    for k, v in acc.items():
        n = max(nl[k], 1)
        d = sorted(dur.get(k, [0.0]))
        print("%-44s launches %d; per launch: VALU %.4g (%.0f per structure), SALU %.4g, LDS %.4g, waves %.4g, VALU per wave %.0f; busy-normalised: ACTIVE_INST_VALU %.4g, WAVE_CYCLES %.4g, WAIT_INST_ANY %.4g; %.2f ms" % (
            k, n, v["SQ_INSTS_VALU"] / n, v["SQ_INSTS_VALU"] / n / S, v["SQ_INSTS_SALU"] / n, v["SQ_INSTS_LDS"] / n, v["SQ_WAVES"] / n, v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1),
            v["SQ_ACTIVE_INST_VALU"] / n, v["SQ_WAVE_CYCLES"] / n, v["SQ_WAIT_INST_ANY"] / n, d[len(d) // 2]))
PY
cat $OUT/r6_attrib_emit.txt
