#!/bin/bash
# tools/profile_bench.sh [structures] — rocprofv3 evidence for bench.py on the GPU box.
# Writes summaries under gpurun_out/; copy what should be judged into profiles/.
# Counters are collected in their own runs (no tracing domains together with --pmc).
set -u
S=${1:-16384}
REPO=$(pwd)
OUT=$REPO/gpurun_out
RAW=/tmp/fdprof
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --structures $S --steps 2 --warmup 1 --no-query --no-cpu-baseline"
INC='--kernel-include-regex k_(pair|rs_|enc_|frames|scan|cq_|match|kabsch|posting).*'
rocprofv3 --kernel-trace --stats --output-format csv $INC -d $RAW/trace -o trace -- $CMD > $OUT/prof_trace.log 2>&1
rocprofv3 --output-format csv $INC --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $RAW/pmc_sq -o pmc_sq -- $CMD > $OUT/prof_pmc_sq.log 2>&1
rocprofv3 --output-format csv $INC --pmc FETCH_SIZE -d $RAW/pmc_fetch -o pmc_fetch -- $CMD > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --output-format csv $INC --pmc WRITE_SIZE -d $RAW/pmc_write -o pmc_write -- $CMD > $OUT/prof_pmc_write.log 2>&1
cd $REPO
find $RAW -type f | head -40 > $OUT/prof_files.txt
# keep only the small stats files
mkdir -p $OUT/prof_stats
find $RAW -name '*stats*.csv' -exec cp {} $OUT/prof_stats/ \;
python tools/summarize_prof.py $RAW > $OUT/prof_summary.txt 2>&1
cat $OUT/prof_files.txt | head -20
tail -70 $OUT/prof_summary.txt
