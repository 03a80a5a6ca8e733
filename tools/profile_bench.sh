#!/bin/bash
# tools/profile_bench.sh [structures] — rocprofv3 evidence for bench.py on the GPU box.
# Writes summaries under gpurun_out/prof_*; copy what should be judged into profiles/.
# Counters are collected in their own runs (no tracing domains together with --pmc).
set -u
S=${1:-16384}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --structures $S --steps 2 --warmup 1 --no-query --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- $CMD > $OUT/prof_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $OUT/prof_pmc_sq -o pmc_sq -- $CMD > $OUT/prof_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o pmc_fetch -- $CMD > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o pmc_write -- $CMD > $OUT/prof_pmc_write.log 2>&1
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/prof_summary.txt 2>&1
tail -60 $OUT/prof_summary.txt
