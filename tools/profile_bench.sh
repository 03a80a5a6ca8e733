#!/bin/bash
# tools/profile_bench.sh [structures] — rocprofv3 evidence for bench.py on the GPU box.
# kernel-trace stats in one run; FETCH_SIZE and WRITE_SIZE in their own runs (no tracing domains with --pmc).
# Writes small summaries under gpurun_out/ (copy into profiles/ to have them judged).
set -u
S=${1:-67750}
REPO=$(pwd); OUT=$REPO/gpurun_out; RAW=/tmp/fdprof
rm -rf $RAW; mkdir -p $OUT $RAW
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --structures $S --steps 2 --warmup 1 --no-query --no-cpu-baseline"
INC='--kernel-include-regex k_.*'
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/prof_trace.log 2>&1
timeout 300 rocprofv3 --output-format csv $INC --pmc FETCH_SIZE -d $RAW/pmc_fetch -o pmc_fetch -- $CMD > $OUT/prof_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --output-format csv $INC --pmc WRITE_SIZE -d $RAW/pmc_write -o pmc_write -- $CMD > $OUT/prof_pmc_write.log 2>&1
cd $REPO
python tools/summarize_prof.py $RAW $S > $OUT/prof_summary.txt 2>&1
cp $RAW/prof_traffic.json $OUT/ 2>/dev/null
grep -E "^k_|^void k_|==" $OUT/prof_summary.txt | head -60
