#!/bin/bash
# Round profile: kernel-trace stats + two separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the bench command.
# Run on the GPU box:  bash scripts/gpu_profile.sh r01      (TRACE_ONLY=1: skip the PMC passes)
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ISLAM_BENCH_LARGE_N=0      # the headline loop only: the N = 300 007 point would put its own (larger) grids into the per-kernel averages
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-frontend"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
if [ -z "$TRACE_ONLY" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
fi
find $OUT -name "*.csv" | head -20
python3 $GRAFT_REPO_ROOT/scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
