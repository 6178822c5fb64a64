#!/bin/bash
# Kernel trace of the sharded LM loop at world 1 (ISLAM_FORCE_SHARDED=1): per-kernel durations and gaps.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sharded
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ISLAM_FORCE_SHARDED=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-frontend > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/trace_gaps.py $F > $OUT/gaps.txt 2>&1
cat $OUT/gaps.txt
