#!/bin/bash
# per-kernel times of the hand-written pose head (scripts/pose_head_bench.py) under rocprofv3 --kernel-trace; writes gpurun_out/pose_head_trace/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pose_head_trace
rm -rf $OUT; mkdir -p $OUT
SKIP_TORCH=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/pose_head_bench.py 20 > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pose_head_trace_summary.py $OUT v > $OUT/summary.txt 2>&1
tail -80 $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
