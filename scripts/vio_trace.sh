#!/bin/bash
# rocprofv3 kernel trace of the stereo_vio measurement (bench.vio_frames_per_sec through scripts/vio_only.py) + interval analysis of its
# pipelined phase.  Run on the GPU box:  bash scripts/vio_trace.sh r04x      (summary: gpurun_out/vio_trace_<tag>/)
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/vio_trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VIO_STEPS=${VIO_STEPS:-24}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vio_trace_$TAG -o vio -- python3 $GRAFT_REPO_ROOT/scripts/vio_only.py > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-300
T=$(find /tmp/vio_trace_$TAG -name "*kernel_trace.csv" | head -1)
S=$(find /tmp/vio_trace_$TAG -name "*kernel_stats.csv" | head -1)
cp $S $OUT/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/scripts/vio_trace_analyze.py $T $VIO_STEPS > $OUT/intervals.txt 2>&1
cat $OUT/intervals.txt
