#!/bin/bash
# hardware counters of the pose head's kernels (one forward + backward per pass; separate --pmc passes, no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pose_head_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  SKIP_TORCH=1 POSE_BENCH_REPS=2 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/pose_head_bench.py 2 > $OUT/run$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/scripts/debug/pose_head_pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +512k -delete
tail -60 $OUT/summary.txt
