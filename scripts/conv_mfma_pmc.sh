#!/bin/bash
# MFMA utilisation of conv3x3_mfma_kernel from hardware counters (own PMC pass, kernel trace only).
# Run on the GPU box:  bash scripts/conv_mfma_pmc.sh   -> gpurun_out/conv_pmc/summary.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv \
    -d $OUT/pmc -o conv -- python3 $GRAFT_REPO_ROOT/scripts/conv_bench.py > $OUT/pmc.log 2>&1
tail -3 $OUT/pmc.log
python3 $GRAFT_REPO_ROOT/scripts/conv_mfma_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the channels-last bf16 kernel of the stereo net (islam_conv_nhwc_bf16) on its own shapes
OUT2=$GRAFT_REPO_ROOT/gpurun_out/conv_nhwc_pmc
mkdir -p $OUT2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv \
    -d $OUT2/pmc -o conv -- python3 $GRAFT_REPO_ROOT/scripts/conv_nhwc_bench.py > $OUT2/pmc.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/conv_mfma_pmc.py $OUT2 > $OUT2/summary.txt 2>&1
cat $OUT2/summary.txt
