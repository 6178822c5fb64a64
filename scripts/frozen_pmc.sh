#!/bin/bash
# Hardware-counter passes over the frozen nets' forward (scripts/frozen_forward_run.py), one rocprofv3 run per counter set
# (kernel trace only next to the counters, as the pool requires):
#   pass 1  SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   -> matrix-core busy share and the shader clock per kernel family
#   pass 2  FETCH_SIZE                                  -> HBM read bytes   (gfx950: doubled for wide coalesced streams, MI355X_MICROARCH.md)
#   pass 3  WRITE_SIZE                                  -> HBM write bytes
# Run on the GPU box:  bash scripts/frozen_pmc.sh r04x   -> gpurun_out/frozen_pmc_<tag>/summary.txt
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/frozen_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/scripts/frozen_forward_run.py"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/fpmc_$TAG/mfma -o f -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fpmc_$TAG/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/fpmc_$TAG/write -o f -- $CMD > $OUT/write.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/frozen_pmc_summary.py /tmp/fpmc_$TAG > $OUT/summary.txt 2>&1
cp /tmp/fpmc_$TAG/exec_summary.json $OUT/ 2>/dev/null
for p in mfma fetch write; do f=$(find /tmp/fpmc_$TAG/$p -name '*counter_collection.csv' | head -1); [ -n "$f" ] && gzip -c $f > $OUT/${p}_counter_collection.csv.gz; done
cat $OUT/summary.txt
