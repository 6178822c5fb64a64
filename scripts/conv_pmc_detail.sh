#!/bin/bash
# Where the hand-written convolution kernels' wave cycles go (SQ counters, own PMC pass).
#   bash scripts/conv_pmc_detail.sh            conv3x3_mfma_kernel (scripts/conv_one.py)
#   bash scripts/conv_pmc_detail.sh nhwc       conv_nhwc_kernel    (scripts/conv_nhwc_one.py, SHAPE=Cin,Cout,k,H,W)
WHICH=${1:-mfma}
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_pmc_detail_$WHICH
if [ "$WHICH" = nhwc ]; then PROG=conv_nhwc_one.py; KERN=conv_nhwc_kernel; else PROG=conv_one.py; KERN=conv3x3_mfma_kernel; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $OUT/pmc -o conv -- python3 $GRAFT_REPO_ROOT/scripts/$PROG > $OUT/pmc.log 2>&1
tail -1 $OUT/pmc.log
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/pmc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if '$KERN' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
wc = sum(acc['SQ_WAVE_CYCLES']) / len(acc['SQ_WAVE_CYCLES'])
for k, v in sorted(acc.items()):
    a = sum(v) / len(v)
    print('%-24s %16.0f  %6.1f%% of SQ_WAVE_CYCLES' % (k, a, 100 * a / wc))
PY
