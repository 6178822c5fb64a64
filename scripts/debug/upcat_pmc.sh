#!/bin/bash
# counter passes over scripts/upcat_bench.py (one rocprofv3 run per counter set; kernel trace only beside the counters)
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export UPCAT_ITERS=2
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/upcat_$tag -o f -- python3 $GRAFT_REPO_ROOT/scripts/upcat_bench.py > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob('/tmp/upcat_%s/**/*counter_collection.csv' % tag, recursive=True)
if not f:
    print(tag, 'no counters'); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:60]
    if 'upsample_cat' not in k and 'resize_bilinear' not in k and 'vectorized_elementwise' not in k and 'Fill' not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id'], k) not in seen:
        seen.add((r['Dispatch_Id'], k)); n[k] += 1
for k in agg:
    print(k, 'launches', n[k], {c: v / n[k] for c, v in agg[k].items()})
PY
done > $OUT/upcat_pmc.txt 2>&1
cat $OUT/upcat_pmc.txt
