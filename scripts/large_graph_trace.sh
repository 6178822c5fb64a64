#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/large_graph_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/large_graph_run.py 300007 2 > $OUT/run.log 2>&1
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, os, collections
d = os.environ.get('OUT', os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'large_graph_trace'))
f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    acc[(n[:48], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print('%-50s grid %9s wg %5s  n=%4d  avg %9.1f us  total %9.1f us  %5.1f%%' % (k[0], k[1], k[2], len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
PY
tail -3 $OUT/run.log; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
