#!/bin/bash
# When do MIOpen's naive convolution kernels run during scripts/vio_only.py?  (kernel trace; prints their time buckets and neighbours)
cd /tmp && export TMPDIR=/tmp
export VIO_STEPS=${VIO_STEPS:-16}
rocprofv3 --kernel-trace --output-format csv -d /tmp/naive_trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/vio_only.py > /tmp/naive_run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/naive_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp']); t1 = int(rows[-1]['End_Timestamp'])
print('trace spans %.1f s, %d launches' % ((t1 - t0) / 1e9, len(rows)))
buckets = collections.Counter(); dur = collections.Counter()
for i, r in enumerate(rows):
    if 'naive_conv' in r['Kernel_Name']:
        b = (int(r['Start_Timestamp']) - t0) // 10**9
        buckets[b] += 1; dur[b] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('naive_conv launches per second of the run:', dict(sorted(buckets.items())))
print('naive_conv ms per second of the run:', {k: round(v / 1e6, 1) for k, v in sorted(dur.items())})
allb = collections.Counter()
for r in rows: allb[(int(r['Start_Timestamp']) - t0) // 10**9] += 1
print('all launches per second:', dict(sorted(allb.items())))
shown = 0
for i, r in enumerate(rows):
    if 'naive_conv' in r['Kernel_Name'] and shown < 6:
        shown += 1
        print('---', r['Kernel_Name'][:70], 'grid', r.get('Grid_Size'), 'queue', r.get('Queue_Id'), 'dur us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        for q in rows[max(0, i - 2):i]: print('     before:', q['Kernel_Name'][:90])
        for q in rows[i + 1:i + 3]: print('     after: ', q['Kernel_Name'][:90])
PY
tail -1 /tmp/naive_run.log | cut -c1-200
