#!/bin/bash
# kernel timeline (start, duration, gap) of a few steady-state LM iterations of the N = 5001 bench graph
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lm_timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/large_graph_run.py 5001 3 > $OUT/run.log 2>&1
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, os
d = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'lm_timeline')
f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Grid_Size_X']) for r in rows]
idx = [i for i, s in enumerate(seq) if s[0].startswith('trial_elim')]
i0 = idx[len(idx) // 2]
prev_end = seq[i0 - 1][2]
t0 = seq[i0][1]
for k in seq[i0:i0 + 16]:
    print('%-36s grid %8s  start %8.2f us  dur %6.2f us  gap %5.2f us' % (k[0], k[3], (k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3, (k[1] - prev_end) / 1e3))
    prev_end = k[2]
PY
tail -1 $OUT/run.log; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
