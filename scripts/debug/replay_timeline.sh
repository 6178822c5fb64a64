#!/bin/bash
# Frozen-net replays in the pipelined steady state of scripts/vio_only.py (kernel trace): per hardware queue of the captured graph's
# branches -- span of one replay, kernel time, idle gaps between consecutive kernels, and which kernels the gaps follow.
cd /tmp && export TMPDIR=/tmp
export VIO_STEPS=${VIO_STEPS:-16}
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt_trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/vio_only.py > /tmp/rt_run.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/rt_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lm = [i for i, r in enumerate(rows) if 'small_lm_kernel' in r['Kernel_Name']]
steps = 16
seg = lm[3 * steps // 2: 3 * steps // 2 + steps]
win = rows[seg[0]:seg[-1]]
nsteps = len(seg) - 1
mainq = rows[lm[0]]['Queue_Id']
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return re.sub(r'<.*', '', n)[:48]
byq = collections.defaultdict(list)
for r in win:
    byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
print('window %.2f ms per step over %d steps' % ((int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])) / nsteps / 1e6, nsteps))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    if len(ks) < 50: continue
    busy = sum(e - s for s, e, _ in ks)
    gaps = collections.defaultdict(lambda: [0, 0])
    big = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        g = s1 - e0
        if g > 200000: big += g; continue            # between replays / steps
        gaps[n0][0] += 1; gaps[n0][1] += max(g, 0)
    gs = sum(v[1] for v in gaps.values())
    print('\nqueue %s%s: %d launches/step, kernels %.2f ms/step, gaps inside a replay %.2f ms/step (%.1f us per launch), idle between %.2f ms/step'
          % (q, ' (main chain)' if q == mainq else '', len(ks) / nsteps, busy / nsteps / 1e6, gs / nsteps / 1e6, gs / max(1, sum(v[0] for v in gaps.values())) / 1e3, big / nsteps / 1e6))
    for n, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print('   gap after %-50s %6.1f /step %7.1f us/step %5.1f us avg' % (n, c / nsteps, t / nsteps / 1e3, t / c / 1e3))
# the ordered kernel sequence of ONE replay on the busiest replay queue (full names of what is not ours)
rq = max((q for q in byq if q != mainq), key=lambda q: len(byq[q]))
ks = [r for r in win if r['Queue_Id'] == rq]
cut = [i for i in range(1, len(ks)) if int(ks[i]['Start_Timestamp']) - int(ks[i - 1]['End_Timestamp']) > 200000]
segs = [ks[a:b] for a, b in zip([0] + cut, cut + [len(ks)])]
one = max(segs, key=len)
print('\none replay on queue %s: %d launches' % (rq, len(one)))
for i, r in enumerate(one):
    n = r['Kernel_Name']
    ours = not re.search(r'at::native|rocclr|_ZN2ck|ck::|miopen|igemm|naive_conv', n)
    print('%4d %7.1f us  grid %8s  %s' % (i, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size', r.get('Grid_Size_X', '?')), short(n) if ours else n[:230]))
PY
