#!/bin/bash
# Kernels of the main chain's queue in the pipelined steady state of scripts/vio_only.py (kernel trace): launches and time per step
cd /tmp && export TMPDIR=/tmp
export VIO_STEPS=${VIO_STEPS:-16}
rocprofv3 --kernel-trace --output-format csv -d /tmp/mc_trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/vio_only.py > /tmp/mc_run.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/mc_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the pipelined phase = the densest stretch: take the window of the last third of small_lm launches before the sequential phase starts
lm = [i for i, r in enumerate(rows) if 'small_lm_kernel' in r['Kernel_Name']]
steps = 16
seg = lm[3 * steps // 2: 3 * steps // 2 + steps]           # one pipelined run of `steps` steps (three runs come first)
lo, hi = seg[0], seg[-1]
win = rows[lo:hi]
nsteps = len(seg) - 1
qcount = collections.Counter(r['Queue_Id'] for r in win)
mainq = rows[lm[0]]['Queue_Id']                 # the queue small_lm_kernel runs on (the replay's stereo branch has MORE launches since the pose head went to ~100)
print('window: %d launches over %d steps; queues %s; main chain = queue %s' % (len(win), nsteps, dict(qcount), mainq))
agg = collections.defaultdict(lambda: [0, 0])
third = collections.Counter()
for r in win:
    if r['Queue_Id'] != mainq: continue
    n = r['Kernel_Name']
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    n = re.sub(r'<.*', '', n)[:70]
    agg[n][0] += 1; agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if re.search(r'miopen|Miopen|MIOpen|^_ZN2ck|ck::|Cijk_|rocblas|naive_conv|igemm|gridwise', r['Kernel_Name']): third[n] += 1
print('third-party (MIOpen / CK / rocBLAS / Tensile) kernels on the main chain: %d launches %s' % (sum(third.values()), dict(third)))
tot = sum(v[1] for v in agg.values())
print('main-chain kernels: %.2f ms per step in %.0f launches per step' % (tot / nsteps / 1e6, sum(v[0] for v in agg.values()) / nsteps))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%-72s %6.1f /step %8.1f us/step %6.1f us avg' % (n, c / nsteps, t / nsteps / 1e3, t / c / 1e3))
PY
