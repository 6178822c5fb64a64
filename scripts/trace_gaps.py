"""Per-kernel durations and inter-kernel gaps of the LM loop from a rocprofv3 kernel trace CSV."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = {'bt_downsweep': 'sweep', 'linearize': 'lin', 'build_normal': 'build', 'bt_eliminate': 'elim', 'bt_top': 'top', 'bt_backsub': 'bsub', 'trial_lin': 'trial', 'trial_kernel': 'trial', 'linbuild': 'linb', 'control_begin': 'cbeg', 'msg_kernel': 'msg', 'decide_kernel': 'decide', 'outer_block': 'outer'}
seq = []
for r in rows:
    n = r['Kernel_Name']
    tag = next((v for k, v in names.items() if k in n), None)
    if tag:
        seq.append((tag, int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Grid_Size_X')))
# LM loop only: bench.py runs the loop first, then the event-timed solves; keep kernels up to the last trial kernel
# (a sharded run -- msg / decide kernels present -- keeps the stretch between the first and the last decision)
dec = [i for i, q in enumerate(seq) if q[0] == 'msg']
if dec:
    seq = seq[dec[len(dec) // 2]:dec[-1] + 1]
else:
    last_trial = max(i for i, q in enumerate(seq) if q[0] == 'trial')
    seq = seq[max(0, last_trial - 360):last_trial + 1]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, (t, s, e, g) in enumerate(seq):
    key = t + ('/' + g if t in ('elim', 'bsub') else '')
    dur[key].append(e - s)
    if i > 0:
        gap[seq[i - 1][0] + '->' + t].append(s - seq[i - 1][2])
print('durations (us, median):')
for k, v in dur.items():
    v.sort(); print('  %-14s n=%3d %.1f' % (k, len(v), v[len(v) // 2] / 1e3))
print('gaps (us, median):')
for k, v in gap.items():
    v.sort(); print('  %-14s n=%3d %.1f' % (k, len(v), v[len(v) // 2] / 1e3))
# one LM iteration = from one trial end to the next trial end
ends = [e for (t, s, e, g) in seq if t == ('msg' if dec else 'trial')]
d = sorted(b - a for a, b in zip(ends, ends[1:]))
print('trial-to-trial (us, median): %.1f' % (d[len(d) // 2] / 1e3))
