"""Launch-by-launch timeline of the last single-GPU and the last world-1 sharded LM run in a rocprofv3 kernel trace of
scripts/sharded_world1_time.py:   python scripts/sharded_world1_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').replace('islam::', '').split('(')[0][:44]


def last_run(marker):
    """kernels from the last control_init_kernel whose run contains `marker`"""
    starts = [i for i, n in enumerate(names) if 'control_init_kernel' in n]
    for a, b in reversed(list(zip(starts, starts[1:] + [len(rows)]))):
        if any(marker in n for n in names[a:b]):
            return rows[a:b]
    return []


for title, marker in (('single GPU', 'trial_elim_kernel'), ('sharded, world 1', 'shard_pack_decide_kernel')):
    run = last_run(marker) if marker != 'trial_elim_kernel' else None
    if run is None:          # the last run WITHOUT the sharded kernels
        starts = [i for i, n in enumerate(names) if 'control_init_kernel' in n]
        for a, b in reversed(list(zip(starts, starts[1:] + [len(rows)]))):
            if not any('shard_' in n for n in names[a:b]):
                run = rows[a:b]
                break
    if not run:
        continue
    t0 = int(run[0]['Start_Timestamp'])
    print('== %s: %d launches, %.1f us from the first start to the last end' % (title, len(run), (int(run[-1]['End_Timestamp']) - t0) / 1e3))
    prev_end = t0
    for r in run[:24]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('  +%7.1f us  gap %5.1f  dur %5.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(r['Kernel_Name'])))
        prev_end = e
