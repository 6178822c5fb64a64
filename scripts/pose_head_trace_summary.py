"""Summarise a rocprofv3 kernel trace of scripts/pose_head_bench.py: the launches of ONE direct (non-graph) forward and ONE backward call:
kernel, grid, duration, start relative to the call's first kernel, stream / queue."""
import csv, glob, os, sys
d = sys.argv[1]
f = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
KEYS = ('conv_gemm_kernel', 'conv_bwd_pair_kernel', 'stem0_fwd', 'fc1_fwd', 'fc23_fwd', 'fc23_bwd', 'fc1_wgrad', 'fc1_dgrad')
def short(n):
    n = n.replace('(anonymous namespace)::', '')
    for k in KEYS:
        if k in n:
            i = n.find(k)
            return n[i:i + 30].split('(')[0]
    return n[:40]
seq = [dict(name=short(r['Kernel_Name']), wg=int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), gy=int(r['Grid_Size_Y']),
            us=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, t0=int(r['Start_Timestamp']), t1=int(r['End_Timestamp']), q=r.get('Queue_Id', '?')) for r in rows]
starts_f = [i for i, s in enumerate(seq) if s['name'].startswith('stem0_fwd')]
starts_b = [i for i, s in enumerate(seq) if s['name'].startswith('fc23_bwd')]
def dump(name, i0, verbose):
    print('==== %s' % name)
    tot, i, t0, last_end = 0.0, i0, seq[i0]['t0'], {}
    while i < len(seq):
        k = seq[i]
        if i > i0 and (k['name'].startswith('stem0_fwd') or k['name'].startswith('fc23_bwd')):
            break
        if not any(x in k['name'] for x in KEYS):
            break
        gap = (k['t0'] - last_end[k['q']]) / 1e3 if k['q'] in last_end else 0.0
        if verbose:
            print('%-30s wg %5d x %3d  %7.1f us  start %8.1f  queue %s  gap-on-queue %5.1f' % (k['name'], k['wg'], k['gy'], k['us'], (k['t0'] - t0) / 1e3, k['q'], gap))
        last_end[k['q']] = k['t1']
        tot += k['us']
        i += 1
    print('kernels %.1f us, wall %.1f us, launches %d' % (tot, (max(s['t1'] for s in seq[i0:i]) - t0) / 1e3, i - i0))
v = len(sys.argv) > 2
dump('forward', starts_f[10], v)
dump('backward', starts_b[10], v)
