"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into a small text/JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, '**', pat), recursive=True)
    return r[0] if r else None


summary = {}
f = find('trace', '*kernel_stats.csv')
if f:
    print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        name = r.get('Name', '')[:70]
        print('%-70s calls=%-6s avg_ns=%-10s total_ns=%-12s pct=%s' % (name, r.get('Calls'), r.get('AverageNs'), r.get('TotalDurationNs'), r.get('Percentage')))
    summary['kernel_stats'] = [{k: r.get(k) for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')} for r in rows[:25]]
f = find('trace', '*kernel_trace.csv')
if f:
    # the same kernel runs at several levels / as a gated no-op: split by grid size, and drop the no-op dispatches
    # (run-ahead iterations cancelled by the epoch gate return at once) from the "live" average
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        g = r.get('Grid_Size_X') or r.get('Grid_Size') or ''
        per[(r['Kernel_Name'][:60], g)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    print('== per (kernel, grid) durations from the kernel trace: all dispatches / live dispatches (> 60 % of the max) ==')
    summary['per_grid'] = {}
    for (k, g), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
        live = [x for x in v if x > 0.6 * max(v)]
        print('%-60s grid=%-7s n=%-5d avg_ns=%-9.0f live_n=%-5d live_avg_ns=%.0f' % (k, g, len(v), sum(v) / len(v), len(live), sum(live) / len(live)))
        summary['per_grid']['%s grid=%s' % (k, g)] = {'n': len(v), 'avg_ns': sum(v) / len(v), 'live_n': len(live), 'live_avg_ns': sum(live) / len(live)}
for tag, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    f = find(tag, '*counter_collection.csv')
    if not f:
        continue
    vals = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == ctr:
            k = r.get('Kernel_Name', '')[:60] + ' grid=' + str(r.get('Grid_Size', ''))
            vals[k].append(float(r.get('Counter_Value', 0)))
    print('== %s per dispatch (raw counter units, KB): all dispatches / live dispatches (gated no-ops dropped) ==' % ctr)
    summary[ctr] = {}
    for k in sorted(vals, key=lambda x: -sum(vals[x]))[:15]:
        v = vals[k]
        live = [x for x in v if x > 0.5 * max(v)] or v
        print('%-70s dispatches=%-6d avg=%.3f live=%-6d live_avg=%.3f' % (k, len(v), sum(v) / len(v), len(live), sum(live) / len(live)))
        summary[ctr][k] = {'dispatches': len(v), 'avg_kb': sum(v) / len(v), 'live_dispatches': len(live), 'live_avg_kb': sum(live) / len(live)}
json.dump(summary, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
