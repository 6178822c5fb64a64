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
for tag, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    f = find(tag, '*counter_collection.csv')
    if not f:
        continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == ctr:
            k = r.get('Kernel_Name', '')[:60] + ' grid=' + str(r.get('Grid_Size', ''))
            acc[k] += float(r.get('Counter_Value', 0))
            cnt[k] += 1
    print('== %s per dispatch (raw counter units, KB) ==' % ctr)
    summary[ctr] = {}
    for k in sorted(acc, key=lambda x: -acc[x])[:15]:
        print('%-70s dispatches=%-6d avg=%.3f' % (k, cnt[k], acc[k] / cnt[k]))
        summary[ctr][k] = {'dispatches': cnt[k], 'avg_kb': acc[k] / cnt[k]}
json.dump(summary, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
