"""Per kernel family / grid: mean of every counter collected by scripts/debug/pose_head_pmc.sh."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv_gemm' not in n and 'conv_bwd_pair' not in n:
            continue
        i = n.find('conv_')
        key = (n[i:i + 34].split('(')[0], r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    print('%-36s grid %8s wg %5s  launches %d' % (k[0], k[1], k[2], max(len(v) for v in acc[k].values())))
    print('    ' + '  '.join('%s=%.3g' % (c, sum(acc[k][c]) / len(acc[k][c])) for c in names if c in acc[k]))
