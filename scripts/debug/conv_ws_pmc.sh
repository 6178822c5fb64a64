#!/bin/bash
# Shader clock and matrix-pipe busy share of the weight-stationary convolution and of the tile kernel beside it (scripts/conv_ws_bench.py) under
# rocprofv3 --pmc (kernel trace only beside the counters).  Run on the GPU box: bash scripts/debug/conv_ws_pmc.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/wspmc -o f -- python3 $GRAFT_REPO_ROOT/scripts/conv_ws_bench.py > $OUT/conv_ws_pmc.log 2>&1
python3 - <<'PY' >> $OUT/conv_ws_pmc.log 2>&1
import csv, glob, collections, re
f = glob.glob('/tmp/wspmc/**/*counter_collection.csv', recursive=True)[0]
disp = {}
for r in csv.DictReader(open(f)):
    d = disp.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'], 'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
g = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d in disp.values():
    m = re.search(r'(conv3x3_ws_kernel<[^>]*>|conv_nhwc_kernel<[^>]*>)', d['name'])
    if not m: continue
    k = g[m.group(1)]
    k[0] += 1; k[1] += d['ns']; k[2] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); k[3] += d.get('GRBM_GUI_ACTIVE', 0.0)
print('%-44s %5s %8s %8s %8s %9s' % ('kernel', 'n', 'avg_us', 'clk_GHz', 'MFMA%pk', 'MFMA%clk'))
for name, (n, ns, busy, gui) in sorted(g.items()):
    print('%-44s %5d %8.1f %8.2f %7.1f%% %8.1f%%' % (name, n, ns / n / 1e3, gui / 8 / ns, 100 * busy / (ns * 2.4 * 1024), 100 * busy / (gui / 8 * 1024)))
PY
tail -20 $OUT/conv_ws_pmc.log
