"""Summarise the PMC pass of scripts/conv_mfma_pmc.sh: per (grid size) dispatch group of conv3x3_mfma_kernel, the average
counter values and MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 4 SIMDs * 256 CUs)
(MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16 and SIMD)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
f = glob.glob(os.path.join(root, 'pmc', '**', '*counter_collection.csv'), recursive=True)
if not f:
    sys.exit('no counter_collection.csv under %s' % root)
rows = list(csv.DictReader(open(f[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get('Kernel_Name', '')
    if 'conv3x3_mfma_kernel' not in name:
        continue
    key = (name.split('(')[0][-40:], r.get('Grid_Size', ''), r.get('Dispatch_Id', ''))
    acc[(key[0], key[1])][r['Counter_Name']].append(float(r['Counter_Value']))
print('conv3x3_mfma_kernel dispatch groups (same template instance and grid): counters averaged over dispatches')
print('%-42s %10s %6s %14s %14s %14s %9s' % ('kernel', 'grid', 'n', 'MFMA_BUSY_CYC', 'GUI_ACTIVE', 'MFMA_MOPS_BF16', 'MfmaUtil'))
for (k, g), c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', [0]))):
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    busy, act = avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), avg.get('GRBM_GUI_ACTIVE', 0.0)
    util = busy / (act * 4 * 256) if act else float('nan')
    print('%-42s %10s %6d %14.0f %14.0f %14.0f %8.1f%%' % (k, g, len(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [])), busy, act,
                                                          avg.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0), 100 * util))
