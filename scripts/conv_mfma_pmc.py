"""Summarise the PMC pass of scripts/conv_mfma_pmc.sh for conv3x3_mfma_kernel, grouped by (template instance, grid):
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs)   [MI355X_MICROARCH.md:
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16 on its SIMD; the counter is summed over the chip],
with the duration taken from the same pass (dispatch start/end timestamps of the counter rows), and the MFMA count
= busy / 32 as a cross-check against 2*B*H*W*Cout*9*Cin / 32768 flops per instruction."""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
f = glob.glob(os.path.join(root, 'pmc', '**', '*counter_collection.csv'), recursive=True)
if not f:
    sys.exit('no counter_collection.csv under %s' % root)
disp = {}
for r in csv.DictReader(open(f[0])):
    name = r['Kernel_Name']
    if 'conv3x3_mfma_kernel' not in name and 'conv_nhwc_kernel' not in name:
        continue
    m = re.search(r'(conv3x3_mfma|conv_nhwc)_kernel<([\d, ]+)>', name)
    d = disp.setdefault(r['Dispatch_Id'], {'k': '%s<%s>' % (m.group(1), m.group(2).replace(' ', '')) if m else 'conv',
                                           'grid': int(r['Grid_Size']),
                                           'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
    d[r['Counter_Name']] = float(r['Counter_Value'])
groups = collections.defaultdict(list)
for d in disp.values():
    groups[(d['k'], d['grid'])].append(d)
print('%-24s %9s %5s %10s %14s %12s %9s' % ('kernel', 'grid', 'n', 'avg_us', 'MFMA_BUSY_CYC', 'MFMAs', 'MfmaUtil'))
tb = tt = 0.0
for (k, g), ds in sorted(groups.items(), key=lambda kv: -sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for d in kv[1])):
    busy = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for d in ds) / len(ds)
    ns = sum(d['ns'] for d in ds) / len(ds)
    tb += busy * len(ds); tt += ns * len(ds)
    print('%-24s %9d %5d %10.1f %14.0f %12.0f %8.1f%%' % (k, g, len(ds), ns / 1e3, busy, busy / 32, 100 * busy / (ns * 2.4 * 1024)))
print('all dispatches of the hand-written convolution kernels: MfmaUtil %.1f%% (time-weighted)' % (100 * tb / (tt * 2.4 * 1024)))
