"""Summarise scripts/frozen_pmc.sh: per kernel family (template instance, grid) of the hand-written front-end kernels -- launches, average
duration, matrix-core busy share (SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x clock): against the 2.4 GHz peak clock AND against
the shader clock the pass measured, GRBM_GUI_ACTIVE / duration), HBM bytes per launch (2 x FETCH_SIZE KB + WRITE_SIZE KB) and GB/s."""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
XCDS = 8.0          # GRBM_GUI_ACTIVE comes back summed over the chip's 8 XCDs (a memory-bound kernel reads 8 x 2.4 GHz)


def load(sub):
    f = glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True)
    disp = {}
    if not f:
        return disp
    for r in csv.DictReader(open(f[0])):
        d = disp.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']),
                                               'ns': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
        d[r['Counter_Name']] = float(r['Counter_Value'])
    return disp


def fam(name, grid):
    m = re.search(r'(conv_nhwc_kernel|conv3x3_ws32_kernel|conv3x3_ws_kernel|conv3x3_mfma_kernel|pyr_level_kernel|hg_residual_kernel|hg_residual64_kernel|corr81_fwd4_kernel|corr81_fwd_kernel|warp_mask_kernel|'
                  r'resize_bilinear\w*|bn_apply_kernel|maxpool2\w*|nchw_to_nhwc_kernel|deconv4x4s2_to2_kernel)(<[^>]*>)?', name)
    if not m:
        return None
    return '%s%s grid=%d' % (m.group(1), (m.group(2) or '').replace(' ', ''), grid)


mf, fe, wr = load('mfma'), load('fetch'), load('write')
groups = collections.defaultdict(lambda: {'n': 0, 'ns': 0.0, 'busy': 0.0, 'gui': 0.0})
for d in mf.values():
    k = fam(d['name'], d['grid'])
    if k is None:
        continue
    g = groups[k]
    g['n'] += 1; g['ns'] += d['ns']; g['busy'] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); g['gui'] += d.get('GRBM_GUI_ACTIVE', 0.0)
byts = collections.defaultdict(lambda: {'nf': 0, 'fetch': 0.0, 'nsf': 0.0, 'nw': 0, 'write': 0.0})
for d in fe.values():
    k = fam(d['name'], d['grid'])
    if k:
        byts[k]['nf'] += 1; byts[k]['fetch'] += d.get('FETCH_SIZE', 0.0); byts[k]['nsf'] += d['ns']
for d in wr.values():
    k = fam(d['name'], d['grid'])
    if k:
        byts[k]['nw'] += 1; byts[k]['write'] += d.get('WRITE_SIZE', 0.0)
print('%-58s %5s %9s %8s %8s %8s %10s %8s' % ('kernel family', 'n', 'avg_us', 'MFMA%pk', 'clk_GHz', 'MFMA%clk', 'HBM_MB', 'TB/s'))
tot = collections.defaultdict(float)
for k, g in sorted(groups.items(), key=lambda kv: -kv[1]['ns']):
    us = g['ns'] / g['n'] / 1e3
    clk = g['gui'] / XCDS / g['ns'] if g['ns'] else 0.0               # GHz: GUI-active cycles per ns (the counter is summed over the 8 XCDs)
    util_pk = 100.0 * g['busy'] / (g['ns'] * 2.4 * 1024)
    util_clk = 100.0 * g['busy'] / (g['gui'] / XCDS * 1024) if g['gui'] else 0.0
    b = byts.get(k)
    mb = tbs = float('nan')
    if b and b['nf'] and b['nw']:
        mb = (2.0 * b['fetch'] / b['nf'] + b['write'] / b['nw']) * 1024 / 1e6          # FETCH/WRITE_SIZE are in KB; fetch doubled (gfx950)
        tbs = mb * 1e6 / (us * 1e-6) / 1e12
    print('%-58s %5d %9.1f %7.1f%% %8.2f %7.1f%% %10.1f %8.2f' % (k[:58], g['n'], us, util_pk, clk, util_clk, mb, tbs))
    if 'conv' in k or 'pyr_level' in k or 'hg_residual' in k:
        tot['busy'] += g['busy']; tot['ns'] += g['ns']; tot['gui'] += g['gui']
# everything ONE forward launches (hand-written and library kernels alike): the pass records `reps` forwards, the first of which also
# carries weight packing and MIOpen's set-up; the steady-state forward is the period of the kernel-name sequence at the end of the trace
import json
reps = float(os.environ.get('FWD_REPS', '4'))
seq = [mf[k] for k in sorted(mf, key=lambda i: int(i))]
names = [d['name'] for d in seq]
period = None
for K in range(50, len(names) // 2 + 1):
    if names[-K:] == names[-2 * K:-K]:
        period = K
        break
last = seq[-period:] if period else seq
all_busy = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for d in last)
all_ns = sum(d['ns'] for d in last)
all_gui = sum(d.get('GRBM_GUI_ACTIVE', 0.0) for d in last)
# the clock: GRBM_GUI_ACTIVE also counts the cycles around a dispatch, so short kernels read far too high; kernels of >= 50 us only
long_ = [d for d in last if d['ns'] >= 50000]
clk_gui = sum(d.get('GRBM_GUI_ACTIVE', 0.0) for d in long_)
clk_ns = sum(d['ns'] for d in long_)
summary = {
    'what': 'scripts/frozen_pmc.sh: the last of %d eager forwards of the frozen stereo + flow nets at B=8, 448x640 (%s kernel launches)' % (int(reps), period),
    'mfma_busy_cycles_per_forward': all_busy,
    # one busy cycle of a SIMD's matrix pipe = 1024 bf16 flops (v_mfma_f32_32x32x16_bf16: 32768 flops in 32 cycles; 16x16x32: 16384 in 16);
    # the counter is the sum over the chip's 1024 SIMDs
    'executed_gflop_per_forward': all_busy * 1024 / 1e9,
    'executed_gflop_per_frame': all_busy * 1024 / 1e9 / 8.0,
    'kernel_ms_per_forward': all_ns / 1e6,
    'shader_clock_ghz_time_weighted': clk_gui / XCDS / clk_ns if clk_ns else None,
    'shader_clock_note': 'GRBM_GUI_ACTIVE / 8 XCDs / duration over the kernels of >= 50 us (%.0f %% of the kernel time); kernels run one at a time under the counter pass' % (100.0 * clk_ns / max(all_ns, 1)),
}
print(json.dumps(summary))
with open(os.path.join(root, 'exec_summary.json'), 'w') as f:
    json.dump(summary, f, indent=1)
if tot['ns']:
    print('matrix-core kernels together (time-weighted): MFMA busy %.1f %% of the 2.4 GHz peak, %.1f %% at the measured clock (%.2f GHz)'
          % (100 * tot['busy'] / (tot['ns'] * 2.4 * 1024), 100 * tot['busy'] / (tot['gui'] / XCDS * 1024), tot['gui'] / XCDS / tot['ns']))
