"""Event clocks (100 MHz wall clock, workgroup 0 / wave 0) of one corr81_fwd4_kernel launch at the level-2 shape, B = 8:
   bash scripts/debug/corr_stamps.sh   (build box)   then on the GPU box
   ISLAM_HIP_LIB=islam_amd/lib/libislam_probe_corr.so python scripts/debug/corr_stamps.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
C, H, W = 32, 112, 160
f1 = torch.randn(8, C, H, W, device=dev); f2 = torch.randn(8, C, H, W, device=dev)
buf = torch.empty(8, 89, H, W, device=dev)
L = ctypes.CDLL(os.environ['ISLAM_HIP_LIB'])
dbg = int(os.environ.get('CORR_DBG', '0'))
L.islam_corr_dbg_set(dbg)
for _ in range(5): ops.corr81_act(f1, f2, buf, 8, 0.1)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): ops.corr81_act(f1, f2, buf, 8, 0.1)
b.record(); torch.cuda.synchronize()
print('dbg %d (1: no stores, 2: no global loads): %.1f us per call' % (dbg, a.elapsed_time(b) / 50 * 1e3))
out = (ctypes.c_longlong * 64)()
L.islam_corr_stamps(out)
n = out[0]
ev = [out[1 + i] / 100.0 for i in range(n)]
print('variant %s: %d events, kernel life of this wave %.2f us' % (os.environ.get('ISLAM_CORR4_VARIANT', '0'), n, out[63] / 100.0))
names = ['barrier 1', 'commit (loads arrived)', 'barrier 2', 'next fetch issued', 'multiplied']
print('  first fetch issued at %.2f us' % ev[0])
i, prev = 1, ev[0]
while i < n:
    row = []
    for k in range(5):
        if i >= n: break
        row.append('%s +%.2f' % (names[k], ev[i] - prev)); prev = ev[i]; i += 1
    print('  chunk: ' + ', '.join(row) + '   (t = %.2f)' % prev)
    # a tile-end stamp follows every second chunk at this shape (C = 32 = two chunks)
    if i < n and ((i - 1) % 11) == 10:
        print('  tile end: stores issued +%.2f' % (ev[i] - prev)); prev = ev[i]; i += 1
