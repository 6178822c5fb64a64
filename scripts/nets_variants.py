import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
dev = torch.device('cuda:0')
B = 8
x2 = torch.randn(B, 6, 448, 640, device=dev)
x1 = torch.rand(B, 6, 448, 640, device=dev)

def timeit(fn, reps=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for bench in ((False, True) if os.environ.get('BOTH') else (False,)):
    torch.backends.cudnn.benchmark = bench
    for cl in (True,):
        for dt in (None, torch.bfloat16):
            st = nets.StereoNet7().to(dev).train()
            pw = nets.PWCDCNet().to(dev).train()
            a, b = x2, x1
            if cl:
                st, pw = st.to(memory_format=torch.channels_last), pw.to(memory_format=torch.channels_last)
                a, b = x2.contiguous(memory_format=torch.channels_last), x1.contiguous(memory_format=torch.channels_last)
            with torch.no_grad(), torch.autocast('cuda', dtype=dt, enabled=dt is not None):
                try:
                    ts = timeit(lambda: st(a))
                    tp = timeit(lambda: pw(b))
                    print('benchmark=%s channels_last=%s dtype=%s: stereo %.1f ms (%.0f TF)  pwc %.1f ms (%.0f TF)' % (bench, cl, dt, ts, 359.4 * B / ts, tp, 105.15 * B / tp), flush=True)
                except Exception as e:
                    print('benchmark=%s channels_last=%s dtype=%s: FAILED %s' % (bench, cl, dt, str(e)[:100]), flush=True)
