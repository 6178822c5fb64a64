"""TartanVO forward (B=8, 448x640) with the frozen nets in fp32 / bf16 / fp16 execution copies: time and output drift."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import synthetic
from islam_amd.TartanVO import TartanVO
dev = torch.device('cuda:0')
B = 8
s = synthetic.stereo_batch(B, seed=100)
s = {kk: (v.to(dev) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v) for kk, v in s.items()}
ref = None
for st_dt, fl_dt in ((None, None), (torch.bfloat16, None), (torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16), (torch.bfloat16, torch.float16)):
    torch.manual_seed(0)
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True, frozen_dtype=st_dt, flow_dtype=fl_dt)
    with torch.no_grad():
        vo.vonet.stereoNet.conv_c13.weight.zero_(); vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    try:
        for _ in range(3):
            r = vo(s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = vo(s)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        out = {k: r[k].detach().float() for k in ('flow', 'disp')}
        if ref is None:
            ref = out
        d = {k: float((out[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-9)) for k in out}
        print('stereo=%s flow=%s: %.1f ms  (%.0f frames/s)  rel max diff vs fp32: %s' % (st_dt, fl_dt, dt, B / dt * 1e3, d), flush=True)
    except Exception as e:
        print('stereo=%s flow=%s FAILED: %s' % (st_dt, fl_dt, repr(e)[:300]), flush=True)
