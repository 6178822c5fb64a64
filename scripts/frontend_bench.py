"""Front-end timing on the GPU box: correlation / warp / scale kernels against their HBM rooflines, network
forward (fp32 and bf16-autocast frozen nets), full TartanVO forward and one bilevel step (BASELINE config 2 shapes)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from islam_amd import ops, synthetic

dev = torch.device('cuda:0')
out = {}


def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


B = 8
shapes = [(196, 7, 10), (128, 14, 20), (96, 28, 40), (64, 56, 80), (32, 112, 160)]
tot_t = tot_b = 0
for C, H, W in shapes:
    f1, f2 = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
    t = timeit(lambda: ops.corr81_forward(f1, f2))
    byts = 4 * B * H * W * (2 * C + 81)
    out['corr81_%dx%dx%d' % (C, H, W)] = dict(us=t * 1e6, GBs=byts / t / 1e9, bytes=byts)
    tot_t += t; tot_b += byts
out['corr81_all_levels'] = dict(us=tot_t * 1e6, GBs=tot_b / tot_t / 1e9, bytes=tot_b, frac_of_8TBs=tot_b / tot_t / 8e12)
for (C, H, W), sc in zip(shapes[1:], (0.625, 1.25, 2.5, 5.0)):
    x, fl = torch.randn(B, C, H, W, device=dev), torch.randn(B, 2, H, W, device=dev)
    t = timeit(lambda: ops.warp_mask(x, fl, sc))
    byts = 4 * B * H * W * (2 * C + 2)
    out['warp_%dx%dx%d' % (C, H, W)] = dict(us=t * 1e6, GBs=byts / t / 1e9)
H, W = 112, 160
disp, flow = torch.rand(B, 1, H, W, device=dev) * 20 + 5, torch.randn(B, 2, H, W, device=dev) * 3
pose = torch.tensor([[0.1, 0.0, 1.0, 0, 0, 0, 1.0]], device=dev).repeat(B, 1)
intr = torch.tensor([[214.7, 214.7, 130.0, 55.3]], device=dev).repeat(B, 1)
base, th = torch.full((B,), 0.54, device=dev), torch.full((B,), 5.0, device=dev)
edge = torch.rand(B, H, W, device=dev) > 0.5
t = timeit(lambda: ops.scale_ls(disp, flow, pose, intr, base, edge, th))
out['scale_ls_B8'] = dict(us=t * 1e6, GBs=B * H * W * 19 / t / 1e9)

from islam_amd.TartanVO import TartanVO
from islam_amd.edges import edge_mask
for name, dt in (('fp32', None), ('bf16', torch.bfloat16)):
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), frozen_dtype=dt)
    with torch.no_grad():
        vo.vonet.stereoNet.conv_c13.weight.zero_(); vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    sample = synthetic.stereo_batch(B)
    sample = {k: (v.to(dev) if isinstance(v, torch.Tensor) and k.startswith('img') or k == 'intrinsic' else v) for k, v in sample.items()}
    t = timeit(lambda: vo(sample)['motion'], reps=10, warm=3)
    out['tartanvo_forward_B8_' + name] = dict(ms=t * 1e3, frames_per_s=B / t, conv_TFLOPs=466.4e9 * B / t / 1e12)
    imgs = sample['img0']
    te = timeit(lambda: edge_mask(imgs), reps=10, warm=2)
    out['edge_mask_B8'] = dict(ms=te * 1e3)
    with torch.no_grad():
        vo.vonet.train()
        x1, x2 = torch.cat([sample['img0'], sample['img1']], 1), torch.cat([sample['img0_norm'], sample['img0_r_norm']], 1)
        ctx = torch.autocast('cuda', dtype=dt, enabled=dt is not None)
        with ctx:
            tf = timeit(lambda: vo.vonet.flowNet(x1), reps=10, warm=3)
            ts = timeit(lambda: vo.vonet.stereoNet(x2), reps=10, warm=3)
        out['nets_B8_' + name] = dict(pwc_ms=tf * 1e3, stereo_ms=ts * 1e3, pwc_TFLOPs=105.15e9 * B / tf / 1e12, stereo_TFLOPs=359.4e9 * B / ts / 1e12)
print(json.dumps(out, indent=1))
