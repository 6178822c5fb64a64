"""Bisect of scripts/debug/coherence_stress.py: which LOAD on the side stream, and what between producer and consumer, makes islam_scale_ls
read bytes its producer wrote one launch earlier as stale?  Prints mismatch counts per consumer variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
load = os.environ.get('LOAD', 'replay')
iters = int(os.environ.get('ITERS', '300'))
B, H, W = 8, 112, 160
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream(dev)
if load in ('replay', 'eager', 'stereo_eager', 'flow_eager'):
    import test_benched_frontend_gpu as T
    vo = T._make(dev, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True, pose_channels_last=True, graph_frozen=(load == 'replay'))
    smp = T._samples(dev, 1)[0]
    imgs = [smp[k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm')]
    vn = vo.vonet
    vn.set_mode(True)
    with torch.no_grad():
        vn.frozen_forward(*imgs)
    torch.cuda.synchronize()

    def burst():
        with torch.no_grad():
            if load == 'stereo_eager':
                vn._run_frozen('stereo', vn.stereoNet, vn.frozen_dtype, torch.cat((imgs[2], imgs[3]), 1), quarter=True)
            elif load == 'flow_eager':
                vn._run_frozen('flow', vn.flowNet, vn.flow_dtype, torch.cat((imgs[0], imgs[1]), 1))
            else:
                vn.frozen_forward(*imgs)
elif load == 'graph_mm':
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, b)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        c = a
        for _ in range(40):
            c = torch.mm(c, b) * 1e-2
    def burst():
        gr.replay()
else:
    def burst():
        pass

disp = torch.full((B, 1, H, W), 10.0, device=dev)
flow = torch.randn(B, 2, H, W, device=dev, generator=g)
pose7 = torch.tensor([[0.1, 0.2, 1.0, 0, 0, 0, 1.0]] * B, device=dev)
intr4 = torch.tensor([[180.0, 180.0, 80.0, 56.0]] * B).to(dev)
baseline = torch.full((B,), 0.5).to(dev)
th = torch.full((B,), 5.0).to(dev)
names = ['direct', 'spacer_kernel', 'old_producer', 'f32_producer']
bad = {k: 0 for k in names}
sc = lambda u, fl=flow: ops.scale_ls(disp, fl, pose7, intr4, baseline, u, th)[2]
for it in range(iters):
    with torch.cuda.stream(side):
        burst()
    e = torch.rand(B, H, W, device=dev, generator=g) > 0.5
    u_old = e.to(torch.uint8)
    torch.cuda.synchronize()                  # u_old has been in memory for a while
    with torch.cuda.stream(side):
        burst()
    u1 = e.to(torch.uint8); r1 = sc(u1)                                   # producer, consumer
    u2 = e.to(torch.uint8); _sp = disp.sum(); r2 = sc(u2)                 # a kernel in between
    r3 = sc(u_old)                                                        # no fresh producer at all
    f4 = flow * 1.0; r4 = sc(u_old, f4)                                   # fresh fp32 producer (flow), old edge
    torch.cuda.synchronize()
    for k, r, u, fl in (('direct', r1, u1, flow), ('spacer_kernel', r2, u2, flow), ('old_producer', r3, u_old, flow), ('f32_producer', r4, u_old, f4)):
        bad[k] += int(not torch.equal(r, sc(u, fl)))
        torch.cuda.synchronize()
print('load=%s iters=%d mismatches: %s' % (load, iters, bad))
