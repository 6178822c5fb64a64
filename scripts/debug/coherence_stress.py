"""Does a consumer kernel see ALL the bytes its producer (the kernel in front of it on the SAME stream) wrote, while a second stream keeps the
chip busy?  scripts/debug/scribble_probe.py showed islam_scale_ls reading 16 stale bytes of the uint8 edge map that `edge.to(uint8)`
had written one launch earlier -- rarely, only beside a running graph replay, gone after a device synchronisation.

main stream, per iteration: a fresh bool map -> .to(uint8) [producer] -> consumer immediately -> the same consumer again after
torch.cuda.synchronize(); the two results must be equal.  consumers: 'torch' (a reduction over rows: another block -> data mapping than
the producer's), 'scale' (islam_scale_ls), 'copy' (u.clone(), same mapping).  load on the side stream: 'matmul' | 'replay' (the frozen
nets' graph) | 'none'."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
load = os.environ.get('LOAD', 'matmul')
iters = int(os.environ.get('ITERS', '300'))
B, H, W = 8, 112, 160
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream(dev)
if load == 'matmul':
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)

    def burst():
        for _ in range(6):
            torch.mm(a, b)
elif load == 'replay':
    import test_benched_frontend_gpu as T
    vo = T._make(dev, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True, pose_channels_last=True, graph_frozen=True)
    smp = T._samples(dev, 1)[0]
    imgs = [smp[k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm')]
    with torch.no_grad():
        vo.vonet.set_mode(True)
        vo.vonet.frozen_forward(*imgs)
    torch.cuda.synchronize()

    def burst():
        with torch.no_grad():
            vo.vonet.frozen_forward(*imgs)
else:
    def burst():
        pass

disp = torch.full((B, 1, H, W), 10.0, device=dev)
flow = torch.randn(B, 2, H, W, device=dev, generator=g)
pose7 = torch.tensor([[0.1, 0.2, 1.0, 0, 0, 0, 1.0]] * B, device=dev)
intr4 = torch.tensor([[180.0, 180.0, 80.0, 56.0]] * B)
baseline = torch.full((B,), 0.5)
th = torch.full((B,), 5.0)
bad = {'torch': 0, 'scale': 0, 'copy': 0, 'scale_dev_args': 0, 'torch_h2d': 0}
intr4_d, baseline_d, th_d = intr4.to(dev), baseline.to(dev), th.to(dev)
small = torch.arange(32, dtype=torch.float32)
for it in range(iters):
    with torch.cuda.stream(side):
        burst()
    e = torch.rand(B, H, W, device=dev, generator=g) > 0.5
    # --- producer -> consumer pairs, nothing in between
    u = e.to(torch.uint8)
    r_torch = u.sum(dim=(0, 1))                                    # (W,): column sums
    u2 = e.to(torch.uint8)
    r_copy = u2.clone()
    r_scale = ops.scale_ls(disp, flow, pose7, intr4, baseline, e, th)[2]        # (converts e itself, then the kernel)
    # the same kernel with its small arguments ALREADY on the device: no host-to-device copy between producer and consumer
    r_scale_d = ops.scale_ls(disp, flow, pose7, intr4_d, baseline_d, e, th_d)[2]
    # torch consumer with a small pageable host-to-device copy between producer and consumer
    u3 = e.to(torch.uint8)
    _ = small.to(dev)
    r_torch_h2d = u3.sum(dim=(0, 1))
    torch.cuda.synchronize()
    bad['scale_dev_args'] += int(not torch.equal(r_scale_d, ops.scale_ls(disp, flow, pose7, intr4_d, baseline_d, e, th_d)[2]))
    bad['torch_h2d'] += int(not torch.equal(r_torch_h2d, u3.sum(dim=(0, 1))))
    bad['torch'] += int(not torch.equal(r_torch, u.sum(dim=(0, 1))))
    bad['copy'] += int(not torch.equal(r_copy, u2))
    bad['scale'] += int(not torch.equal(r_scale, ops.scale_ls(disp, flow, pose7, intr4, baseline, e, th)[2]))
print('load=%s iters=%d mismatches: %s' % (load, iters, bad))
