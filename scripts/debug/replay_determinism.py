"""Bit-reproducibility of the frozen nets' forward under concurrency: the captured graph (flow and stereo branches side by side) replayed N
times on the same inputs, with and without a conv_nhwc loop on another stream; every output must equal the first replay's bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_benched_frontend_gpu as T
from islam_amd import ops
dev = torch.device('cuda:0')
n = int(os.environ.get('N', '60'))
vo = T._make(dev, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True, pose_channels_last=True, graph_frozen=os.environ.get('GF', '1') == '1')
smp = T._samples(dev, 1)[0]
imgs = [smp[k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm')]
vn = vo.vonet
vn.set_mode(False if os.environ.get('EVAL', '1') == '1' else True)     # eval: BatchNorm running statistics, so replays are identical by construction
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(16, 128, 112, 160, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = ops.pack_conv_nhwc_weight(torch.randn(128, 128, 3, 3, device=dev, generator=g) / 30)
side = torch.cuda.Stream(dev)
with torch.no_grad():
    f0, d0 = vn.frozen_forward(*imgs)
    torch.cuda.synchronize()
    for load in (False, True):
        bad_f = bad_d = 0
        for i in range(n):
            if load:
                with torch.cuda.stream(side):
                    for _ in range(40):
                        ops.conv_nhwc(x, w, 128, 3)
            f, d = vn.frozen_forward(*imgs)
            torch.cuda.synchronize()
            bad_f += int(not torch.equal(f, f0)); bad_d += int(not torch.equal(d, d0))
        print('side-stream conv load %-5s: flow differs in %d, disparity in %d of %d forwards' % (load, bad_f, bad_d, n))
