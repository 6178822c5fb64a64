"""How much slower does the trainable pose head (forward + backward, HIP-graph replays: ~400 tiny kernels) get while the frozen nets of
another batch run on a side stream?  (The pipelined bilevel step is bound by exactly that.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
dev = torch.device('cuda:0')
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev).train()
vonet.set_frozen_dtype(torch.bfloat16, torch.bfloat16)
vonet.set_graph_frozen(True)
vonet.set_pose_channels_last(True)
vonet.graph_pose = True
imgs = [torch.randn(8, 3, 448, 640, device=dev) for _ in range(4)]
intr = torch.randn(8, 2, 112, 160, device=dev)
with torch.no_grad():
    flow, disp = vonet.frozen_forward(*imgs)


def pose_step():
    _, _, pose = vonet(imgs[0], imgs[1], imgs[2], imgs[3], intr, frozen=(flow, disp))
    g = torch.autograd.grad(pose.sum(), [p for p in vonet.flowPoseNet.parameters() if p.requires_grad])
    return g


for _ in range(4):
    pose_step()
torch.cuda.synchronize()


import ctypes
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(reserve_groups=2, ncu=256):
    """all CUs but `reserve_groups` per XCD (bit i of the mask = CU i / 8 of XCD i % 8)"""
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    per = ncu // 8
    off = set()
    for g in range(reserve_groups):
        c = per - 1 - g * 8
        off.update(range(c * 8, c * 8 + 8))
    for i in range(ncu):
        if i not in off:
            mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask) == 0
    return torch.cuda.ExternalStream(st.value, device=dev)


def timed(n, side_busy, side=None, main=None):
    side = side or torch.cuda.Stream(device=dev)
    if main is not None:
        with torch.cuda.stream(main):
            return timed(n, side_busy, side, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if side_busy:
            with torch.cuda.stream(side), torch.no_grad():
                vonet.frozen_forward(*imgs)
        pose_step()
        torch.cuda.current_stream().synchronize()
    el = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return el


print('pose head fwd + bwd alone             : %.2f ms per step' % timed(10, False))
print('pose head fwd + bwd beside frozen nets: %.2f ms per step (main stream only; the side stream keeps running)' % timed(10, True))
for groups in (2, 4, 8):
    print('  ... side stream confined to %d CUs, pose head on the default stream : %.2f ms per step' % (256 - 8 * groups, timed(10, True, masked_stream(groups))))
    print('  ... side stream confined to %d CUs, pose head on its own stream     : %.2f ms per step' % (256 - 8 * groups, timed(10, True, masked_stream(groups), torch.cuda.Stream(device=dev))))
print('  ... plain side stream, pose head on a high-priority stream           : %.2f ms per step' % timed(10, True, None, torch.cuda.Stream(device=dev, priority=-1)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
with torch.no_grad():
    for _ in range(5):
        vonet.frozen_forward(*imgs)
e1.record(); torch.cuda.synchronize()
print('frozen forward alone                  : %.2f ms' % (e0.elapsed_time(e1) / 5))
