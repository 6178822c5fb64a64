"""Frozen forward (graph replay, as TartanVO.prefetch runs it) on a plain side stream / CU-masked streams: time per forward."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
dev = torch.device('cuda:0')
hip = ctypes.CDLL('libamdhip64.so')
ncu = 256
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev).train()
vonet.set_frozen_dtype(torch.bfloat16, torch.bfloat16)
vonet.set_graph_frozen(os.environ.get('GRAPH', '1') == '1')
imgs = [torch.randn(8, 3, 448, 640, device=dev) for _ in range(4)]


def masked(bits):
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(ncu):
        if bits(i):
            mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask) == 0
    return torch.cuda.ExternalStream(st.value, device=dev)


def run(s, n=6):
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            vonet.frozen_forward(*imgs)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(n):
            vonet.frozen_forward(*imgs)
        b.record(s)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print('graph replay: %s' % (os.environ.get('GRAPH', '1') == '1'))
print('default stream  %.2f ms' % run(torch.cuda.current_stream()))
print('plain side      %.2f ms' % run(torch.cuda.Stream(device=dev)))
print('masked all ones %.2f ms' % run(masked(lambda i: True)))
print('masked 240      %.2f ms' % run(masked(lambda i: not (248 <= i < 256 or 184 <= i < 192))))
