"""What hipExtStreamCreateWithCUMask's bit order means on this device: time one big convolution on streams with different masks."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
hip = ctypes.CDLL('libamdhip64.so')
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
print('CUs', ncu)
CL = torch.channels_last
x = torch.randn(16, 352, 224, 320, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
w = (torch.randn(128, 352, 3, 3, device=dev) / 56).to(torch.bfloat16)
wp = ops.pack_conv_nhwc_weight(w)


def stream_with(bits):
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(ncu):
        if bits(i):
            mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev), sum(1 for i in range(ncu) if bits(i))


def run(s):
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.conv_nhwc(x, wp, 128, 3)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(10):
            ops.conv_nhwc(x, wp, 128, 3)
        b.record(s)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 * 1e3


print('plain stream: %.0f us' % run(torch.cuda.Stream(device=dev)))
for name, f in (('all ones', lambda i: True), ('first half of the bits', lambda i: i < ncu // 2), ('even bits', lambda i: i % 2 == 0),
                ('bits with i % 8 == 0', lambda i: i % 8 == 0), ('bits with i // 32 == 0 (first word)', lambda i: i // 32 == 0),
                ('all but bits 248-255 and 184-191', lambda i: not (248 <= i < 256 or 184 <= i < 192)),
                ('all but i % 16 == 15', lambda i: i % 16 != 15)):
    s, n = stream_with(f)
    print('%-40s (%3d bits set): %.0f us' % (name, n, run(s)))
