"""Inside PWCDCNet.forward_mfma on the golden flow input: every call of _head_up_mirror against torch's Conv2d / ConvTranspose2d in
float64 on the same mirror (bf16 operands) -- is the matrix-core head itself right on this box?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.golden.netfill import fill_state_dict, make_input
from tests.test_golden_gpu import _g, _stats
from islam_amd import nets
print(torch.cuda.get_device_name(0))
ref = _g('pwc')
net = fill_state_dict(nets.PWCDCNet()).to('cuda').eval()
x = make_input('pwc').to('cuda')
orig = net._head_up_mirror
r = lambda t: t.detach().to(torch.bfloat16).double()


def checked(l, mir, tot, up, up_out=None, up_coff=0):
    flow, upf = orig(l, mir, tot, up, up_out=up_out, up_coff=up_coff)
    head = getattr(net, 'predict_flow%d' % l)
    C = head.weight.shape[1]
    x64 = mir[:, :C].double()
    want = torch.nn.functional.conv2d(x64, r(head.weight), head.bias.double(), padding=1)
    msg = 'level %d: mirror %s finite %s  head max|diff| %.3e of %.3e' % (l, tuple(mir.shape), bool(torch.isfinite(mir.float()).all()),
                                                                          float((flow.double() - want).abs().max()), float(want.abs().max()))
    if up:
        dc = getattr(net, 'upfeat%d' % l)
        wu = torch.nn.functional.conv_transpose2d(x64, r(dc.weight), dc.bias.double(), stride=2, padding=1)
        got = up_out[:, up_coff:up_coff + 2] if up_out is not None else upf
        msg += '  upfeat max|diff| %.3e of %.3e' % (float((got.double() - wu).abs().max()), float(wu.abs().max()))
    print(msg)
    return flow, upf


net._head_up_mirror = checked
with torch.no_grad():
    flows, _ = net.forward_mfma(x)
print('  '.join('flow%d max %.3e rms %.3e' % ((i,) + _stats(f, ref['flow%d' % i])[:2]) for i, f in enumerate(flows)))
