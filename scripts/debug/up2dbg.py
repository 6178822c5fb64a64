import os, sys
sys.path.insert(0, '/root/repo')
import torch
from islam_amd import nets, ops
print('FLOW_UP2', nets.FLOW_UP2)
net = nets.PWCDCNet().cuda()
dc = net.upfeat6
print(dc, dc.weight.dtype, dc.out_channels, dc.kernel_size, dc.stride, dc.padding, dc.output_padding, dc.groups)
t = torch.randn(2, dc.in_channels, 7, 10, device='cuda')
calls = []
orig = ops.deconv_to2
ops.deconv_to2 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
y = net._up2('upfeat6', t)
print('used hip kernel:', bool(calls), y.shape)
