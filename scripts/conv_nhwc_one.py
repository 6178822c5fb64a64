"""One shape of islam_conv_nhwc_bf16, a few launches: the workload of scripts/conv_pmc_detail.sh (SQ counters of the kernel).
SHAPE = Cin,Cout,k,H,W (default: lastconv of the stereo feature extractor, 352->128 3x3 at 224x320, 16 images)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
Cin, Cout, k, H, W = [int(v) for v in os.environ.get('SHAPE', '352,128,3,224,320').split(',')]
x = torch.randn(16, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).to(torch.bfloat16)
wp = ops.pack_conv_nhwc_weight(w)
for _ in range(6):
    ops.conv_nhwc(x, wp, Cout, k)
torch.cuda.synchronize()
print('done', Cin, Cout, k, H, W)
