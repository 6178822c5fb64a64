"""Phase clocks (wave 0 of tile 0) of every convolution launch of one pose-head forward + backward:
   bash scripts/debug/pose_head_stamps.sh   (build box)   then on the GPU box
   ISLAM_HIP_LIB=islam_amd/lib/libislam_probe_pose.so ISLAM_POSE_ONE_STREAM=1 python scripts/debug/pose_head_stamps.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from islam_amd import nets, pose_head
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = nets.VOFlowRes().to(dev).to(memory_format=torch.channels_last)
x = torch.randn(8, 4, 112, 160, device=dev).contiguous(memory_format=torch.channels_last)
gy = torch.randn(8, 6, device=dev)
h = pose_head.PoseHeadHip(net)
L = ctypes.CDLL(os.environ['ISLAM_HIP_LIB'])
for _ in range(3):
    y = h(x); y.backward(gy)
torch.cuda.synchronize()
L.islam_pose_stamps(None, 0)
y = h(x); y.backward(gy)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (160 * 16))()
L.islam_pose_stamps(buf, -1)
names = ['setup', 'issue first loads', 'first stash+barrier', 'issue loads', 'mma', 'stash(wait loads)', 'barrier', 'epilogue']
print('slot mode KC wide grid      chunks | wall us | shader clocks: ' + ', '.join(names))
for s in range(160):
    r = buf[s * 16:(s + 1) * 16]
    if r[12] == 0:
        continue
    print('%3d  %d  %3d  %d  %4dx%-3d  %3d | %7.2f | ' % (s, r[10], r[11], r[14], r[12], r[13], r[9], r[8] / 100.0) + ', '.join('%6d' % v for v in r[:8]))
