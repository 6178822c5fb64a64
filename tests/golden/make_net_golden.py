"""Generates tests/golden/nets_*.npz by IMPORTING THE REFERENCE modules from /root/reference
(build container only -- the reference never travels to the GPU box; only these vectors do).

  python tests/golden/make_net_golden.py

The reference's PWC net needs cupy (CUDA kernels as strings) and cv2: both are stubbed, and its
FunctionCorrelation / warp are replaced by the CPU oracle (oracle/corr81.c), which restates the
in-repo kernel source.  Everything else (conv stacks, BN in train mode, hourglass, SSP, pose heads,
IMU denoiser) is the reference's own code running on CPU."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
for name in ('cupy', 'cv2', 'pypose'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules['cupy'].memoize = lambda **kw: (lambda f: f)      # decorator used at import time (correlation.py:273)

from oracle import cwrap  # noqa: E402
from tests.golden.netfill import fill_state_dict, make_input, tame_vonet, vonet_sample  # noqa: E402


def oracle_corr(tenFirst, tenSecond):
    return torch.from_numpy(cwrap.corr81_fwd(tenFirst.detach().numpy(), tenSecond.detach().numpy()))


def oracle_warp(self, x, flo):
    return torch.from_numpy(cwrap.warp(x.detach().numpy(), flo.detach().numpy()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from Network.PWC import PWCNet as ref_pwc
    from Network.StereoNet7 import StereoNet7 as RefStereo
    from Network.VOFlowNet import VOFlowRes as RefPose
    from Network.IMUDenoiseNet import IMUCorrector_CNN_GRU_WO_COV as RefDen
    ref_pwc.FunctionCorrelation = oracle_corr
    ref_pwc.PWCDCNet.warp = oracle_warp

    keys = {}
    with torch.no_grad():
        # PWC-DC-Net, 128x192 image pair
        net = fill_state_dict(ref_pwc.PWCDCNet(uncertainty=False))
        x = make_input('pwc')
        flows, _ = net(x)
        np.savez_compressed(os.path.join(HERE, 'nets_pwc.npz'), **{'flow%d' % i: f.numpy() for i, f in enumerate(flows)})
        keys['flowNet'] = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        # StereoNet7, train-mode BatchNorm (TartanVO.py:91), 256x256, batch 2
        net = fill_state_dict(RefStereo())
        net.train()
        x = make_input('stereo')
        d, _ = net(x)
        rm = net.state_dict()['feature_extraction.firstconv.0.1.running_mean'].numpy().copy()
        np.savez_compressed(os.path.join(HERE, 'nets_stereo.npz'), disp=d.numpy(), running_mean_after=rm)
        keys['stereoNet'] = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        # VOFlowRes
        net = fill_state_dict(RefPose(intrinsic=True, down_scale=True, stereo=0, fix_parts=('flow', 'stereo')))
        x = make_input('pose')
        p = net(x)
        np.savez_compressed(os.path.join(HERE, 'nets_pose.npz'), pose=p.numpy())
        keys['flowPoseNet'] = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        # whole VONet (Network/VONet.py:28-39) at 448x640, train mode as TartanVO.forward sets it (TartanVO.py:91)
        from Network.VONet import VONet as RefVONet
        net = tame_vonet(fill_state_dict(RefVONet(fix_parts=('flow', 'stereo'))))
        net.train()
        sample = vonet_sample()
        args = [sample[k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm', 'intrinsic')]
        flow, disp, pose = net(*args)
        np.savez_compressed(os.path.join(HERE, 'nets_vonet.npz'), flow=flow.numpy(), disp=disp.numpy(), pose=pose.numpy())
        assert len(net.state_dict()) == 765
        # IMU denoiser, 83 samples (remainder stretch, Q14)
        net = fill_state_dict(RefDen())
        acc, gyro = make_input('acc'), make_input('gyro')
        ca, cg, _, _ = net({'acc': acc, 'gyro': gyro}, eval=True)
        np.savez_compressed(os.path.join(HERE, 'nets_denoise.npz'), cacc=ca.numpy(), cgyro=cg.numpy())
        keys['denoiser'] = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    import json
    with open(os.path.join(HERE, 'nets_keys.json'), 'w') as f:
        json.dump({k: {kk: list(vv) for kk, vv in v.items()} for k, v in keys.items()}, f)
    print({k: len(v) for k, v in keys.items()})


if __name__ == '__main__':
    main()
