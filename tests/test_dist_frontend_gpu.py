"""GPU test of the frame-parallel front-end with the bf16 execution copy: two ranks (gloo, both on the
one GPU of the test box) shard the frames of a window; the stereo net's BatchNorm layers are ShardedBatchNorm2d, which must
run their own all-reducing forward on the reduced-precision copy too -- never the fused single-rank BatchNorm / the
convolution-epilogue statistics -- so that outputs and the running statistics equal the un-sharded forward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


class _StereoFrontEnd(torch.nn.Module):
    """TartanVO stand-in: the real VONet's frozen stereo branch through its bf16 execution copy, a per-frame 'pose' from it."""

    def __init__(self):
        super().__init__()
        from islam_amd import nets
        self.vonet = nets.VONet(fix_parts=('flow', 'stereo'))
        self.vonet.set_frozen_dtype(torch.bfloat16)

    def forward(self, sample):
        x = torch.cat([sample['img0_norm'], sample['img0_r_norm']], 1).cuda()
        disp = self.vonet._run_frozen('stereo', self.vonet.stereoNet, torch.bfloat16, x)[0].float()
        return {'motion': torch.cat([disp.mean((1, 2, 3)).reshape(-1, 1)] * 7, 1), 'disp': disp}


def _sample(B, H=256, W=256):
    g = torch.Generator().manual_seed(3)
    gain = (0.4 + 0.8 * torch.arange(B, dtype=torch.float32)).view(B, 1, 1, 1)     # frames of different contrast: per-rank statistics
    return {'img0': torch.rand(B, 3, H, W, generator=g), 'img0_norm': torch.randn(B, 3, H, W, generator=g) * gain,      # would differ
            'img0_r_norm': torch.randn(B, 3, H, W, generator=g) * gain, 'datatype': ['kitti'] * B}


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    from tests.helpers import GLOO_TIMEOUT_S
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=GLOO_TIMEOUT_S))
    torch.cuda.set_device(0)
    from islam_amd import nets
    from islam_amd.dist_train import FrameParallelVO, ShardedBatchNorm2d
    torch.manual_seed(0)
    vo = _StereoFrontEnd().cuda().train()
    fp = FrameParallelVO(vo)
    with torch.no_grad():
        res = fp(_sample(4))
    ex = vo.vonet._exec['stereo'].module()
    n_sharded = sum(isinstance(m, ShardedBatchNorm2d) for m in ex.modules())
    hip_convs = sum(1 for m in ex.modules() if isinstance(m, torch.nn.Conv2d) and '_nhwc_packed' in m.__dict__)
    bn = {k: v.clone().cpu() for k, v in vo.vonet.stereoNet.state_dict().items() if 'running_' in k}
    torch.save({'motion': res['motion'].tensor().cpu(), 'disp': res['disp'].cpu(), 'bn': bn, 'n_sharded': n_sharded, 'hip_convs': hip_convs,
                'level': nets.HIP_CONV_LEVEL}, os.path.join(out_dir, 'f%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_parallel_stereo_copy_uses_global_batch_statistics(cuda, tmp_path):
    from tests.helpers import spawn_ranks
    spawn_ranks(_worker, 2, str(tmp_path))
    outs = [torch.load(os.path.join(str(tmp_path), 'f%d.pt' % r)) for r in range(2)]
    torch.manual_seed(0)
    vo = _StereoFrontEnd().cuda().train()
    with torch.no_grad():
        ref = vo(_sample(4))
    scale = float(ref['disp'].abs().max())
    for r in range(2):
        assert outs[r]['n_sharded'] == 34                                        # the execution copy kept the sharded BatchNorms
        if outs[r]['level'] >= 1:
            assert outs[r]['hip_convs'] > 0                                      # hourglass convolutions still on the HIP kernel
        d = (outs[r]['disp'] - ref['disp'][r::2].cpu()).abs()
        assert float(d.max()) <= 6e-2 * scale                                    # bf16 noise (different kernels), no per-rank statistics
        for k, v in outs[r]['bn'].items():
            want = vo.vonet.stereoNet.state_dict()[k].cpu()
            if k.endswith('running_var'):            # 0.9 + 0.1 * batch variance: bf16 activations, other kernels
                torch.testing.assert_close(v, want, rtol=2e-2, atol=1e-4)
            else:                                    # 0.1 * batch mean of activations that carry ~4e-3 relative bf16 noise
                torch.testing.assert_close(v, want, rtol=2e-2, atol=1e-3)
    torch.testing.assert_close(outs[0]['motion'], outs[1]['motion'], rtol=0, atol=0)
    # what per-rank statistics would have produced is far outside that tolerance: each rank normalising its own two frames
    with torch.no_grad():
        torch.manual_seed(0)
        lone = _StereoFrontEnd().cuda().train()
        half = {k: (v[0::2] if isinstance(v, torch.Tensor) else v[0::2]) for k, v in _sample(4).items()}
        own = lone(half)['disp'].cpu()
    assert float((own - ref['disp'][0::2].cpu()).abs().max()) > 2 * 6e-2 * scale
