"""Deterministic weight fill shared by the golden-vector generator and the tests: every state-dict
entry is filled from a generator seeded by its key, so the reference module (when the fixtures were
made) and the re-implementation (when the tests run) hold identical weights without storing them."""
import hashlib
import math

import torch


def fill_state_dict(module):
    sd = module.state_dict()
    out = {}
    for k, v in sd.items():
        seed = int(hashlib.sha256(k.encode()).hexdigest()[:8], 16)
        g = torch.Generator().manual_seed(seed)
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        r = torch.randn(v.shape, generator=g, dtype=torch.float32)
        if k.endswith('running_var'):
            out[k] = 1.0 + 0.1 * r.abs()
        elif k.endswith('running_mean'):
            out[k] = 0.01 * r
        elif k.endswith('bias') or 'bias_' in k:
            out[k] = 0.01 * r
        elif v.dim() >= 2:
            fan_in = v[0].numel() if v.dim() > 1 else v.numel()
            if 'deconv' in k or 'upfeat' in k:          # ConvTranspose: (in, out, kh, kw)
                fan_in = v.shape[0] * v[0, 0].numel() / 4.0
            out[k] = r * math.sqrt(2.0 / max(fan_in, 1))
        else:                                            # BatchNorm weight
            out[k] = 1.0 + 0.1 * r
    module.load_state_dict(out)
    return module


INPUT_SPECS = {'pwc': ((1, 6, 128, 192), 'rand', 101), 'stereo': ((2, 6, 256, 256), 'randn', 102),
               'pose': ((2, 4, 112, 160), 'randn', 103), 'acc': ((83, 3), 'randn', 104), 'gyro': ((83, 3), 'randn', 105)}


def make_input(name):
    """Seeded test inputs (regenerated identically by the generator script and by the tests)."""
    shape, kind, seed = INPUT_SPECS[name]
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) if kind == 'rand' else torch.randn(shape, generator=g)


VONET_SEED = 106


def vonet_sample():
    """Input of the whole-VONet fixture: one synthetic stereo pair at the production size (the pose head's 256*6 flatten
    needs 448x640 images); band-limited texture, so the edge mask of TartanVO.forward is non-trivial."""
    from islam_amd import synthetic
    return synthetic.stereo_batch(1, seed=VONET_SEED)


def tame_vonet(net):
    """The seeded fill has no normalisation after the hourglass stacks: the disparity head ends up ~4e7 and the pose head
    ~1e3.  Rescale the LAST layer of each so that TartanVO's glue sees plausible magnitudes (disparity ~20 px, pose O(1) in
    pose_std units); applied identically by the generator (to the reference module) and by the tests."""
    with torch.no_grad():
        for name, s in (('stereoNet.conv_c13', 3.6e-8), ('flowPoseNet.voflow_trans.2', 1e-3), ('flowPoseNet.voflow_rot.2', 1e-3)):
            mod = net.get_submodule(name)
            mod.weight.mul_(s)
            mod.bias.mul_(s)
    return net
