"""The hand-written pose head (csrc/pose_head.hip behind islam_pose_head_forward / _backward; reference Network/VOFlowNet.py:20-39,110-157,
185-194 and the backward train.py:280-283 runs through it) against

  * the reference-generated golden vector tests/golden/nets_pose.npz,
  * a float64 CPU run of the same torch modules (forward and every one of the 120 parameter gradients).

Tolerances: forward 1e-5, gradients 1e-4, relative to the largest magnitude of the compared tensor -- the
kernels are exact fp32 (v_mfma_f32_32x32x2_f32 = an fmaf chain) and sum in a fixed order: two runs are bit-identical."""
import copy
import os

import numpy as np
import pytest
import torch

from tests.golden.netfill import fill_state_dict, make_input

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def _head(cuda, seed=None, scale_last=1.0):
    from islam_amd import nets, pose_head
    torch.manual_seed(0 if seed is None else seed)
    net = nets.VOFlowRes()
    if seed is None:
        fill_state_dict(net)
    if scale_last != 1.0:
        with torch.no_grad():
            for m in (net.voflow_trans[2], net.voflow_rot[2]):
                m.weight.mul_(scale_last)
                m.bias.mul_(scale_last)
    ref = copy.deepcopy(net).double()                           # float64 CPU reference of the same parameters
    net = net.to(cuda).to(memory_format=torch.channels_last)
    return net, pose_head.PoseHeadHip(net), ref


def _rel(got, ref):
    ref = ref.detach().double().cpu()
    return float((got.detach().double().cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


def test_forward_matches_the_reference_generated_vector(cuda):
    net, head, _ = _head(cuda)
    x = make_input('pose').to(cuda).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        out = head(x)
    ref = np.load(os.path.join(G, 'nets_pose.npz'))['pose']
    assert out.shape == (2, 6)
    assert float(np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()) <= 2e-5      # (the fixture itself is an fp32 run)


@pytest.mark.parametrize('B,seed', [(2, None), (8, 5), (3, 7)])
def test_forward_and_all_gradients_match_float64_autograd(cuda, B, seed):
    net, head, ref = _head(cuda, seed, scale_last=1.0 if seed is not None else 1e-3)
    g = torch.Generator().manual_seed(100 + B)
    x = torch.randn(B, 4, 112, 160, generator=g)
    gy = torch.randn(B, 6, generator=g)
    yr = ref(x.double())
    yr.backward(gy.double())
    xd = x.to(cuda).contiguous(memory_format=torch.channels_last)
    y = head(xd)
    assert y.requires_grad
    assert _rel(y, yr) <= 1e-5
    y.backward(gy.to(cuda))
    worst = ('', 0.0)
    for (name, p), pr in zip(net.named_parameters(), ref.parameters()):
        assert p.grad is not None and p.grad.shape == p.shape, name
        e = _rel(p.grad, pr.grad)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] <= 1e-4, worst
    # a second forward + backward ACCUMULATES (train.py:280-283: .backward() per batch, optimizer.step() per epoch) -- and is bit-identical
    first = [p.grad.clone() for p in net.parameters()]
    y2 = head(xd)
    assert torch.equal(y2, y)
    y2.backward(gy.to(cuda))
    for p, f in zip(net.parameters(), first):
        assert torch.equal(p.grad, f + f)


def test_graph_replay_equals_direct_calls(cuda):
    from islam_amd import pose_head
    net, head, _ = _head(cuda, seed=3)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(8, 4, 112, 160, generator=g).to(cuda).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(8, 6, generator=g).to(cuda)
    y = head(x)
    y.backward(gy)
    want = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    hg = pose_head.PoseHeadHip(net, graphs=True)
    for rep in range(2):
        yg = hg(x)
        assert torch.equal(yg, y)
        yg.backward(gy)
    for p, w in zip(net.parameters(), want):
        assert torch.equal(p.grad, w + w)


def test_backward_of_a_stale_forward_raises(cuda):
    net, head, _ = _head(cuda, seed=4)
    x = torch.randn(2, 4, 112, 160).to(cuda).contiguous(memory_format=torch.channels_last)
    y1 = head(x)
    head(x)
    with pytest.raises(RuntimeError, match='overwritten'):
        y1.sum().backward()


def test_unsupported_inputs_are_refused(cuda):
    from islam_amd import pose_head
    net, head, _ = _head(cuda, seed=4)
    assert not pose_head.supported(net, torch.randn(2, 4, 112, 160))                                            # CPU tensor
    assert not pose_head.supported(net, torch.randn(2, 4, 112, 160, device=cuda))                               # not channels-last
    assert not pose_head.supported(net, torch.randn(2, 4, 64, 160, device=cuda).contiguous(memory_format=torch.channels_last))   # a 2x3 feature map is needed
    assert pose_head.supported(net, torch.randn(2, 4, 112, 160, device=cuda).contiguous(memory_format=torch.channels_last))
