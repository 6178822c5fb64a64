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


@pytest.mark.parametrize('B,seed,H,W', [(2, None, 112, 160), (8, 5, 112, 160), (3, 7, 112, 160), (1, 11, 112, 160), (12, 17, 112, 160),
                                        (2, 12, 96, 160), (2, 14, 100, 132)])
def test_forward_and_all_gradients_match_float64_autograd(cuda, B, seed, H, W):
    """B = 8 at 112 x 160 is the benched shape; 96 x 160 and 100 x 132 reach the 2 x 3 feature map through other (odd) intermediate sizes:
    48-24-12-6-3-2 rows, 50-25-13-7-4-2 rows x 66-33-17-9-5-3 columns.  Seeds are fixed and the kernels deterministic, so the figures are the
    same on every box (measured: forward 4e-8 .. 1e-7, gradients 7e-7 .. 1e-6).  What a float64 reference CANNOT hold an fp32 run to is a
    pre-activation within rounding of zero: the two disagree on that ReLU's mask and one term of a gradient sum appears or disappears --
    at B = 16 (57 M activations in the first layers) three of four seeds have such an element and single tensors move by 1e-3 .. 7e-3
    (scripts/debug/pose_head_b16_errors.py; round 5 chased the same effect in MIOpen's atomics).  The upper end of the served range is
    therefore held by the flip-tolerant test below."""
    net, head, ref = _head(cuda, seed, scale_last=1.0 if seed is not None else 1e-3)
    g = torch.Generator().manual_seed(100 + B)
    x = torch.randn(B, 4, H, W, generator=g)
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


def test_largest_batch_against_float64_up_to_relu_flips(cuda):
    """B = 16, the largest batch the entry points serve: forward 1e-5; gradients 1e-4 for all but the few tensors a ReLU flip reaches (see
    the docstring above: three of four seeds have one at this size) -- at most 8 of the 120 may exceed 1e-4 and none 2e-2.  A wrong
    kernel (a mis-indexed tile, a lost K chunk at the larger tile counts / other split factors of this batch) moves every tensor."""
    net, head, ref = _head(cuda, 13)
    g = torch.Generator().manual_seed(116)
    x = torch.randn(16, 4, 112, 160, generator=g)
    gy = torch.randn(16, 6, generator=g)
    yr = ref(x.double())
    yr.backward(gy.double())
    y = head(x.to(cuda).contiguous(memory_format=torch.channels_last))
    assert _rel(y, yr) <= 1e-5
    y.backward(gy.to(cuda))
    errs = sorted((_rel(p.grad, pr.grad), name) for (name, p), pr in zip(net.named_parameters(), ref.parameters()))
    assert errs[-1][0] <= 2e-2, errs[-4:]
    assert sum(e > 1e-4 for e, _ in errs) <= 8, errs[-10:]
    assert errs[len(errs) // 2][0] <= 5e-6, errs[len(errs) // 2]               # the typical tensor is at fp32 round-off


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
