"""GPU parity of the front-end kernels (correlation, warp, scale recovery) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import cwrap, scale as oscale

pytestmark = pytest.mark.gpu

# the five (C,H,W) shapes PWC-Net calls the correlation with at 448x640 + ragged edge cases
PWC_SHAPES = [(196, 7, 10), (128, 14, 20), (96, 28, 40), (64, 56, 80), (32, 112, 160)]
EDGE_SHAPES = [(1, 1, 1), (3, 5, 7), (17, 9, 33), (33, 4, 65), (5, 13, 31)]
# the four-pixel kernel (W % 4 == 0, H W >= 1024) away from its whole 32 x 7 tiles: last band of 3 / 1 / 2 rows, partial tile columns,
# a channel count that is not a whole chunk
RAGGED4_SHAPES = [(20, 45, 64), (8, 36, 36), (16, 100, 132), (40, 30, 44)]


@pytest.mark.parametrize('C,H,W', PWC_SHAPES + EDGE_SHAPES + RAGGED4_SHAPES)
def test_corr81_forward(cuda, C, H, W):
    from islam_amd import ops
    B = 2
    g = torch.Generator().manual_seed(C * 1000 + H)
    f1 = torch.randn(B, C, H, W, generator=g)
    f2 = torch.randn(B, C, H, W, generator=g)
    out = ops.corr81_forward(f1.to(cuda), f2.to(cuda)).cpu().numpy()
    ref = cwrap.corr81_fwd(f1.numpy(), f2.numpy())
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-6)


def test_corr81_forward_full_batch(cuda):
    """BASELINE size: B=8 at the largest level, against the oracle on a slice + linearity property."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn(8, 32, 112, 160, generator=g).to(cuda)
    f2 = torch.randn(8, 32, 112, 160, generator=g).to(cuda)
    out = ops.corr81_forward(f1, f2)
    ref = cwrap.corr81_fwd(f1[3:4].cpu().numpy(), f2[3:4].cpu().numpy())
    np.testing.assert_allclose(out[3:4].cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
    out2 = ops.corr81_forward(2.0 * f1, f2)                     # linear in each argument
    torch.testing.assert_close(out2, 2.0 * out, rtol=1e-6, atol=1e-6)
    # centre channel (dy=dx=0) is the per-pixel mean product
    torch.testing.assert_close(out[:, 40], (f1 * f2).mean(1), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('C,H,W', [(64, 56, 80), (96, 28, 40), (32, 112, 160)])
def test_corr81_forward_at_the_batch_of_the_bench(cuda, C, H, W):
    """B = 8: the slice plan (channel slices + reduction through the scratch buffer) depends on the number of tiles, i.e. on B and on
    the kernel variant -- the scratch a caller sizes and the scratch the launch uses must follow the same plan."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(C + H)
    f1 = torch.randn(8, C, H, W, generator=g).to(cuda)
    f2 = torch.randn(8, C, H, W, generator=g).to(cuda)
    out = ops.corr81_forward(f1, f2)
    for b in (0, 7):
        ref = cwrap.corr81_fwd(f1[b:b + 1].cpu().numpy(), f2[b:b + 1].cpu().numpy())
        np.testing.assert_allclose(out[b:b + 1].cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
    buf = torch.full((8, 81 + 7, H, W), 3.0, device=cuda)
    ops.corr81_act(f1, f2, buf, 4, 0.1)                          # into a channel slice, LeakyReLU applied
    torch.testing.assert_close(buf[:, 4:85], torch.nn.functional.leaky_relu(out, 0.1), rtol=1e-6, atol=1e-7)
    assert float((buf[:, :4] - 3.0).abs().max()) == 0.0 and float((buf[:, 85:] - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize('C,H,W', [(8, 14, 20), (5, 9, 33), (32, 28, 40), (3, 8, 32)])
def test_corr81_backward(cuda, C, H, W):
    from islam_amd import ops
    B = 2
    g = torch.Generator().manual_seed(C + W)
    f1 = torch.randn(B, C, H, W, generator=g)
    f2 = torch.randn(B, C, H, W, generator=g)
    go = torch.randn(B, 81, H, W, generator=g)
    a = f1.to(cuda).requires_grad_(True)
    b = f2.to(cuda).requires_grad_(True)
    ops.FunctionCorrelation(a, b).backward(go.to(cuda))
    r1, r2 = cwrap.corr81_bwd(f1.numpy(), f2.numpy(), go.numpy())
    np.testing.assert_allclose(a.grad.cpu().numpy(), r1, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), r2, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('C,H,W,scale', [(128, 14, 20, 0.625), (96, 28, 40, 1.25), (64, 56, 80, 2.5), (32, 112, 160, 5.0),
                                         (3, 1, 1, 1.0), (2, 5, 1, 1.0), (4, 7, 9, 1.0)])
def test_warp_mask(cuda, C, H, W, scale):
    from islam_amd import ops
    B = 2
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(B, C, H, W, generator=g)
    flo = torch.randn(B, 2, H, W, generator=g) * (3.0 / scale)
    flo[0, :, 0, 0] = 0.0                                        # exact-integer sample position
    out = ops.warp_mask(x.to(cuda), flo.to(cuda), scale).cpu().numpy()
    ref = cwrap.warp(x.numpy(), (flo * scale).numpy())
    # bit-exact coordinates; the 4-tap sum is evaluated in the same order -> exact up to fma-free rounding
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)
    # and against torch's own grid_sample (what the reference calls), run on the GPU
    xx = torch.arange(0, W).view(1, -1).repeat(H, 1)
    yy = torch.arange(0, H).view(-1, 1).repeat(1, W)
    grid = torch.cat((xx.view(1, 1, H, W).repeat(B, 1, 1, 1), yy.view(1, 1, H, W).repeat(B, 1, 1, 1)), 1).float()
    vg = grid + flo * scale
    vg[:, 0] = 2.0 * vg[:, 0].clone() / max(W - 1, 1) - 1.0
    vg[:, 1] = 2.0 * vg[:, 1].clone() / max(H - 1, 1) - 1.0
    vg = vg.permute(0, 2, 3, 1)
    o = torch.nn.functional.grid_sample(x, vg, align_corners=True)
    m = torch.nn.functional.grid_sample(torch.ones_like(x), vg, align_corners=True)
    m[m < 0.9999] = 0
    m[m > 0] = 1
    np.testing.assert_allclose(out, (o * m).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('with_edge', [False, True])
def test_scale_ls(cuda, with_edge):
    from islam_amd import ops
    B, H, W = 3, 112, 160
    rng = np.random.default_rng(3)
    disp = rng.uniform(0.0, 12.0, (B, 1, H, W)).astype(np.float32)
    flow = rng.normal(0, 3.0, (B, 2, H, W)).astype(np.float32)
    flow[:, :, :4, :4] = 0.0                                    # zero-flow pixels are masked out
    q = rng.normal(size=(B, 4)) * 0.05
    q[:, 3] = 1.0
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    pose = np.concatenate([rng.normal(size=(B, 3)), q], 1).astype(np.float32)
    intr = np.tile(np.array([[214.7, 214.7, 130.0, 55.3]], np.float32), (B, 1))
    base = np.array([0.54, 0.25, 0.11], np.float32)
    th = np.array([5.0, 1.0, 1.0], np.float32)
    edge = (rng.uniform(size=(B, H, W)) > 0.6) if with_edge else None
    t = lambda a: torch.tensor(a, device=cuda)
    s, z, m, dm, sums = ops.scale_ls(t(disp), t(flow), t(pose), t(intr), t(base), t(edge) if with_edge else None, t(th))
    for b in range(B):
        so, zo, mo, dmo, (MM, Mw) = oscale.scale_from_disp_flow(disp[b], flow[b], pose[b], *intr[b], base[b],
                                                               edge[b] if with_edge else None, th[b])
        assert (m[b].cpu().numpy() != mo).mean() < 1e-4          # float32 boundary ties only
        np.testing.assert_array_equal(dm[b].cpu().numpy(), dmo)
        np.testing.assert_allclose(z[b].cpu().numpy(), zo, rtol=1e-6)
        np.testing.assert_allclose(sums[b, 0].item(), MM, rtol=1e-4)
        np.testing.assert_allclose(sums[b, 1].item(), Mw, rtol=1e-4, atol=1e-3 * abs(MM))
        np.testing.assert_allclose(s[b].item(), so, rtol=2e-4, atol=1e-6)
    assert sums[:, 17].cpu().numpy().tolist() == m.reshape(B, -1).sum(1).cpu().numpy().tolist()


def test_scale_ls_depth_input(cuda):
    """The depth= branch of scale_from_disp_flow (dense_ba.py:125-131) through islam_scale_ls_depth and the dense_ba surface."""
    from islam_amd import dense_ba, ops
    B, H, W = 2, 96, 128
    rng = np.random.default_rng(11)
    depth = rng.uniform(-1.0, 160.0, (B, 1, H, W)).astype(np.float32)       # negatives, zeros and beyond fx*baseline are masked
    depth[:, :, :3, :3] = 0.0
    flow = rng.normal(0, 3.0, (B, 2, H, W)).astype(np.float32)
    q = rng.normal(size=(B, 4)) * 0.05
    q[:, 3] = 1.0
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    pose = np.concatenate([rng.normal(size=(B, 3)), q], 1).astype(np.float32)
    intr = np.tile(np.array([[214.7, 214.7, 64.0, 48.0]], np.float32), (B, 1))
    base = np.array([0.54, 0.25], np.float32)
    t = lambda a: torch.tensor(a, device=cuda)
    s, z, m, dm, sums = ops.scale_ls(t(depth), t(flow), t(pose), t(intr), t(base), None, None, depth_input=True)
    for b in range(B):
        so, zo, mo, dmo, (MM, Mw) = oscale.scale_from_disp_flow(None, flow[b], pose[b], *intr[b], base[b], None, depth=depth[b])
        assert 0.1 < dmo.mean() < 0.95
        assert (m[b].cpu().numpy() != mo).mean() < 1e-4
        np.testing.assert_array_equal(dm[b].cpu().numpy(), dmo)
        np.testing.assert_array_equal(z[b].cpu().numpy(), zo)
        np.testing.assert_allclose(sums[b, 0].item(), MM, rtol=1e-4)
        np.testing.assert_allclose(s[b].item(), so, rtol=2e-4, atol=1e-6)
    s1, z1, m1, dm1 = dense_ba.scale_from_disp_flow(None, t(flow[0]), t(pose[0]), *[float(v) for v in intr[0]], float(base[0]),
                                                    depth=t(depth[0, 0]))
    np.testing.assert_allclose(s1.item(), s[0].item(), rtol=1e-6)
    assert torch.equal(z1, z[0]) and torch.equal(dm1, dm[0])


def test_scale_ls_empty_mask_is_nan(cuda):
    from islam_amd import ops
    B, H, W = 1, 16, 16
    z = torch.zeros
    s, *_ = ops.scale_ls(z(B, 1, H, W, device=cuda), z(B, 2, H, W, device=cuda),
                         torch.tensor([[0., 0, 1, 0, 0, 0, 1]], device=cuda), torch.tensor([[100., 100, 8, 8]], device=cuda),
                         torch.tensor([0.5], device=cuda), None, torch.tensor([1.0], device=cuda))
    assert torch.isnan(s).all()                                 # reference: s = 0/0


@pytest.mark.parametrize('C,H,W,scale', [(20, 14, 20, 0.625), (3, 9, 33, 2.5), (32, 28, 40, 5.0), (2, 5, 1, 1.0)])
def test_warp_backward_matches_torch_autograd(cuda, C, H, W, scale):
    """islam_warp_mask_bwd vs autograd through torch's grid_sample formulation of PWCDCNet.warp (PWCNet.py:170-206)."""
    from islam_amd import ops
    B = 2
    g = torch.Generator().manual_seed(C * 31 + W)
    x0 = torch.randn(B, C, H, W, generator=g)
    f0 = torch.randn(B, 2, H, W, generator=g) * (2.0 / scale)
    go = torch.randn(B, C, H, W, generator=g).to(cuda)
    x, fl = x0.to(cuda).requires_grad_(True), f0.to(cuda).requires_grad_(True)
    ops.warp(x, fl, scale).backward(go)
    xr, fr = x0.to(cuda).requires_grad_(True), f0.to(cuda).requires_grad_(True)
    xx = torch.arange(0, W, device=cuda).view(1, -1).repeat(H, 1)
    yy = torch.arange(0, H, device=cuda).view(-1, 1).repeat(1, W)
    grid = torch.cat((xx.view(1, 1, H, W).repeat(B, 1, 1, 1), yy.view(1, 1, H, W).repeat(B, 1, 1, 1)), 1).float()
    vg = grid + fr * scale
    vgx = 2.0 * vg[:, 0] / max(W - 1, 1) - 1.0
    vgy = 2.0 * vg[:, 1] / max(H - 1, 1) - 1.0
    vgrid = torch.stack((vgx, vgy), -1)
    out = torch.nn.functional.grid_sample(xr, vgrid, align_corners=True)
    m = torch.nn.functional.grid_sample(torch.ones_like(xr), vgrid, align_corners=True).detach()
    m = (m >= 0.9999).float()
    (out * m).backward(go)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(fl.grad, fr.grad, rtol=1e-3, atol=1e-4)


def test_frozen_stereo_net_bf16_execution_copy(cuda):
    """BASELINE config 2 ("bf16 nets"): the frozen stereo net runs through a bf16 channels_last execution copy whose weights
    are cast once; the fp32 master keeps the checkpoint, train-mode BatchNorm statistics still land in it, and a
    weight change (load_state_dict) is picked up."""
    from islam_amd import nets
    torch.manual_seed(0)
    net = nets.VONet(fix_parts=('flow', 'stereo')).to(cuda).train()
    x = torch.randn(2, 6, 256, 320, device=cuda)
    run = lambda: net._run_frozen('stereo', net.stereoNet, net.frozen_dtype, x)[0].float()
    with torch.no_grad():
        d0 = run()
    rm0 = net.stereoNet.state_dict()['feature_extraction.firstconv.0.1.running_mean'].clone()
    net.set_frozen_dtype(torch.bfloat16)
    with torch.no_grad():
        d1 = run()
    assert all(v.dtype == torch.float32 for v in net.state_dict().values() if v.is_floating_point())
    assert float((d1 - d0).abs().max()) <= 0.05 * float(d0.abs().max()) + 1e-3
    rm1 = net.stereoNet.state_dict()['feature_extraction.firstconv.0.1.running_mean']
    assert not torch.equal(rm1, rm0)                                     # statistics updated through the shared buffers
    copy_a = net._exec['stereo'].module()
    assert copy_a is net._exec['stereo'].module()                        # cached
    sd = {k: (v * 0.5 if k.endswith('conv_c13.weight') else v) for k, v in net.stereoNet.state_dict().items()}
    net.stereoNet.load_state_dict(sd)
    assert net._exec['stereo'].module() is not copy_a                    # re-cast after the master changed


def test_flow_net_on_the_matrix_core_convolution(cuda):
    """PWCDCNet.forward_mfma (HIP implicit-GEMM convolutions, slice-written DenseNet blocks, batched pyramid) against the
    plain fp32 forward of the same weights: all five flow outputs agree to bf16-operand accuracy."""
    from islam_amd import nets
    torch.manual_seed(0)
    net = nets.PWCDCNet().to(cuda).eval()
    x = torch.rand(2, 6, 192, 256, device=cuda)
    with torch.no_grad():
        ref, _ = net(x)
        got, _ = net.forward_mfma(x)
    for r, g in zip(ref, got):
        assert r.shape == g.shape
        assert float((r - g).abs().max()) < 3e-2 * float(r.abs().max()) + 1e-3
    # the weights are re-packed when the fp32 master changes
    with torch.no_grad():
        net.conv2_0[0].weight.mul_(0.5)
        ref2, _ = net(x)
        got2, _ = net.forward_mfma(x)
    assert float((ref2[0] - got2[0]).abs().max()) < 3e-2 * float(ref2[0].abs().max()) + 1e-3
    assert float((ref2[0] - ref[0]).abs().max()) > 1e-3 * float(ref[0].abs().max())


def test_pose_head_channels_last_is_the_same_function(cuda):
    """VONet.set_pose_channels_last: same parameters, same pose and same gradients as the NCHW pose head (MIOpen may pick
    other kernels, hence a tolerance)."""
    from islam_amd import nets
    torch.manual_seed(5)
    vn = nets.VONet(fix_parts=('flow', 'stereo'))
    head = vn.flowPoseNet.to(cuda)
    B = 2
    flow = torch.randn(B, 2, 112, 160, device=cuda)
    intr = torch.rand(B, 2, 112, 160, device=cuda)
    frozen = (flow, torch.zeros(B, 1, 112, 160, device=cuda))

    def run():
        for p in head.parameters():
            p.grad = None
        _, _, pose = vn(None, None, None, None, intr, frozen=frozen)
        (pose * torch.arange(1, 7, device=cuda)).sum().backward()
        return pose.detach().clone(), [p.grad.detach().clone() for p in head.parameters()]
    p0, g0 = run()
    keys0 = list(head.state_dict().keys())
    vn.set_pose_channels_last(True)
    p1, g1 = run()
    assert list(head.state_dict().keys()) == keys0
    assert any(p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous() for p in head.parameters())
    torch.testing.assert_close(p1, p0, rtol=1e-3, atol=1e-5)
    for a, b in zip(g1, g0):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=5e-3, atol=2e-3 * float(b.abs().max()) + 1e-8)   # fp32 sums in another order
    vn.set_pose_channels_last(False)
    p2, _ = run()
    torch.testing.assert_close(p2, p0, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize('mode', [True, 'accumulate'])
def test_pose_head_graph_replay_trains_like_eager(cuda, mode):
    """VONet.graph_pose: forward and backward of the trainable pose head replayed from HIP graphs give the eager pose and
    parameter gradients, keep following the optimizer's in-place updates, and leave other batch shapes on the eager path.
    'accumulate': the backward node adds the gradients to .grad itself (nets._PoseGraph) -- also on top of gradients that are
    already there (second backward without zero_grad)."""
    from islam_amd import nets
    torch.manual_seed(7)
    B = 2
    mk = lambda: nets.VONet(fix_parts=('flow', 'stereo'))
    va, vb = mk(), mk()
    vb.load_state_dict(va.state_dict())
    for v in (va, vb):
        v.flowPoseNet.to(cuda).train()
    vb.graph_pose = mode
    opt = [torch.optim.SGD(v.flowPoseNet.parameters(), lr=1e-3) for v in (va, vb)]
    wts = torch.arange(1, 7, device=cuda, dtype=torch.float32)
    for it in range(3):
        flow = torch.randn(B, 2, 112, 160, device=cuda)
        intr = torch.rand(B, 2, 112, 160, device=cuda)
        frozen = (flow, torch.zeros(B, 1, 112, 160, device=cuda))
        poses, grads = [], []
        for v, o in zip((va, vb), opt):
            o.zero_grad(set_to_none=True)
            _, _, pose = v(None, None, None, None, intr, frozen=frozen)
            (pose * wts).sum().backward()
            if it == 1:                                          # accumulation onto existing gradients
                _, _, pose2 = v(None, None, None, None, intr, frozen=frozen)
                (pose2 * wts).sum().backward()
            poses.append(pose.detach().clone())
            grads.append([p.grad.detach().clone() for p in v.flowPoseNet.parameters()])
            o.step()
        torch.testing.assert_close(poses[1], poses[0], rtol=1e-3, atol=1e-5)
        for a, b in zip(grads[1], grads[0]):
            torch.testing.assert_close(a, b, rtol=5e-3, atol=2e-3 * float(b.abs().max()) + 1e-8)
    assert vb._pose_graphed == (B, 4, 112, 160)
    # another batch size: eager path, same module, same parameters
    flow = torch.randn(1, 2, 112, 160, device=cuda)
    intr = torch.rand(1, 2, 112, 160, device=cuda)
    out = [v(None, None, None, None, intr, frozen=(flow, None))[2] for v in (va, vb)]
    torch.testing.assert_close(out[1], out[0], rtol=1e-3, atol=1e-5)
    assert list(va.state_dict().keys()) == list(vb.state_dict().keys())
    # destroy the captured graphs now, with the device idle (see test_graph_replay_of_the_frozen_forward)
    torch.cuda.synchronize()
    del va, vb, opt, out, poses, grads, pose
    import gc
    gc.collect()
    torch.cuda.synchronize()


def test_pose_graph_forward_without_backward_is_harmless(cuda):
    """graph_pose='accumulate' (nets._PoseGraphFn): a forward that never gets a backward (metrics pass, an exception between the VO
    forward and the gradient step) does not block later forwards -- forward, forward, backward-of-the-SECOND gives the eager
    gradients; only the backward of a forward whose activations a later replay overwrote raises (train.py:212-283 has no such
    restriction: eager autograd keeps every forward's activations)."""
    from islam_amd import nets
    torch.manual_seed(11)
    B = 2
    va, vb = nets.VONet(fix_parts=('flow', 'stereo')), nets.VONet(fix_parts=('flow', 'stereo'))
    vb.load_state_dict(va.state_dict())
    for v in (va, vb):
        v.flowPoseNet.to(cuda).train()
    vb.graph_pose = 'accumulate'
    wts = torch.arange(1, 7, device=cuda, dtype=torch.float32)
    mk = lambda: (torch.randn(B, 2, 112, 160, device=cuda), torch.rand(B, 2, 112, 160, device=cuda))
    (f1, i1), (f2, i2) = mk(), mk()
    z = torch.zeros(B, 1, 112, 160, device=cuda)
    _, _, p1 = vb(None, None, None, None, i1, frozen=(f1, z))            # never back-propagated
    _, _, p2 = vb(None, None, None, None, i2, frozen=(f2, z))
    (p2 * wts).sum().backward()
    _, _, q2 = va(None, None, None, None, i2, frozen=(f2, z))
    (q2 * wts).sum().backward()
    torch.testing.assert_close(p2, q2, rtol=1e-3, atol=1e-5)
    for a, b in zip(vb.flowPoseNet.parameters(), va.flowPoseNet.parameters()):
        torch.testing.assert_close(a.grad, b.grad, rtol=5e-3, atol=2e-3 * float(b.grad.abs().max()) + 1e-8)
    with pytest.raises(RuntimeError, match='overwritten'):
        (p1 * wts).sum().backward()                                      # its activations are gone: must fail loudly, not silently
    torch.cuda.synchronize()
    del va, vb, p1, p2, q2
    import gc
    gc.collect()
    torch.cuda.synchronize()


def test_pose_head_fused_elementwise_tail(cuda):
    """VOFlowRes.set_fused_tail (islam_bias_act_f32_nhwc / _bwd): bias + ReLU (+ shortcut) behind every encoder convolution in one
    launch each way.  Same convolutions on the same inputs: the forward is the same sequence of fp32 additions (bit-equal pose), the
    bias gradients are summed in another (fixed) order."""
    from islam_amd import nets, ops
    torch.manual_seed(7)
    vn = nets.VONet(fix_parts=('flow', 'stereo'))
    head = vn.flowPoseNet.to(cuda)
    vn.set_pose_channels_last(True)
    assert head.fused_tail
    B = 3
    flow = torch.randn(B, 2, 112, 160, device=cuda)
    intr = torch.rand(B, 2, 112, 160, device=cuda)
    frozen = (flow, torch.zeros(B, 1, 112, 160, device=cuda))

    def run():
        for p in head.parameters():
            p.grad = None
        _, _, pose = vn(None, None, None, None, intr, frozen=frozen)
        (pose * torch.arange(1, 7, device=cuda)).sum().backward()
        return pose.detach().clone(), [p.grad.detach().clone() for p in head.parameters()]
    p1, g1 = run()
    head.set_fused_tail(False)
    p0, g0 = run()
    torch.testing.assert_close(p1, p0, rtol=1e-6, atol=1e-7)
    for (name, _), a, b in zip(head.named_parameters(), g1, g0):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=1e-6 * max(float(b.abs().max()), 1e-30) + 1e-12, msg=name)
    # the op on its own: values, gradients w.r.t. all three inputs, odd pixel counts
    for shape, with_res, relu in (((2, 32, 5, 7), False, True), ((3, 256, 2, 3), True, True), ((1, 64, 9, 11), True, False)):
        g = torch.Generator(device='cpu').manual_seed(sum(shape))
        mk = lambda *s: torch.randn(*s, generator=g).to(cuda)
        x = mk(*shape).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bias = mk(shape[1]).requires_grad_(True)
        res = mk(*shape).contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
        w = mk(*shape)
        y = ops.bias_act(x, bias, res, relu)
        ref = x + bias.view(1, -1, 1, 1) + (res if with_res else 0)
        ref = torch.relu(ref) if relu else ref
        assert torch.equal(y, ref)
        gi = torch.autograd.grad((y * w).sum(), [x, bias] + ([res] if with_res else []))
        gr = torch.autograd.grad((ref * w).sum(), [x, bias] + ([res] if with_res else []))
        assert torch.equal(gi[0], gr[0]) and (not with_res or torch.equal(gi[2], gr[2]))
        torch.testing.assert_close(gi[1], gr[1], rtol=1e-5, atol=1e-5)
        gi2 = torch.autograd.grad((ops.bias_act(x, bias, res, relu) * w).sum(), [bias])
        assert torch.equal(gi[1], gi2[0])                                   # fixed-order sums: the same bits every time


@pytest.mark.parametrize('B,C,H,W', [(2, 2, 7, 10), (1, 565, 14, 20), (3, 149, 9, 67), (1, 5, 1, 1), (2, 64, 33, 130)])
def test_two_channel_transposed_convolution(cuda, B, C, H, W):
    """islam_deconv4x4s2_to2_f32 (PWC-Net's deconv / upfeat: ConvTranspose2d(C, 2, 4, 2, 1)) against torch in fp32, also into a channel
    slice of a larger tensor, and bit-reproducible."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, C, H, W, generator=g).to(cuda)
    w = (torch.randn(C, 2, 4, 4, generator=g) / (4 * C) ** 0.5).to(cuda)
    b = torch.randn(2, generator=g).to(cuda)
    want = torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    got = ops.deconv_to2(x, w, b)
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) <= 2e-6 * max(float(want.abs().max()), 1.0)
    buf = torch.full((B, 5, 2 * H, 2 * W), 3.0, device=cuda)
    ops.deconv_to2(x, w, None, out=buf, coff=2)
    assert torch.equal(buf[:, 2:4], ops.deconv_to2(x, w, None)) and bool((buf[:, :2] == 3.0).all()) and bool((buf[:, 4:] == 3.0).all())
    assert torch.equal(got, ops.deconv_to2(x, w, b))


@pytest.mark.parametrize('Cin,C,B,H,W', [(3, 16, 2, 64, 128), (3, 16, 1, 37, 71), (3, 16, 1, 2, 2), (16, 32, 2, 32, 64), (16, 32, 1, 45, 39),
                                         (16, 32, 1, 3, 130)])
def test_fused_pyramid_level(cuda, Cin, C, B, H, W):
    """islam_flow_pyramid_level (conv s2 + conv + conv, each + bias + LeakyReLU(0.1), intermediates in LDS) against the three torch
    convolutions evaluated in float64 on the SAME bf16-rounded operands (weights, input, and each intermediate activation rounded to
    bf16 after the fp32 activation) -- what is left is the fp32 summation order and roundings that fall on the other side of a bf16 tie;
    ragged sizes cover partial tiles and the zero padding of the intermediate tiles at the image border."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(100 * Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(cuda)
    ws, bs = [], []
    for cin in (Cin, C, C):
        ws.append((torch.randn(C, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(cuda))
        bs.append((0.1 * torch.randn(C, generator=g)).to(cuda))
    r = lambda t: t.to(torch.bfloat16).double()
    want = x
    for i, (w, b) in enumerate(zip(ws, bs)):
        want = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(r(want), r(w), b.double(), stride=2 if i == 0 else 1, padding=1), 0.1).float()
    got = ops.flow_pyramid_level(x, [ops.pack_pyramid_weight(w) for w in ws], bs, 0.1)
    assert got.shape == want.shape
    err = (got.double() - want.double()).abs()
    scale = float(want.abs().max())
    # a bf16 tie that rounds the other way moves one operand by 2^-8 relative: bounded by ~1e-2 of the output scale, rare
    assert float(err.max()) <= 2e-2 * scale, float(err.max()) / scale
    assert float((err > 1e-4 * scale).double().mean()) < 0.02, float((err > 1e-4 * scale).double().mean())
    assert torch.equal(got, ops.flow_pyramid_level(x, [ops.pack_pyramid_weight(w) for w in ws], bs, 0.1))
    if Cin == 3:              # the pair form: both frames of a (B, 6, H, W) tensor as one batch, first frames first, bit for bit
        x2 = torch.randn(B, 3, H, W, generator=g).to(cuda)
        packed = [ops.pack_pyramid_weight(w) for w in ws]
        pair = ops.flow_pyramid_level_pair(torch.cat((x, x2), 1).contiguous(), packed, bs, 0.1)
        assert torch.equal(pair, ops.flow_pyramid_level(torch.cat((x, x2), 0).contiguous(), packed, bs, 0.1))


@pytest.mark.parametrize('B,C,H,W,up', [(2, 529, 7, 10, True), (1, 149, 9, 67, True), (1, 33, 1, 1, True), (2, 565, 14, 20, False), (1, 5, 33, 130, False),
                                        (1, 37, 66, 128, True), (2, 21, 112, 160, False), (1, 9, 127, 68, True)])   # (large maps: four pixels per lane)
def test_flow_head_with_upsampled_features_in_one_pass(cuda, B, C, H, W, up):
    """islam_flow_head_up_f32: Conv2d(C, 2, 3, 1, 1) and ConvTranspose2d(C, 2, 4, 2, 1) of the same tensor against torch in float64."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g).to(cuda)
    wh = (torch.randn(2, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(cuda)
    bh = torch.randn(2, generator=g).to(cuda)
    wu = (torch.randn(C, 2, 4, 4, generator=g) / (4 * C) ** 0.5).to(cuda)
    bu = torch.randn(2, generator=g).to(cuda)
    flow, upf = ops.flow_head_up(x, wh.permute(1, 0, 2, 3).contiguous(), bh, wu if up else None, bu if up else None)
    want = torch.nn.functional.conv2d(x.double(), wh.double(), bh.double(), padding=1)
    assert flow.shape == want.shape
    assert float((flow.double() - want).abs().max()) <= 2e-6 * max(float(want.abs().max()), 1.0)
    if up:
        wantu = torch.nn.functional.conv_transpose2d(x.double(), wu.double(), bu.double(), stride=2, padding=1)
        assert float((upf.double() - wantu).abs().max()) <= 2e-6 * max(float(wantu.abs().max()), 1.0)
        assert torch.equal(upf, ops.deconv_to2(x, wu, bu))          # the same sums as the kernel without the head
    else:
        assert upf is None
    again = ops.flow_head_up(x, wh.permute(1, 0, 2, 3).contiguous(), bh, wu if up else None, bu if up else None)
    assert torch.equal(flow, again[0])


@pytest.mark.parametrize('B,C,H,W', [(2, 196, 7, 10), (1, 32, 33, 70), (2, 16, 112, 160)])
def test_correlation_with_activation_into_a_channel_slice(cuda, B, C, H, W):
    """islam_corr81_fwd_act == LeakyReLU(islam_corr81_fwd) bit for bit, in its channel slice, both with one and with several channel
    slices of the reduction; the rest of the destination is untouched."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(C)
    a, b = torch.randn(B, C, H, W, generator=g).to(cuda), torch.randn(B, C, H, W, generator=g).to(cuda)
    want = torch.nn.functional.leaky_relu(ops.corr81_forward(a, b), 0.1)
    buf = torch.full((B, 81 + 7, H, W), 3.0, device=cuda)
    ops.corr81_act(a, b, buf, 5, 0.1)
    assert torch.equal(buf[:, 5:86], want)
    assert bool((buf[:, :5] == 3.0).all()) and bool((buf[:, 86:] == 3.0).all())


@pytest.mark.parametrize('H,W,up', [(14, 20, True), (9, 23, True), (28, 40, False)])
def test_flow_head_and_upsampled_features_from_the_mirror(cuda, H, W, up):
    """PWCDCNet._head_up_mirror (predict_flow + upfeat as ONE 3x3 matrix-core convolution of the bf16 channels-last mirror, the
    transposed convolution's four parity classes as output channels + pixel shuffle) against torch's Conv2d / ConvTranspose2d in float64
    on the same bf16 operands, also into a channel slice of a larger buffer."""
    from islam_amd import nets
    torch.manual_seed(3)
    net = nets.PWCDCNet().to(cuda)
    l = 5
    head, dc = net.predict_flow5, net.upfeat5
    with torch.no_grad():
        for p in list(head.parameters()) + list(dc.parameters()):
            p.copy_(torch.randn_like(p) * (0.02 if p.dim() > 1 else 0.5))
    C = head.weight.shape[1]
    Cp = (C + 7) // 8 * 8
    B = 2
    mir = torch.zeros((B, Cp, H, W), dtype=torch.bfloat16, device=cuda).contiguous(memory_format=torch.channels_last)
    mir[:, :C] = torch.randn(B, C, H, W, device=cuda).to(torch.bfloat16)
    r = lambda t: t.detach().to(torch.bfloat16).double()
    x64 = mir[:, :C].double()
    want_flow = torch.nn.functional.conv2d(x64, r(head.weight), head.bias.double(), padding=1)
    flow, upf = net._head_up_mirror(l, mir, C, up)
    scale = float(want_flow.abs().max())
    assert float((flow.double() - want_flow).abs().max()) <= 2e-5 * scale + 1e-6
    if up:
        want_up = torch.nn.functional.conv_transpose2d(x64, r(dc.weight), dc.bias.double(), stride=2, padding=1)
        assert tuple(upf.shape) == tuple(want_up.shape)
        assert float((upf.double() - want_up).abs().max()) <= 2e-5 * float(want_up.abs().max()) + 1e-6
        buf = torch.full((B, 7, 2 * H, 2 * W), 3.0, device=cuda)
        net._head_up_mirror(l, mir, C, True, up_out=buf, up_coff=4)
        assert torch.equal(buf[:, 4:6], upf) and bool((buf[:, :4] == 3.0).all()) and bool((buf[:, 6:] == 3.0).all())
    else:
        assert upf is None


@pytest.mark.parametrize('kitti', [True, False])
def test_fused_pose_glue_matches_the_operator_by_operator_path(cuda, kitti):
    """islam_amd/glue.py: the (B,6) -> (B,7) algebra behind the networks (TartanVO.py:107-198 incl. the stereo scale's gradient,
    dense_ba.py:88-176) as ONE autograd node in numpy against the same algebra run operator by operator on the LieTensor shim
    (TartanVO(host_glue=True, fused_glue=False)): motions to 1e-12, the gradient w.r.t. the pose head's output to 1e-9 of its size --
    for an arbitrary upstream gradient on all seven slots, through the frame change BilevelLoop applies (train.py:214-215)."""
    from islam_amd import lietensor as pp, synthetic
    from islam_amd.TartanVO import TartanVO
    torch.manual_seed(5)
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=kitti, host_glue=True)
    smp = synthetic.stereo_batch(4, seed=21)
    smp = {k: (v.to(cuda) if isinstance(v, torch.Tensor) and (k.startswith('img') or k == 'intrinsic') else v) for k, v in smp.items()}
    B = 4
    g = torch.Generator().manual_seed(2)
    flow = (torch.randn(B, 2, 112, 160, generator=g) * 0.6).to(cuda)
    disp = (0.6 + 0.4 * torch.rand(B, 1, 112, 160, generator=g)).to(cuda)           # x 12.5 -> 7.5 .. 12.5 px: above the KITTI threshold
    pose0 = torch.tensor([[0.3, -0.2, 1.0, 0.02, -0.01, 0.03], [1.0, 0.1, 0.2, -0.05, 0.02, 0.01],
                          [-0.4, 0.5, 0.8, 0.0, 0.0, 0.0], [0.1, 0.1, 0.9, 0.3, -0.2, 0.1]], device=cuda) / vo.pose_std      # (one row with a zero rotation)
    T_IL = pp.SE3(torch.tensor([0.1, -0.05, 0.2, 0.0499792, 0.0, 0.0, 0.9987503], dtype=torch.float64))
    w = torch.randn(B, 7, generator=g, dtype=torch.float64)

    def run(fused):
        vo.fused_glue = fused
        pose = pose0.clone().requires_grad_(True)
        vo.vonet.forward = lambda *a, **k: (flow, disp, pose)                 # the networks' outputs pinned: the glue alone is under test
        res = vo(smp)
        m = res['motion_host']
        P = T_IL @ m @ T_IL.Inv()
        (P.tensor() * w).sum().backward()
        return m.tensor().detach().clone(), pose.grad.detach().double().cpu().clone(), {k: res[k] for k in ('mask', 'depth', 'depth_mask')}
    m0, g0, r0 = run(False)
    m1, g1, r1 = run(True)
    assert float(g0.abs().max()) > 0
    torch.testing.assert_close(m1, m0, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(g1, g0, rtol=1e-6, atol=1e-9 * float(g0.abs().max()))       # (the gradient returns to the device in fp32)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
