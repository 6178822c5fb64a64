"""CPU tests of the host-side mirror: LieTensor shim (values vs the oracle, PyPose gradient conventions vs finite
differences), transformation helpers, the Canny oracle's known answers (OpenCV parity is unpinned)."""
import numpy as np
import pytest
import torch

from islam_amd import evaluate, lietensor as pp, transformation as tf
from oracle import lie


def _r(shape, seed, scale=1.0):
    return torch.tensor(np.random.default_rng(seed).normal(size=shape) * scale, dtype=torch.float64)


def test_values_match_oracle():
    xi, yi = _r((6, 6), 0, 0.6), _r((6, 6), 1, 0.6)
    X, Y = pp.se3(xi).Exp(), pp.se3(yi).Exp()
    Xn, Yn = lie.se3_exp(xi.numpy()), lie.se3_exp(yi.numpy())
    np.testing.assert_allclose(X.tensor().numpy(), Xn, atol=1e-14)
    np.testing.assert_allclose((X @ Y).tensor().numpy(), lie.se3_mul(Xn, Yn), atol=1e-14)
    np.testing.assert_allclose(X.Inv().tensor().numpy(), lie.se3_inv(Xn), atol=1e-14)
    np.testing.assert_allclose(X.Log().tensor().numpy(), xi.numpy(), atol=1e-13)
    p = _r((6, 3), 2)
    np.testing.assert_allclose((X @ p).numpy(), lie.se3_act(Xn, p.numpy()), atol=1e-14)
    np.testing.assert_allclose(X.rotation().Log().tensor().numpy(), lie.so3_log(Xn[:, 3:]), atol=1e-14)
    np.testing.assert_allclose(X.matrix()[:, :3, :3].numpy(), lie.quat_matrix(Xn[:, 3:]), atol=1e-14)
    assert isinstance(X[2:4], pp.LieTensor) and X[2:4].ltype is pp.SE3_type and X[1].shape == (7,)
    assert isinstance(X.cpu().detach().clone(), pp.LieTensor) and not isinstance(X.tensor(), pp.LieTensor)
    assert isinstance(X.numpy(), np.ndarray) and len(X) == 6
    assert isinstance(torch.cat((X.translation(), X.rotation().tensor()), 1), torch.Tensor)
    assert pp.SE3([0, 0, 0, 0, 0, 0, 1]).dtype == torch.get_default_dtype()
    assert [type(m) for m in X][:1] == [pp.LieTensor]


@pytest.mark.parametrize('op', ['mul_left', 'mul_right', 'inv', 'log', 'act', 'exp'])
def test_gradients_are_left_tangent_padded(op):
    """g_X[:6] = d f(Exp(d) X) / d d at d=0, g_X[6] = 0  (SURVEY Appendix C item 9)."""
    X0, Y0 = lie.se3_exp(_r((6,), 3, 0.5).numpy()), lie.se3_exp(_r((6,), 4, 0.5).numpy())
    w6, w3 = np.linspace(0.3, 1.1, 6), np.array([0.7, -0.4, 1.3])
    pt = np.array([0.5, -1.0, 2.0])

    def f_np(X):
        if op == 'mul_left':
            return float(w6 @ lie.se3_log(lie.se3_mul(X, Y0)))
        if op == 'mul_right':
            return float(w6 @ lie.se3_log(lie.se3_mul(Y0, X)))
        if op == 'inv':
            return float(w6 @ lie.se3_log(lie.se3_inv(X)))
        if op == 'log':
            return float(w6 @ lie.se3_log(X))
        return float(w3 @ lie.se3_act(X, pt))

    if op == 'exp':
        xi = _r((6,), 5, 0.4).requires_grad_(True)
        (torch.tensor(w6) @ (pp.se3(xi).Exp() @ pp.SE3(torch.tensor(Y0))).Log().tensor()).backward()
        h, fd = 1e-6, np.zeros(6)
        g = lambda v: float(w6 @ lie.se3_log(lie.se3_mul(lie.se3_exp(v), Y0)))
        for k in range(6):
            e = np.zeros(6)
            e[k] = h
            fd[k] = (g(xi.detach().numpy() + e) - g(xi.detach().numpy() - e)) / (2 * h)
        np.testing.assert_allclose(xi.grad.numpy(), fd, atol=1e-7)
        return
    X = torch.tensor(X0, requires_grad=True)
    XL, YL = pp.SE3(X), pp.SE3(torch.tensor(Y0))
    if op == 'mul_left':
        out = torch.tensor(w6) @ (XL @ YL).Log().tensor()
    elif op == 'mul_right':
        out = torch.tensor(w6) @ (YL @ XL).Log().tensor()
    elif op == 'inv':
        out = torch.tensor(w6) @ XL.Inv().Log().tensor()
    elif op == 'log':
        out = torch.tensor(w6) @ XL.Log().tensor()
    else:
        out = torch.tensor(w3) @ (XL @ torch.tensor(pt))
    out.backward()
    h, fd = 1e-6, np.zeros(6)
    for k in range(6):
        e = np.zeros(6)
        e[k] = h
        fd[k] = (f_np(lie.se3_mul(lie.se3_exp(e), X0)) - f_np(lie.se3_mul(lie.se3_exp(-e), X0))) / (2 * h)
    np.testing.assert_allclose(X.grad.numpy()[:6], fd, atol=1e-7)
    assert X.grad[6] == 0


def test_transformation_helpers():
    m6 = _r((5, 6), 7, 0.2)
    X = tf.cvtSE3_pypose(m6)
    np.testing.assert_allclose(X.tensor().numpy(), np.concatenate([m6[:, :3].numpy(), lie.so3_exp(m6[:, 3:].numpy())], 1), atol=1e-15)
    T = lie.from_matrix_SE3(np.array(tf._T_AXES))
    K = tf.tartan2kitti_pypose(m6)
    np.testing.assert_allclose(K.tensor().numpy(), lie.se3_mul(lie.se3_mul(T[None], X.tensor().numpy()), lie.se3_inv(T)[None]), atol=1e-14)
    # NED (x fwd, y right, z down) -> camera (x right, y down, z fwd): a forward motion becomes +z
    fwd = tf.tartan2kitti_pypose(torch.tensor([[1.0, 0, 0, 0, 0, 0]], dtype=torch.float64)).tensor().numpy()[0]
    np.testing.assert_allclose(fwd[:3], [0, 0, 1], atol=1e-15)
    T0 = pp.SE3(torch.tensor(lie.se3_exp(_r((6,), 8, 0.3).numpy())))
    P = tf.motion2pose_pypose(K, T0)
    assert P.shape == (6, 7)
    acc = T0.tensor().numpy()
    for k in range(5):                                            # sequential left-to-right products (G2)
        acc = lie.se3_mul(acc, K.tensor().numpy()[k])
        np.testing.assert_array_equal(P.tensor().numpy()[k + 1], acc)
    np.testing.assert_allclose(tf.pose2motion_pypose(P).tensor().numpy(), K.tensor().numpy(), atol=1e-14)
    assert P.dtype == torch.float64


def test_edge_mask_oracle_properties():
    """oracle/canny.py (the restatement of OpenCV 4.7 the HIP edge kernel is checked against, tests/test_edge_gpu.py):
    known answers of each stage."""
    from oracle import canny
    H, W = 64, 96
    img = np.full((1, 3, H, W), 0.2, np.float32)
    img[:, :, :, 48:] = 0.8                                       # one vertical step edge
    e = canny.canny(canny.to_u8(img)[0])
    cols = np.nonzero(e.any(0))[0].tolist()
    assert cols in ([47], [48]) and e[5:-5, cols[0]].all()         # a single-pixel-wide line (non-maximum suppression)
    assert set(np.unique(e)) == {0, 255}
    m = canny.edge_mask(img)
    assert m.shape == (1, 16, 24) and m.dtype == bool
    band = np.nonzero(m[0].any(0))[0]
    assert 3 <= len(band) <= 6 and band.min() >= 9 and band.max() <= 14      # 5x5 dilation of the line at x=12
    assert not canny.edge_mask(np.full((1, 3, H, W), 0.5, np.float32)).any()  # flat image: no edges
    # hysteresis: a weak edge (50 < |gradient| <= 100) survives only when it is connected to a strong one
    weak = np.full((H, W, 3), 80, np.uint8)
    weak[:, 48:] = 80 + 20                                        # Sobel response 4 * 20 = 80: a candidate, never strong
    assert not canny.canny(weak).any()
    joined = weak.copy()
    joined[40:, 48:] = 200                                        # the lower part of the same line is strong
    e2 = canny.canny(joined)
    assert e2[45:, 47:49].any() and e2[2:36, 47:49].any(1).all()  # ... and carries the weak part with it
    # quarter resize: mean of the centre 2x2 of each 4x4 cell, round half up
    u8 = np.arange(16, dtype=np.uint8).reshape(4, 4, 1)
    assert canny.resize_quarter(u8).item() == (5 + 6 + 9 + 10 + 2) // 4
    # truncating uint8 conversion and first-maximum channel selection
    assert canny.to_u8(np.full((1, 3, 1, 1), 0.999, np.float32)).item(0) == 254
    # dilate: a single pixel becomes a 5x5 block, clipped at the border
    one = np.zeros((9, 9), np.uint8)
    one[0, 4] = 255
    d = canny.dilate(one)
    assert d[:3, 2:7].all() and d.sum() == 255 * 15


def test_preprocess_matches_reference_geometry():
    """Device-side CropCenter / Normalize / intrinsics layer vs a plain numpy restatement of Datasets/utils.py."""
    from islam_amd import preprocess
    # KITTI raw 375x1242 -> fix-ratio upscale to 448x1484, crop x1=422
    assert preprocess.crop_center_geometry(375, 1242) == (448, 1484, 422, 0)
    assert preprocess.crop_center_geometry(480, 640) == (480, 640, 0, 16)          # TartanAir: crop only
    assert preprocess.crop_center_geometry(480, 752) == (480, 752, 56, 16)          # EuRoC
    g = torch.Generator().manual_seed(0)
    B, H, W = 2, 480, 752
    imgs = [torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8) for _ in range(3)]
    calib = torch.tensor([[458.6, 457.3, 367.2, 248.4]]).repeat(B, 1)
    ext = torch.tensor([[0.11, 0, 0, 0, 0, 0, 1.0]]).repeat(B, 1)
    s = preprocess.make_sample(*imgs, calib, ext, ['euroc'] * B)
    assert s['img0'].shape == (B, 3, 448, 640) and s['intrinsic'].shape == (B, 2, 112, 160)
    ref = imgs[0][:, 16:464, 56:696].permute(0, 3, 1, 2).float() / 255.0          # no resize needed: pure crop, exact
    torch.testing.assert_close(s['img0'], ref, rtol=0, atol=0)
    mean = torch.tensor(preprocess.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(preprocess.IMAGENET_STD).view(1, 3, 1, 1)
    torch.testing.assert_close(s['img0_norm'], (ref - mean) / std)
    np.testing.assert_allclose(s['intrinsic_calib'].numpy(), [[458.6, 457.3, 367.2 - 56, 248.4 - 16]] * B, rtol=1e-6)
    u = (np.arange(W, dtype=np.float32) - 367.2 + 0.5) / 458.6
    v = (np.arange(H, dtype=np.float32) - 248.4 + 0.5) / 457.3
    np.testing.assert_allclose(s['intrinsic'][0, 0, 0].numpy(), u[56:696][::4], rtol=1e-6)
    np.testing.assert_allclose(s['intrinsic'][0, 1, :, 0].numpy(), v[16:464][::4], rtol=1e-6)
    # resize path: intrinsics scale with the image, the layer stays consistent with the scaled calibration
    k = preprocess.make_sample(*[torch.randint(0, 256, (1, 375, 1242, 3), generator=g, dtype=torch.uint8) for _ in range(3)],
                               torch.tensor([[718.856, 718.856, 607.1928, 185.2157]]), ext[:1], ['kitti'])
    sc = 448 / 375
    np.testing.assert_allclose(k['intrinsic_calib'][0].numpy(), [718.856 * 1484 / 1242, 718.856 * sc, 607.1928 * 1484 / 1242 - 422, 185.2157 * sc], rtol=1e-5)
    assert k['img0'].shape == (1, 3, 448, 640) and 0.0 <= float(k['img0'].min()) and float(k['img0'].max()) <= 1.0


def test_ate_evaluator_known_answers():
    """Rigid + similarity alignment recover a known transform; RPE of a trajectory against itself is zero."""
    rng = np.random.default_rng(0)
    gt = np.cumsum(rng.normal(size=(50, 3)), 0)
    R = lie.quat_matrix(lie.so3_exp(np.array([0.3, -0.2, 0.5])))
    est = (gt - np.array([1.0, 2.0, 3.0])) @ R / 1.7
    assert evaluate.ate(est, gt, with_scale=True)[0] < 1e-9
    assert evaluate.ate(est * 1.7, gt)[0] < 1e-9
    noisy = gt + rng.normal(scale=0.1, size=gt.shape)
    a = evaluate.ate(noisy, gt)[0]
    assert 0.1 < a < 0.25
    X = np.concatenate([gt, np.tile([0, 0, 0, 1.0], (50, 1))], 1)
    assert evaluate.rpe(X, X) == (0.0, 0.0)


def test_dense_ba_helpers_known_answers():
    """pixel2point against the reference's own docstring vector (dense_ba.py:29-47); proj mask; keypoint sampler."""
    from islam_amd import dense_ba
    K = torch.tensor([[2.0, 0, 4.5], [0, 2.0, 4.5], [0, 0, 1]])
    px = torch.tensor([[0.5, 0.0], [1.0, 0.0], [0.0, 1.3], [1.0, 0.0], [0.5, 1.5], [5.0, 1.5]])
    dep = torch.tensor([5.0, 3.0, 6.5, 2.0, 0.5, 0.7])
    want = torch.tensor([[-10.0, -11.25, 5.0], [-5.25, -6.75, 3.0], [-14.625, -10.4, 6.5], [-3.5, -4.5, 2.0],
                         [-1.0, -0.75, 0.5], [0.175, -1.05, 0.7]])
    torch.testing.assert_close(dense_ba.pixel2point(px, dep, K), want)
    x = torch.tensor([[0.5, 0.2, 1.0], [3.0, 0.0, 1.0], [0.1, 0.1, 0.05]])
    p, m = dense_ba.proj(x, return_mask=True)
    assert m.tolist() == [True, False, False] and torch.equal(p[1], torch.zeros(3))
    mask = torch.zeros(2, 8, 10, dtype=torch.bool)
    mask[0, 2:6, 3:7] = True
    mask[1, 0, 0] = True
    kp = dense_ba.sample_keypoints(mask, 6, torch.Generator().manual_seed(0))
    assert kp.shape == (2, 6, 2)
    assert mask[0][kp[0, :, 1].long(), kp[0, :, 0].long()].all() and len({tuple(r) for r in kp[0].tolist()}) == 6
    assert (kp[1] == 0).all()


def test_host_lm_control_follows_ieee_division_at_an_exactly_converged_point():
    """at D == 0 the trust-region quality is 0/0 = NaN, which fails both comparisons and takes the
    shrink-the-radius branch in PyPose (and in the device code); the host LMControl of the sharded / dense paths must take
    the same damping trajectory as the oracle's TrustRegion."""
    from islam_amd.lm_control import LMControl
    from oracle import pvgo as opvgo
    ctl, tr = LMControl(radius=1e4), opvgo.TrustRegion(radius=1e4)
    ctl.set_initial_loss(2.5)
    for last, loss, q in ((2.5, 2.5, 0.0), (2.5, 2.5, -0.0), (2.5, 2.4, 0.0), (2.5, 2.6, 0.0), (2.5, 2.0, -0.8)):
        ctl.loss = last
        ctl.begin_step()
        ctl.after_trial(loss, q)
        # the oracle's update takes J.D and R; a one-row system with (JD)^T (2R + JD) = q:  JD = 1, R = (q - 1) / 2
        tr.update(last, loss, np.array([1.0]), np.array([(q - 1.0) / 2.0]))
        assert ctl.damping == tr.pg['damping'], (last, loss, q)


@pytest.mark.parametrize('kitti', [True, False])
def test_fused_pose_glue_math_against_the_lietensor_operators(kitti):
    """islam_amd/glue.py (the numpy core of the fused pose algebra, TartanVO.py:107-198 + the scale gradient of dense_ba.py:88-176) against
    the same algebra written with the LieTensor shim's operators and differentiated by autograd -- on the CPU, with a stand-in for the one
    device call (islam_scale_ls: any positive-definite sums will do; the algebra only consumes them).  The GPU test
    tests/test_frontend_gpu.py::test_fused_pose_glue_matches_the_operator_by_operator_path holds the two product paths against each
    other with the real kernel."""
    import numpy as np
    import torch
    from islam_amd import glue, lietensor as pp
    from islam_amd.transformation import cvtSE3_pypose, tartan2kitti_pypose
    rng = np.random.default_rng(4)
    B = 5
    p0 = np.concatenate([rng.normal(0, 0.5, (B, 3)), rng.normal(0, 0.2, (B, 3))], 1)
    p0[2, 3:] = 0.0                                                    # a zero rotation: the Taylor branches
    sums = rng.normal(0, 1.0, (B, 18))
    sums[:, 0] = rng.uniform(2.0, 5.0, B)                              # MM > 0
    sums[:, 1] = rng.uniform(0.5, 3.0, B)
    intr4 = np.tile(np.array([[180.0, 175.0, 80.0, 56.0]]), (B, 1))
    w = rng.normal(size=(B, 7))
    # ---- the fused core
    M, saved = glue.glue_forward_np(p0, kitti, lambda pose_enu: sums)
    # ---- operator by operator (TartanVO.forward's host_glue branch + TartanVO.stereo_scale's re-attachment), autograd through the shim
    p = torch.tensor(p0, requires_grad=True)
    pose_enu = tartan2kitti_pypose(p)
    st = torch.tensor(sums)
    sv = torch.tensor(np.float32(1.0 / sums[:, 0] * sums[:, 1]).astype(np.float64))
    fx, fy, cx, cy = torch.tensor(intr4).unbind(-1)
    MM = st[:, 0]
    ga = (torch.stack([-st[:, 2], -st[:, 3], st[:, 4]], -1) - sv[:, None] * torch.stack([-2 * st[:, 5], -2 * st[:, 6], 2 * st[:, 7]], -1)) / MM[:, None]
    GR = torch.stack([fx[:, None] * st[:, 8:11], fy[:, None] * st[:, 11:14], st[:, 14:17]], 1) / MM[:, None, None]
    Tinv = pose_enu.Inv()
    tn = torch.nn.functional.normalize(Tinv.translation(), dim=-1)
    a = torch.stack([fx * tn[:, 0] + cx * tn[:, 2], fy * tn[:, 1] + cy * tn[:, 2], tn[:, 2]], -1)
    R = Tinv.rotation()
    eye = torch.eye(3, dtype=torch.float64)
    cols = torch.stack([R.Act(eye[j].expand(B, 3)) for j in range(3)], -1)
    sur = (ga * a).sum(-1) + (GR * cols).sum((-1, -2))
    scale = sv + (sur - sur.detach())
    pose = torch.cat([torch.nn.functional.normalize(p[:, :3], dim=1) * scale.view(-1, 1), p[:, 3:]], dim=1)
    motion = tartan2kitti_pypose(pose) if kitti else cvtSE3_pypose(pose)
    (motion.tensor() * torch.tensor(w)).sum().backward()
    np.testing.assert_allclose(M, motion.tensor().detach().numpy(), rtol=1e-13, atol=1e-13)
    got = glue.glue_backward_np(w[:, :6], saved, intr4, kitti)
    np.testing.assert_allclose(got, p.grad.numpy(), rtol=1e-9, atol=1e-11 * np.abs(p.grad.numpy()).max())
