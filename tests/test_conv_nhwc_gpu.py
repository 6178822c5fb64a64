"""GPU tests of islam_conv_nhwc_bf16 (channels-last bf16 implicit-GEMM convolution with BatchNorm folded into both ends) against
plain torch fp32 on the same bf16 operands.  The output must be the round-to-nearest-even bf16 of the fp32 result (up to fp32
summation order): no truncation bias (what MIOpen's kernels for these shapes have, scripts/calib/bf16_rounding_probe.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _mk(B, Cin, H, W, Cout, k, seed=0, dev='cuda'):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=CL)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev).to(torch.bfloat16)
    return x, w


def _ulps(got, ref32):
    """|got - ref| in units of the bf16 spacing at ref."""
    r = ref32.float()
    spacing = torch.pow(2.0, torch.floor(torch.log2(r.abs().clamp_min(1e-30))) - 7)
    return ((got.float() - r).abs() / spacing)


@pytest.mark.parametrize('B,Cin,H,W,Cout,k', [(2, 32, 64, 96, 32, 3), (1, 64, 40, 72, 64, 3), (2, 128, 24, 40, 128, 3), (1, 352, 24, 32, 128, 3),
                                               (2, 128, 33, 47, 64, 1), (1, 64, 17, 50, 128, 1), (1, 48, 19, 35, 40, 3), (3, 32, 7, 5, 32, 3)])
def test_plain_convolution_rounds_to_nearest(cuda, B, Cin, H, W, Cout, k):
    from islam_amd import ops
    x, w = _mk(B, Cin, H, W, Cout, k)
    y = ops.conv_nhwc(x, ops.pack_conv_nhwc_weight(w), Cout, k)
    assert y.shape == (B, Cout, H, W) and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=CL)
    ref = F.conv2d(x.float(), w.float(), None, 1, k // 2)
    # RNE of a sum that differs from torch's only by summation order: half a bf16 spacing + the fp32 noise of the sum, which is
    # relative to the LARGEST partial sum, not to the (possibly cancelled) result -> bound it against the output's scale
    err = (y.float() - ref).abs()
    scale = float(ref.abs().max())
    assert float((err - 0.5 * torch.pow(2.0, torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)).max()) <= 1e-5 * scale
    want = ref.to(torch.bfloat16)
    differs = (y != want)
    assert float(differs.float().mean()) < 0.02          # a truncating kernel differs on ~50 % of the outputs
    big = ref.abs() > 1e-2 * scale
    signed = (((y.float().abs() - ref.abs()) / ref.abs().clamp_min(1e-20))[big]).mean() * 512
    assert abs(float(signed)) < 0.05                      # ... with a mean signed error of -1.44 (in units of 2^-9)


def test_epilogue_bias_residual_relu(cuda):
    from islam_amd import ops
    x, w = _mk(2, 64, 30, 44, 64, 3, seed=1)
    g = torch.Generator().manual_seed(2)
    bias = torch.randn(64, generator=g).to(cuda)
    res = torch.randn(2, 64, 30, 44, generator=g).to(cuda).to(torch.bfloat16).contiguous(memory_format=CL)
    y = ops.conv_nhwc(x, ops.pack_conv_nhwc_weight(w), 64, 3, bias=bias, res=res, relu=True)
    a = F.conv2d(x.float(), w.float(), None, 1, 1)
    want = F.relu(((a + bias.view(1, -1, 1, 1)).to(torch.bfloat16).float() + res.float()).to(torch.bfloat16))
    assert float((y != want).float().mean()) < 0.02
    assert float((y.float() - want.float()).abs().max()) <= 2 ** -6 * float(want.float().abs().max())


def test_batchnorm_statistics_from_the_epilogue_and_apply_on_load(cuda):
    """convbn + ReLU -> convbn (+ residual), the PSM BasicBlock (submodule.py:66-88), train mode: statistics of both BatchNorms
    from the convolutions' epilogues, the first normalisation + ReLU applied while the second convolution stages its input."""
    from islam_amd import ops
    torch.manual_seed(3)
    B, C, H, W = 4, 32, 48, 80
    x, w1 = _mk(B, C, H, W, C, 3, seed=4)
    _, w2 = _mk(B, C, H, W, C, 3, seed=5)
    bn1, bn2 = torch.nn.BatchNorm2d(C).to(cuda).train(), torch.nn.BatchNorm2d(C).to(cuda).train()
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    import copy
    r1, r2 = copy.deepcopy(bn1), copy.deepcopy(bn2)
    # reference: fp32 everywhere on the bf16 operands, bf16 roundings at the same places
    with torch.no_grad():
        a1 = F.conv2d(x.float(), w1.float(), None, 1, 1)
        n1 = F.relu(r1(a1.to(torch.bfloat16).float()).to(torch.bfloat16))
        a2 = F.conv2d(n1.float(), w2.float(), None, 1, 1)
        want = (r2(a2.to(torch.bfloat16).float()).to(torch.bfloat16).float() + x.float()).to(torch.bfloat16)
    y1, f1 = ops.conv_nhwc(x, ops.pack_conv_nhwc_weight(w1), C, 3, stats=True)
    # the folded partial sums are the sums over the stored (bf16) outputs: what nn.BatchNorm2d would reduce
    s = f1.view(256, 2, C).double().sum(0)
    a1r = y1.double()
    np.testing.assert_allclose(s[0].cpu().numpy(), a1r.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(s[1].cpu().numpy(), (a1r ** 2).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5)
    ss1 = ops.bn_finalize(f1, bn1, B * H * W)
    y2, f2 = ops.conv_nhwc(y1, ops.pack_conv_nhwc_weight(w2), C, 3, in_affine=ss1, stats=True)
    ss2 = ops.bn_finalize(f2, bn2, B * H * W)
    out = ops.bn_apply_(y2, ss2, relu=False, res=x)
    d = (out.float() - want.float()).abs()
    assert float(d.max()) <= 2 ** -5 * float(want.float().abs().max()) and float((out != want).float().mean()) < 0.1
    for mine, ref in ((bn1, r1), (bn2, r2)):
        torch.testing.assert_close(mine.running_mean, ref.running_mean, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(mine.running_var, ref.running_var, rtol=2e-3, atol=1e-5)
        assert int(mine.num_batches_tracked) == 1
    # deterministic: fixed-order statistics
    _, f1b = ops.conv_nhwc(x, ops.pack_conv_nhwc_weight(w1), C, 3, stats=True)
    assert torch.equal(f1, f1b)


def test_argument_checks(cuda):
    from islam_amd import _lib, ops
    x, w = _mk(1, 32, 16, 16, 32, 3)
    with pytest.raises(_lib.IslamHipError):
        ops.conv_nhwc(x, ops.pack_conv_nhwc_weight(w), 32, 3, bias=torch.zeros(32, device=cuda), stats=True)
    x6 = torch.zeros(1, 6, 16, 16, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=CL)
    with pytest.raises(_lib.IslamHipError):
        ops.conv_nhwc(x6, torch.zeros(9 * 64 * 32, device=cuda, dtype=torch.bfloat16), 32, 3)


def test_relu_of_the_input_on_load(cuda):
    from islam_amd import ops
    x, w = _mk(2, 64, 21, 37, 32, 1, seed=7)
    b = torch.linspace(-0.5, 0.5, 32, device=cuda)
    y = ops.conv_nhwc(x, ops.pack_conv_nhwc_weight(w), 32, 1, bias=b, relu=True, in_relu=True)
    want = ops.conv_nhwc(F.relu(x), ops.pack_conv_nhwc_weight(w), 32, 1, bias=b, relu=True)
    assert torch.equal(y, want)
    ref = F.relu((F.conv2d(F.relu(x).float(), w.float(), None) + b.view(1, -1, 1, 1)).to(torch.bfloat16))
    assert float((y != ref).float().mean()) < 0.02


@pytest.mark.parametrize('B,H,W,pieces,dense', [(2, 28, 40, (81, 64, 2, 2), (128, 128, 96, 64, 32)), (1, 7, 10, (81,), (128, 96, 32)),
                                               (2, 33, 45, (81, 32, 2, 2), (64, 32))])
def test_flow_block_on_the_channels_last_mirror(cuda, B, H, W, pieces, dense):
    """islam_conv_nhwc_flow + islam_nchw_f32_to_nhwc_bf16: a DenseNet block of the flow net ([newest | ... | input] buffer, every
    layer reads a suffix and writes the slice before it) through the bf16 mirror equals the same block on the fp32 NCHW kernel
    (islam_conv3x3_mfma: identical bf16 operands, different summation order) and a torch reference on bf16-rounded operands."""
    import torch.nn.functional as F
    from islam_amd import ops
    g = torch.Generator().manual_seed(5)
    od, nd = sum(pieces), sum(dense)
    tot = od + nd
    base = torch.randn(B, od, H, W, generator=g).to(cuda)
    ws = []
    cin = od
    for c in dense:
        ws.append(((torch.randn(c, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(cuda), (torch.randn(c, generator=g) * 0.1).to(cuda)))
        cin += c
    # fp32 NCHW kernel
    buf0 = torch.zeros(B, tot, H, W, device=cuda)
    buf0[:, nd:] = base
    off = nd
    for (w, b), c in zip(ws, dense):
        ops.conv3x3_mfma(buf0, ops.pack_conv3x3_weight(w), b, c, 1, 1, 0.1, buf0, off - c, off)
        off -= c
    # mirror path
    buf1 = torch.zeros(B, tot, H, W, device=cuda)
    buf1[:, nd:] = base
    totp = nd + (od + 7) // 8 * 8
    mir = torch.full((B, totp, H, W), float('nan'), dtype=torch.bfloat16, device=cuda).contiguous(memory_format=torch.channels_last)
    ops.nchw_to_nhwc_mirror(buf1, nd, od, mir, nd)
    assert torch.equal(mir[:, nd:nd + od].float(), base.to(torch.bfloat16).float())               # rounded to nearest even
    assert float(mir[:, nd + od:].float().abs().max()) == 0.0 if totp > tot else True             # padding channels zeroed
    off = nd
    for (w, b), c in zip(ws, dense):
        wp = ops.pack_conv_nhwc_weight(F.pad(w, (0, 0, 0, 0, 0, totp - off - w.shape[1])).to(torch.bfloat16))
        ops.conv_nhwc_flow(mir, off, totp - off, wp, b, buf1, off - c, c, 0.1, ymir=mir, moff=off - c)
        off -= c
    scale = float(buf0.abs().max())
    assert float((buf1 - buf0).abs().max()) <= 2e-3 * scale              # bf16 re-rounding of slightly different fp32 sums downstream
    assert torch.equal(mir[:, :tot].float(), buf1.to(torch.bfloat16).float())                     # the mirror IS the rounded buffer
    # torch reference on the operands the kernels see: bf16-rounded activations and weights, fp32 accumulation
    ref = torch.zeros(B, tot, H, W, device=cuda)
    ref[:, nd:] = base
    off = nd
    for (w, b), c in zip(ws, dense):
        y = F.conv2d(ref[:, off:].to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), b.double(), padding=1)
        ref[:, off - c:off] = F.leaky_relu(y, 0.1).float()
        off -= c
    assert float((buf1 - ref).abs().max()) <= 2e-3 * scale
    # without a mirror output (dc_conv1): fp32 only, slope 1 = no activation
    w, b = ws[0]
    out = torch.empty(B, dense[0], H, W, device=cuda)
    wp = ops.pack_conv_nhwc_weight(F.pad(w, (0, 0, 0, 0, 0, totp - nd - w.shape[1])).to(torch.bfloat16))
    ops.conv_nhwc_flow(mir, nd, totp - nd, wp, b, out, 0, dense[0], 1.0)
    y = F.conv2d(base.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), b.double(), padding=1).float()
    assert float((out - y).abs().max()) <= 1e-4 * float(y.abs().max())


@pytest.mark.parametrize('d,H,W,Cin,Cout', [(2, 28, 40, 128, 128), (4, 28, 40, 128, 96), (8, 16, 24, 96, 64), (16, 32, 48, 64, 32), (2, 6, 10, 40, 8)])
def test_dilated_flow_convolution_on_sub_grids(cuda, d, H, W, Cin, Cout):
    """islam_conv_nhwc_flow(dilation=d) = d*d dense convolutions on the sub-grids of the map: equals the dilated convolution of the
    fp32 NCHW kernel (same bf16 operands) and of torch on bf16-rounded operands, through both outputs (fp32 NCHW, bf16 mirror)."""
    import torch.nn.functional as F
    from islam_amd import ops
    g = torch.Generator().manual_seed(d)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(cuda)
    b = (torch.randn(Cout, generator=g) * 0.1).to(cuda)
    xm = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = F.leaky_relu(F.conv2d(xm.double(), w.to(torch.bfloat16).double(), b.double(), padding=d, dilation=d), 0.1).float()
    wp = ops.pack_conv_nhwc_weight(w.to(torch.bfloat16))
    y32 = torch.full((B, Cout, H, W), float('nan'), device=cuda)
    ym = torch.full((B, Cout, H, W), float('nan'), dtype=torch.bfloat16, device=cuda).contiguous(memory_format=torch.channels_last)
    ops.conv_nhwc_flow(xm, 0, Cin, wp, b, y32, 0, Cout, 0.1, ymir=ym, moff=0, dilation=d)
    scale = float(ref.abs().max())
    assert float((y32 - ref).abs().max()) <= 1e-4 * scale
    assert torch.equal(ym.float(), y32.to(torch.bfloat16).float())
    only = torch.full_like(ym, float('nan'))
    ops.conv_nhwc_flow(xm, 0, Cin, wp, b, None, 0, Cout, 0.1, ymir=only, moff=0, dilation=d)              # mirror-only output
    assert torch.equal(only, ym)
    if Cin >= 16 and d <= 8:                                                                               # the fp32 NCHW kernel's range
        old = ops.conv3x3_mfma(xm.float().contiguous(), ops.pack_conv3x3_weight(w), b, Cout, 1, d, 0.1)
        assert float((y32 - old).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize('B,Cin,H,W,Cout,skip_c,relu', [(2, 512, 7, 10, 512, 384, True), (1, 896, 14, 20, 320, 256, True), (2, 256, 37, 45, 64, 0, True),
                                                        (1, 64, 16, 33, 128, 8, False), (3, 32, 5, 3, 72, 0, True), (2, 64, 9, 20, 32, 0, True), (1, 40, 11, 37, 24, 16, False)])
def test_transposed_convolution_on_the_parity_classes(cuda, B, Cin, H, W, Cout, skip_c, relu):
    """islam_deconv4x4s2_nhwc_bf16 (the stereo decoder's ConvTranspose2d(k=4, s=2, p=1) + bias + ReLU as four 2x2 convolutions, written
    into a channel slice of the concatenation) against torch's conv_transpose2d in fp32 on the same bf16 operands."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=CL)
    w = (torch.randn(Cin, Cout, 4, 4, generator=g) / (Cin * 4) ** 0.5).cuda().to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g).cuda()
    out = None
    if skip_c:
        out = torch.full((B, Cout + skip_c, 2 * H, 2 * W), 7.0, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL)
    y = ops.deconv_nhwc(x, ops.pack_deconv_nhwc_weight(w), bias, Cout, out=out, yoff=0, relu=relu)
    ref = F.conv_transpose2d(x.float(), w.float(), bias, stride=2, padding=1)
    if relu:
        ref = F.relu(ref)
    got = y[:, :Cout].float()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    scale = float(ref.abs().max())
    assert float((err - 0.5 * torch.pow(2.0, torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)).max()) <= 2e-5 * scale
    if skip_c:
        assert bool((y[:, Cout:] == 7.0).all())                      # the rest of the concatenation buffer is left alone


@pytest.mark.parametrize('B,C1,C2,H,W,Cout', [(2, 512, 384, 7, 10, 320), (1, 320, 256, 14, 20, 192), (2, 192, 192, 28, 40, 128), (1, 128, 128, 37, 45, 64),
                                              (3, 64, 8, 9, 20, 32), (1, 32, 24, 5, 3, 72)])
def test_transposed_convolution_of_a_concatenation_read_from_its_two_tensors(cuda, B, C1, C2, H, W, Cout):
    """islam_deconv4x4s2_nhwc_bf16_cat(x1, x2) == islam_deconv4x4s2_nhwc_bf16(torch.cat((x1, x2), 1)), bit for bit: the decoder's
    concatenations (Network/StereoNet7.py:121-138) are read where their halves lie; the benched decoder's four shapes and two small ones."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(C1 + C2)
    x1 = torch.randn(B, C1, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=CL)
    x2 = torch.randn(B, C2, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=CL)
    w = (torch.randn(C1 + C2, Cout, 4, 4, generator=g) / ((C1 + C2) * 4) ** 0.5).cuda().to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g).cuda()
    wp = ops.pack_deconv_nhwc_weight(w)
    want = ops.deconv_nhwc(torch.cat((x1, x2), 1).contiguous(memory_format=CL), wp, bias, Cout, relu=True)
    got = ops.deconv_nhwc(x1, wp, bias, Cout, relu=True, x2=x2)
    assert torch.equal(got, want)
    out = torch.full((B, Cout + 16, 2 * H, 2 * W), 7.0, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=CL)
    ops.deconv_nhwc(x1, wp, bias, Cout, out=out, yoff=8, relu=False, x2=x2)
    assert torch.equal(out[:, 8:8 + Cout], ops.deconv_nhwc(torch.cat((x1, x2), 1).contiguous(memory_format=CL), wp, bias, Cout, relu=False))
    assert bool((out[:, :8] == 7.0).all()) and bool((out[:, 8 + Cout:] == 7.0).all())


@pytest.mark.parametrize('B,C1,C2,H,W,Cout', [(2, 64, 64, 64, 96, 64), (1, 64, 64, 37, 70, 64), (2, 16, 8, 10, 12, 32), (1, 48, 40, 9, 7, 72)])
def test_stride2_convolution_of_a_concatenation_read_from_its_two_tensors(cuda, B, C1, C2, H, W, Cout):
    """islam_conv_nhwc_bf16_s2_cat == islam_conv_nhwc_bf16_s2 on the concatenated tensor, bit for bit (kernel size 2, bias, ReLU, cropped
    output: the quarter-resolution tail StereoNet7._deconv_c11_quarter runs)."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(C1 * 3 + C2)
    x1 = torch.randn(B, C1, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=CL)
    x2 = torch.randn(B, C2, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=CL)
    w = (torch.randn(Cout, C1 + C2, 2, 2, generator=g) / ((C1 + C2) * 4) ** 0.5).cuda().to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g).cuda()
    wp = ops.pack_conv_nhwc_weight(w)
    hw = (H // 2, W // 2)
    want = ops.conv_nhwc_s2(torch.cat((x1, x2), 1).contiguous(memory_format=CL), wp, Cout, 2, bias=bias, relu=True, out_hw=hw)
    got = ops.conv_nhwc_s2(x1, wp, Cout, 2, bias=bias, relu=True, out_hw=hw, x2=x2)
    assert torch.equal(got, want)


def test_stereo_decoder_on_the_transposed_convolution_kernel(cuda):
    """StereoNet7._deconv_act on the bf16 channels-last execution copy: HIP path == torch path (MIOpen + ReLU + torch.cat) within bf16
    rounding, for a plain output and for one written into its concatenation."""
    from islam_amd import nets
    torch.manual_seed(3)
    net = nets.StereoNet7().cuda().to(torch.bfloat16).to(memory_format=CL).eval()
    x = torch.randn(2, 896, 14, 20, device='cuda').to(torch.bfloat16).contiguous(memory_format=CL)
    skip = torch.randn(2, 256, 28, 40, device='cuda').to(torch.bfloat16).contiguous(memory_format=CL)
    with torch.no_grad():
        old, oldp = nets.HIP_DECONV, nets.CAT_PAIRS
        nets.CAT_PAIRS = True
        try:
            pair = net._deconv_act(net.deconv_c7, x, skip)      # (ISLAM_CAT_PAIRS=1: the concatenation as a pair of dense tensors, nets._Cat)
            assert isinstance(pair, nets._Cat) and pair.b is skip and pair.readable(576, 32)
            nxt = net._deconv_act(net.deconv_c8, pair)          # ... which the next transposed convolution reads where they lie
            nets.CAT_PAIRS = False
            got = net._deconv_act(net.deconv_c7, x, skip)       # the concatenation buffer
            assert torch.equal(got, pair.materialize())
            assert torch.equal(net._deconv_act(net.deconv_c8, got), nxt)
            nets.HIP_DECONV = False
            ref = net._deconv_act(net.deconv_c7, x, skip)
        finally:
            nets.HIP_DECONV, nets.CAT_PAIRS = old, oldp
        ref32 = torch.cat((F.relu(F.conv_transpose2d(x.float(), net.deconv_c7.weight.float(), net.deconv_c7.bias.float(), stride=2, padding=1)),
                           skip.float()), 1)
    assert got.shape == ref.shape == ref32.shape and got.is_contiguous(memory_format=CL)
    e_hip = float((got.float() - ref32).abs().max()), float((ref.float() - ref32).abs().max())
    assert e_hip[0] <= 1.0 * float(ref32.abs().max()) * 2 ** -8          # half a bf16 spacing at the output's scale
    assert bool((got[:, 320:] == skip).all())


@pytest.mark.parametrize('B,Cin,H,W,Cout', [(3, 64, 40, 72, 128), (2, 128, 19, 50, 192), (1, 32, 33, 31, 72)])
def test_statistics_of_layers_with_several_channel_blocks(cuda, B, Cin, H, W, Cout):
    """Layers with more than one 64-channel output block launch in the XCD-aware order (a workgroup's pixel tile and channel block
    come from its logical work index): the per-tile partial sums must still land on their tile's row -- the folded statistics equal
    the sums over the stored output, and they are the same bits on every run."""
    from islam_amd import ops
    x, w = _mk(B, Cin, H, W, Cout, 3, seed=11)
    wp = ops.pack_conv_nhwc_weight(w)
    y, f = ops.conv_nhwc(x, wp, Cout, 3, stats=True)
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    assert float((y.float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    s = f.view(256, 2, Cout).double().sum(0)
    yd = y.double()
    np.testing.assert_allclose(s[0].cpu().numpy(), yd.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(s[1].cpu().numpy(), (yd ** 2).sum((0, 2, 3)).cpu().numpy(), rtol=1e-5)
    y2, f2 = ops.conv_nhwc(x, wp, Cout, 3, stats=True)
    assert torch.equal(f, f2) and torch.equal(y, y2)


@pytest.mark.parametrize('B,Cin,H,W,Cout,k', [(16, 32, 224, 320, 32, 3), (2, 64, 40, 72, 64, 3), (2, 128, 24, 40, 128, 3), (1, 64, 17, 50, 128, 1),
                                               (3, 32, 7, 5, 32, 3), (1, 48, 19, 35, 40, 3)])
def test_convolution_with_fold_and_finalize_in_one_launch(cuda, B, Cin, H, W, Cout, k):
    """islam_conv_nhwc_bf16_bn == islam_conv_nhwc_bf16(stats) + islam_bn_finalize, bit for bit: raw output, [scale | shift], running
    statistics, batch counter; twice in a row (the ticket counter must come back to zero) and with the producer's affine on load."""
    from islam_amd import ops
    x, w = _mk(B, Cin, H, W, Cout, k, seed=B + Cin + k)
    wp = ops.pack_conv_nhwc_weight(w)
    g = torch.Generator().manual_seed(5)
    aff = torch.cat((torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.1)).to(cuda)

    def bn():
        m = torch.nn.BatchNorm2d(Cout).to(cuda).train()
        with torch.no_grad():
            m.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
            m.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        return m
    for in_affine in (None, aff):
        a, b = bn(), bn()
        b.load_state_dict(a.state_dict())
        for _ in range(2):
            y0, folded = ops.conv_nhwc(x, wp, Cout, k, in_affine=in_affine, stats=True)
            s0 = ops.bn_finalize(folded, a, B * H * W)
            y1, s1 = ops.conv_nhwc_bn(x, wp, Cout, k, b, in_affine=in_affine)
            assert torch.equal(y0, y1) and torch.equal(s0, s1)
            assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
            assert int(a.num_batches_tracked) == int(b.num_batches_tracked)
    assert all(int(t[0].abs().sum()) == 0 for t in ops._BN_COUNTERS.values())


@pytest.mark.parametrize('B,Cin,H,W,Cout,k,ytot,yoff', [(2, 128, 24, 40, 64, 1, 136, 64), (3, 32, 17, 33, 32, 3, 72, 8), (1, 64, 40, 72, 128, 3, 128, 0)])
def test_convolution_into_a_channel_slice(cuda, B, Cin, H, W, Cout, k, ytot, yoff):
    """islam_conv_nhwc_bf16_into: the dense result, bit for bit, in channels [yoff, yoff + Cout) of a larger tensor whose other channels
    stay untouched; batch ranges of the input go to the same destination rows (how StereoNet7 assembles conv_c0's input)."""
    from islam_amd import ops
    x, w = _mk(B, Cin, H, W, Cout, k, seed=ytot)
    wp = ops.pack_conv_nhwc_weight(w)
    aff = torch.cat((torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1)).to(cuda)
    want = ops.conv_nhwc(x, wp, Cout, k, in_affine=aff)
    out = torch.full((B, ytot, H, W), 7.0, dtype=torch.bfloat16, device=cuda).contiguous(memory_format=CL)
    ops.conv_nhwc_into(x, wp, Cout, k, out, yoff, in_affine=aff)
    assert torch.equal(out[:, yoff:yoff + Cout], want)
    keep = torch.ones(ytot, dtype=torch.bool)
    keep[yoff:yoff + Cout] = False
    assert bool((out[:, keep.to(cuda)] == 7.0).all())
    if B > 1:                                   # a batch range of the input into the rows of a smaller destination
        out1 = torch.zeros((1, ytot, H, W), dtype=torch.bfloat16, device=cuda).contiguous(memory_format=CL)
        ops.conv_nhwc_into(x[1:2], wp, Cout, k, out1, yoff, in_affine=aff)
        assert torch.equal(out1[:, yoff:yoff + Cout], want[1:2])


@pytest.mark.parametrize('B,Cin,H,W,Cout,k', [(2, 32, 64, 96, 64, 3), (1, 32, 33, 47, 64, 3), (2, 32, 64, 96, 64, 1), (1, 64, 37, 70, 32, 2), (2, 64, 16, 40, 128, 3),
                                               (1, 40, 9, 9, 24, 3)])
def test_stride_two_convolution(cuda, B, Cin, H, W, Cout, k):
    """islam_conv_nhwc_bf16_s2 against torch.conv2d(stride = 2, padding = k // 2) on the same bf16 operands: round-to-nearest output,
    the producer's affine + ReLU on load with zero padding of the NORMALISED input, batch statistics of the bf16-rounded output, bias +
    ReLU epilogue, and an output window smaller than the convolution's (the quarter-resolution tail)."""
    from islam_amd import ops
    x, w = _mk(B, Cin, H, W, Cout, k, seed=H + k)
    wp = ops.pack_conv_nhwc_weight(w)
    P = k // 2
    ref = F.conv2d(x.float(), w.float(), None, 2, P)
    y = ops.conv_nhwc_s2(x, wp, Cout, k)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=CL)
    scale = float(ref.abs().max())
    err = (y.float() - ref).abs()
    assert float((err - 0.5 * torch.pow(2.0, torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)).max()) <= 1e-5 * scale
    # affine + ReLU on load, statistics
    g = torch.Generator().manual_seed(7)
    aff = torch.cat((torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.1)).to(cuda)
    xn = F.relu((x.float() * aff[:Cin].view(1, -1, 1, 1) + aff[Cin:].view(1, -1, 1, 1)).to(torch.bfloat16)).float()
    ref2 = F.conv2d(xn, w.float(), None, 2, P)
    y2, folded = ops.conv_nhwc_s2(x, wp, Cout, k, in_affine=aff, stats=True)
    assert float((y2.float() - ref2).abs().max()) <= 1.0e-2 * float(ref2.abs().max())
    tot = folded.view(256, 2, Cout).double().sum(0)
    yb = y2.float().double()
    np.testing.assert_allclose(tot[0].cpu().numpy(), yb.sum((0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-3 * float(yb.abs().max()))
    np.testing.assert_allclose(tot[1].cpu().numpy(), (yb * yb).sum((0, 2, 3)).cpu().numpy(), rtol=1e-4)
    # bias + ReLU, cropped output window
    bias = torch.randn(Cout, generator=g).to(cuda)
    ho, wo = ref.shape[2] - 1, ref.shape[3] - 1
    if ho >= 1 and wo >= 1:
        y3 = ops.conv_nhwc_s2(x, wp, Cout, k, bias=bias, relu=True, out_hw=(ho, wo))
        ref3 = F.relu((ref + bias.view(1, -1, 1, 1)).to(torch.bfloat16).float())[:, :, :ho, :wo]
        assert y3.shape == ref3.shape and float((y3.float() - ref3).abs().max()) <= 1.0e-2 * max(float(ref3.abs().max()), 1.0)


@pytest.fixture
def ws_mode():
    """islam_conv_ws_mode for the duration of a test (0: tile kernel, 2: weight-stationary kernel on every whole-tile 128 -> 128 layer)."""
    from islam_amd._lib import lib
    prev = lib().islam_conv_ws_mode(-1)
    yield lambda m: lib().islam_conv_ws_mode(m)
    lib().islam_conv_ws_mode(prev)


@pytest.mark.parametrize('C,CO,B,H,W', [(128, 128, 1, 4, 32), (128, 128, 2, 8, 64), (128, 128, 3, 12, 96), (128, 128, 5, 20, 160), (128, 128, 9, 112, 160),
                                        (64, 128, 1, 4, 32), (64, 128, 2, 8, 64), (64, 128, 3, 12, 96), (64, 128, 9, 112, 160),
                                        (32, 32, 1, 16, 32), (32, 32, 2, 32, 64), (32, 32, 3, 48, 96), (32, 32, 5, 224, 320)])
def test_persistent_kernels_match_the_tile_kernel_bit_for_bit(cuda, ws_mode, C, CO, B, H, W):
    """csrc/conv_ws.hip (128 -> 128 and 64 -> 128: one workgroup per CU walks a range of 32 x 4-pixel tiles with its weights in the register file) and
    csrc/conv_ws32.hip (32 -> 32: two persistent workgroups per CU, 32 x 16-pixel tiles, the next tile's halo requested a tile ahead)
    behind the three entry points that dispatch to them: the same bf16 outputs as conv_nhwc_kernel -- same operands, same accumulation
    order -- with and without the producer's BatchNorm + ReLU on load, from one tile (one workgroup) to several tiles per workgroup;
    the per-workgroup BatchNorm partial sums fold to the sums over the stored output; the same [scale | shift] and running statistics
    through the one-launch fold + finalize up to the summation order of the partial sums; a channel slice of a wider tensor as
    destination."""
    from islam_amd import ops
    x, w = _mk(B, C, H, W, CO, 3, seed=H + B)
    wp = ops.pack_conv_nhwc_weight(w)
    g = torch.Generator().manual_seed(6)
    aff = torch.cat((torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3)).to(cuda)
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    for in_affine in (None, aff):
        ws_mode(0)
        y0, f0 = ops.conv_nhwc(x, wp, CO, 3, in_affine=in_affine, stats=True)
        ws_mode(2)
        y2, f2 = ops.conv_nhwc(x, wp, CO, 3, in_affine=in_affine, stats=True)
        assert torch.equal(y0, y2)
        if in_affine is None:
            assert float((y2.float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
        s = f2.view(256, 2, CO).double().sum(0)
        want = torch.stack([y2.double().sum((0, 2, 3)), (y2.double() ** 2).sum((0, 2, 3))])
        np.testing.assert_allclose(s.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-3)
        _, f2b = ops.conv_nhwc(x, wp, CO, 3, in_affine=in_affine, stats=True)
        assert torch.equal(f2, f2b)                                           # fixed-order sums
        bn0, bn2 = torch.nn.BatchNorm2d(CO).to(cuda).train(), torch.nn.BatchNorm2d(CO).to(cuda).train()
        ws_mode(0)
        _, ss0 = ops.conv_nhwc_bn(x, wp, CO, 3, bn0, in_affine=in_affine)
        ws_mode(2)
        y2b, ss2 = ops.conv_nhwc_bn(x, wp, CO, 3, bn2, in_affine=in_affine)
        assert torch.equal(y2b, y2)
        torch.testing.assert_close(ss2, ss0, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(bn2.running_var, bn0.running_var, rtol=1e-5, atol=1e-7)
        out = torch.full((B, CO + 72, H, W), 7.0, dtype=torch.bfloat16, device=cuda).contiguous(memory_format=CL)
        ops.conv_nhwc_into(x, wp, CO, 3, out, 64, in_affine=in_affine)
        assert torch.equal(out[:, 64:64 + CO], y2) and bool((out[:, :64] == 7.0).all()) and bool((out[:, 64 + CO:] == 7.0).all())
    assert all(int(t[0].abs().sum()) == 0 for t in ops._BN_COUNTERS.values())


def test_persistent_kernels_serve_whole_tile_layers_only(cuda, ws_mode):
    """Images that are not whole tiles (32 x 4 pixels at 128 channels, 32 x 16 at 32), and epilogues the kernels do not have (bias,
    residual, ReLU), stay on the tile kernel whatever the mode says: same results with the mode on and off."""
    from islam_amd import ops
    ws_mode(2)
    for (C, B, H, W) in [(128, 2, 24, 40), (128, 1, 33, 47), (128, 1, 6, 32), (32, 2, 24, 64), (32, 1, 16, 40)]:
        x, w = _mk(B, C, H, W, C, 3, seed=W)
        wp = ops.pack_conv_nhwc_weight(w)
        ws_mode(2)
        y2 = ops.conv_nhwc(x, wp, C, 3)
        ws_mode(0)
        assert torch.equal(y2, ops.conv_nhwc(x, wp, C, 3))
    x, w = _mk(2, 128, 8, 64, 128, 3, seed=1)
    wp = ops.pack_conv_nhwc_weight(w)
    bias = torch.randn(128).to(cuda)
    res = torch.randn(2, 128, 8, 64).to(cuda).to(torch.bfloat16).contiguous(memory_format=CL)
    ws_mode(2)
    got = [ops.conv_nhwc(x, wp, 128, 3, relu=True), ops.conv_nhwc(x, wp, 128, 3, bias=bias, res=res), ops.conv_nhwc(x, wp, 128, 3, in_relu=True)]
    ws_mode(0)
    want = [ops.conv_nhwc(x, wp, 128, 3, relu=True), ops.conv_nhwc(x, wp, 128, 3, bias=bias, res=res), ops.conv_nhwc(x, wp, 128, 3, in_relu=True)]
    assert all(torch.equal(a, b) for a, b in zip(got, want))


def test_persistent_kernels_on_random_whole_tile_shapes(cuda, ws_mode):
    """Seeded sweep over shapes the fixed cases do not hit (odd tile counts per workgroup, one image row of tiles, many images of one tile,
    destinations with a channel offset): persistent kernel == tile kernel, bit for bit, and the folded statistics are the sums over the output."""
    from islam_amd import ops
    rng = np.random.default_rng(11)
    for case in range(14):
        C, CO, th = [(128, 128, 4), (64, 128, 4), (32, 32, 16)][case % 3]
        B = int(rng.integers(1, 7))
        H = th * int(rng.integers(1, 9 if th == 4 else 4))
        W = 32 * int(rng.integers(1, 5))
        x, w = _mk(B, C, H, W, CO, 3, seed=100 + case)
        wp = ops.pack_conv_nhwc_weight(w)
        aff = None
        if rng.integers(0, 2):
            aff = torch.tensor(np.concatenate([rng.uniform(0.5, 1.5, C), rng.normal(0, 0.3, C)]), dtype=torch.float32, device=cuda)
        ws_mode(0)
        y0, f0 = ops.conv_nhwc(x, wp, CO, 3, in_affine=aff, stats=True)
        ws_mode(2)
        y2, f2 = ops.conv_nhwc(x, wp, CO, 3, in_affine=aff, stats=True)
        assert torch.equal(y0, y2), (C, CO, B, H, W)
        s = f2.view(256, 2, CO).double().sum(0)
        want = torch.stack([y2.double().sum((0, 2, 3)), (y2.double() ** 2).sum((0, 2, 3))])
        np.testing.assert_allclose(s.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-3)
        yoff = 8 * int(rng.integers(0, 5))
        out = torch.full((B, CO + 40, H, W), 3.0, dtype=torch.bfloat16, device=cuda).contiguous(memory_format=CL)
        ops.conv_nhwc_into(x, wp, CO, 3, out, yoff, in_affine=aff)
        assert torch.equal(out[:, yoff:yoff + CO], y2) and bool((out[:, :yoff] == 3.0).all()) and bool((out[:, yoff + CO:] == 3.0).all())
