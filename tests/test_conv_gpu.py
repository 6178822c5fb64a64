"""3x3 convolution on the matrix cores (islam_conv3x3_mfma) vs torch: the kernel rounds its operands to bf16 and accumulates
in fp32, so the reference is F.conv2d in fp32 on bf16-rounded inputs and weights (only the summation order differs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin, H, W, Cout, stride, dilation
    (2, 6, 64, 96, 16, 2, 1),          # conv1a: first pyramid layer
    (2, 16, 32, 48, 16, 1, 1),
    (1, 117, 28, 40, 128, 1, 1),       # conv2_0-like: odd channel count
    (2, 245, 14, 20, 128, 1, 1),
    (1, 533, 7, 10, 32, 1, 1),
    (1, 565, 28, 40, 2, 1, 1),         # predict_flow: 2 output channels, no activation
    (1, 128, 28, 40, 128, 1, 2),       # context network: dilated
    (1, 128, 28, 40, 96, 1, 8),
    (1, 96, 37, 53, 196, 2, 1),        # ragged sizes, stride 2
    (3, 32, 9, 70, 64, 1, 4),
]


@pytest.mark.parametrize('B,Cin,H,W,Cout,S,D', CASES)
def test_conv3x3_mfma_matches_torch(cuda, B, Cin, H, W, Cout, S, D):
    from islam_amd import ops
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(cuda)
    b = torch.randn(Cout, generator=g).to(cuda)
    slope = 1.0 if Cout == 2 else 0.1
    y = ops.conv3x3_mfma(x, ops.pack_conv3x3_weight(w), b, Cout, stride=S, dilation=D, slope=slope)
    xr, wr = x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    ref = F.conv2d(xr.double(), wr.double(), b.double(), stride=S, padding=D, dilation=D)
    ref = torch.where(ref >= 0, ref, ref * slope).float()
    assert y.shape == ref.shape
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=2e-5 * float(ref.abs().max()))
    # against the unrounded fp32 convolution: bf16 operand rounding only (~2^-9 relative per product)
    full = F.conv2d(x, w, b, stride=S, padding=D, dilation=D)
    full = torch.where(full >= 0, full, full * slope)
    assert float((y - full).abs().max()) < 2e-2 * float(full.abs().max())


def test_conv3x3_mfma_writes_into_a_channel_slice(cuda):
    """`coff`: the DenseNet-style concatenation of PWCNet.py:237-292 without torch.cat."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 48, 20, 36, generator=g).to(cuda)
    w = (torch.randn(40, 48, 3, 3, generator=g) * 0.05).to(cuda)
    buf = torch.full((2, 100, 20, 36), 7.0, device=cuda)
    ops.conv3x3_mfma(x, ops.pack_conv3x3_weight(w), None, 40, out=buf, coff=25)
    ref = F.leaky_relu(F.conv2d(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), padding=1), 0.1)
    torch.testing.assert_close(buf[:, 25:65], ref, rtol=1e-4, atol=1e-4)
    assert bool((buf[:, :25] == 7.0).all()) and bool((buf[:, 65:] == 7.0).all())
    # input slice: the suffix [60, 100) of the buffer (here: 5 conv channels + 35 untouched ones) as a 40-channel input
    w2 = (torch.randn(8, 40, 3, 3, generator=g) * 0.05).to(cuda)
    src = buf.clone()
    ops.conv3x3_mfma(src, ops.pack_conv3x3_weight(w2), None, 8, out=buf, coff=52, xoff=60)
    ref2 = F.leaky_relu(F.conv2d(src[:, 60:].to(torch.bfloat16).float(), w2.to(torch.bfloat16).float(), padding=1), 0.1)
    torch.testing.assert_close(buf[:, 52:60], ref2, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('Hi,Wi,Ho,Wo,align', [(1, 2, 112, 160, True), (3, 5, 112, 160, True), (14, 20, 112, 160, True),
                                               (56, 80, 112, 160, True), (3, 5, 28, 40, False), (7, 10, 56, 80, False),
                                               (64, 96, 32, 48, False)])
def test_resize_bilinear_nhwc_bf16_matches_torch(cuda, Hi, Wi, Ho, Wo, align):
    """The SPP / decoder resizes of the frozen stereo net: ATen's upsample_bilinear2d arithmetic on channels-last bf16."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(Hi * 7 + Wo)
    x = torch.randn(3, 32, Hi, Wi, generator=g).to(cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = ops.resize_bilinear(x, (Ho, Wo), align)
    ref = F.interpolate(x.float(), [Ho, Wo], mode='bilinear', align_corners=align)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y.float(), ref, rtol=8e-3, atol=8e-3)           # one bf16 rounding of the result
    # other dtypes / layouts fall through to torch
    z = ops.resize_bilinear(x.float(), (Ho, Wo), align)
    torch.testing.assert_close(z, ref)


@pytest.mark.parametrize('relu,with_res', [(True, False), (False, True), (True, True), (False, False)])
def test_bias_act_add_epilogue_matches_torch(cuda, relu, with_res):
    """One-pass epilogue of the bias-free convolutions in the stereo net's execution copy vs the separate ATen ops."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(5)
    mk = lambda: torch.randn(3, 64, 19, 27, generator=g).to(cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, res = mk(), (mk() if with_res else None)
    bias = torch.randn(64, generator=g).to(cuda).to(torch.bfloat16)
    ref = y + bias.view(1, -1, 1, 1)
    if with_res:
        ref = ref + res
    if relu:
        ref = torch.relu(ref)
    got = ops.bias_act_add_(y.clone(memory_format=torch.channels_last), bias.float(), res, relu)
    assert torch.equal(got.float(), ref.float())


def test_hourglass_residual_fused_path_matches_plain(cuda):
    from islam_amd import nets
    torch.manual_seed(0)
    for cin, cout in ((64, 64), (64, 128)):
        m = nets._HGResidual(cin, cout).to(cuda).to(torch.bfloat16).to(memory_format=torch.channels_last)
        for p in m.parameters():
            p.requires_grad_(False)
        x = torch.randn(2, cin, 24, 40, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        fused = m(x)
        plain = m(x.contiguous())                    # NCHW-contiguous input: the unfused ATen path
        torch.testing.assert_close(fused.float(), plain.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('C,relu,with_res', [(32, True, False), (64, False, True), (128, True, False), (32, False, False)])
def test_fused_batchnorm_train_matches_torch(cuda, C, relu, with_res):
    """islam_bn_train_nhwc_bf16 vs nn.BatchNorm2d in training mode (+ add, + ReLU): output to bf16 accuracy, running
    statistics and the batch counter like torch's."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(C)
    mk = lambda: (torch.randn(4, C, 37, 53, generator=g) * 1.7 + 0.3).to(cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x, res = mk(), (mk() if with_res else None)
    ref_bn = torch.nn.BatchNorm2d(C).to(cuda).train()
    with torch.no_grad():
        ref_bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ref_bn.bias.copy_(torch.randn(C, generator=g))
    import copy
    hip_bn = copy.deepcopy(ref_bn)
    ref = ref_bn(x.float())
    if with_res:
        ref = ref + res.float()
    if relu:
        ref = torch.relu(ref)
    got = ops.bn_train_(x.clone(memory_format=torch.channels_last), hip_bn, relu, res)
    assert got.dtype == torch.bfloat16
    torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(hip_bn.running_mean, ref_bn.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(hip_bn.running_var, ref_bn.running_var, rtol=1e-4, atol=1e-5)
    assert int(hip_bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    # deterministic
    hip2 = copy.deepcopy(ref_bn)
    hip2.load_state_dict({k: v for k, v in torch.nn.BatchNorm2d(C).state_dict().items()}, strict=False)
    a = ops.bn_train_(x.clone(memory_format=torch.channels_last), copy.deepcopy(hip_bn), relu, res)
    b = ops.bn_train_(x.clone(memory_format=torch.channels_last), copy.deepcopy(hip_bn), relu, res)
    assert torch.equal(a, b)
