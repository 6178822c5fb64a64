"""GPU tests of the small channels-last bf16 kernels around the stereo net's convolutions: 2x2 max pooling (optionally of
relu(x)), k x k average pooling (the SPP branches) and the hourglass's up-sample + add -- against the torch ops they replace
(Network/PSM/hourglass.py:52-69, submodule.py:103-122, StereoNet7.py:117-125)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(shape, cuda, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(cuda, torch.bfloat16).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize('shape', [(2, 64, 56, 80), (3, 192, 14, 20), (1, 8, 7, 9), (2, 384, 2, 2)])
@pytest.mark.parametrize('relu', [False, True])
def test_maxpool2_is_exact(cuda, shape, relu):
    from islam_amd import ops
    x = _cl(shape, cuda, 1)
    y = ops.maxpool2(x, relu=relu)
    ref = F.max_pool2d(F.relu(x) if relu else x, kernel_size=2)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, ref)                                   # a maximum of bf16 values is one of them: bit-exact


@pytest.mark.parametrize('shape,k', [((2, 128, 112, 160), 8), ((2, 128, 14, 20), 2), ((1, 64, 17, 19), 2), ((2, 64, 64, 72), 64)])
def test_avgpool_matches_fp32_pooling(cuda, shape, k):
    from islam_amd import ops
    x = _cl(shape, cuda, 2)
    y = ops.avgpool(x, k)
    ref = F.avg_pool2d(x.float(), k, k)
    assert y.shape == ref.shape
    # fp32 accumulation, one rounding to bf16: within half a bf16 ulp (2^-9 relative) of the fp32 mean + summation-order noise
    err = (y.float() - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) <= 1e-6


@pytest.mark.parametrize('shape', [(2, 64, 28, 40), (1, 192, 7, 10), (3, 128, 1, 3)])
def test_upsample_add_equals_the_two_ops(cuda, shape):
    from islam_amd import ops
    B, C, h, w = shape
    low, u = _cl(shape, cuda, 3), _cl((B, C, 2 * h, 2 * w), cuda, 4)
    y = ops.resize_bilinear_add(low, u)
    two = u + ops.resize_bilinear(low, (2 * h, 2 * w), align_corners=False)
    assert torch.equal(y, two)                                   # same roundings: bf16 after the interpolation, bf16 after the add
    ref = u.float() + F.interpolate(low.float(), scale_factor=2, mode='bilinear', align_corners=False)
    assert float((y.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


@pytest.mark.parametrize('B,C,H,W', [(2, 6, 64, 96), (1, 6, 448, 640), (3, 8, 10, 6), (1, 2, 2, 2)])
def test_half_resolution_image_into_the_concatenation(cuda, B, C, H, W):
    """islam_half_image_into_nhwc_bf16 == F.interpolate(x, scale_factor=0.5, mode='bilinear') bit for bit, zero padded to 8 channels, in its
    slot of a larger channels-last tensor whose other channels stay untouched (Network/StereoNet7.py:101-105)."""
    from islam_amd import ops
    g = torch.Generator().manual_seed(B * 100 + H)
    x = (torch.randn(B, C, H, W, generator=g) * 3).to(cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out = torch.full((B, 24, H // 2, W // 2), 5.0, dtype=torch.bfloat16, device=cuda).contiguous(memory_format=torch.channels_last)
    ops.half_image_into(x, out, 8)
    want = F.interpolate(x, scale_factor=0.5, mode='bilinear')
    assert torch.equal(out[:, 8:8 + C], want)
    assert bool((out[:, 8 + C:16] == 0).all()) and bool((out[:, :8] == 5.0).all()) and bool((out[:, 16:] == 5.0).all())


@pytest.mark.parametrize('B,chans,tail,hw,align', [(2, (64, 128, 32, 32, 32, 32), 32, (14, 20), True), (1, (8, 16), 0, (9, 7), False),
                                                  (3, (24,), 8, (5, 11), True), (2, (32, 32, 32, 32, 32, 32, 32, 32), 16, (6, 6), False)])
def test_upsample_and_concatenate_in_one_launch(cuda, B, chans, tail, hw, align):
    """islam_upsample_cat_nhwc_bf16 == one islam_resize_bilinear_nhwc_bf16_into per piece + a copy of the tail (bit for bit): the
    `bigger` feature extractor's 352-channel input of lastconv (islam_amd/nets.py: feature_extraction.forward; reference:
    Network/StereoNet7.py:36-46, Network/PSM/submodule.py:139-152)."""
    from islam_amd import ops
    torch.manual_seed(3)
    cl = lambda t: t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    pieces = [cl(torch.randn(B, c, *hw, device=cuda)) for c in chans]
    size = (hw[0] * 2, hw[1] * 2)
    t = cl(torch.randn(B, tail, *size, device=cuda)) if tail else None
    got = ops.upsample_cat(pieces, size, tail=t, align_corners=align)
    ref = torch.empty_like(got)
    off = 0
    for p in pieces:
        ops.resize_bilinear_into(p, ref, off, align_corners=align)
        off += p.shape[1]
    if tail:
        ref[:, off:].copy_(t)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, ref)
    # ... and resize_bilinear itself is F.interpolate up to one bf16 rounding of the fp32 result
    want = torch.cat([torch.nn.functional.interpolate(p.float(), size, mode='bilinear', align_corners=align) for p in pieces], 1)
    assert (got[:, :off].float() - want).abs().max() <= 2 ** -7 * want.abs().max()


@pytest.mark.parametrize('B,C2,H,W', [(2, 6, 10, 14), (1, 2, 5, 3), (3, 16, 4, 6)])
def test_stereo_pair_stacked_and_padded_to_eight_channels(cuda, B, C2, H, W):
    """islam_stack_pair_pad8_nhwc_bf16: [left images; right images] of a channels-last pair with the channels zero-padded to eight -- the
    input of the stereo net's first layer on the channels-last kernel (islam_amd/nets.py: StereoNet7.forward; reference:
    Network/StereoNet7.py:95-97)."""
    from islam_amd import ops
    torch.manual_seed(5)
    x = torch.randn(B, C2, H, W, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = ops.stack_pair_pad8(x)
    c = C2 // 2
    want = torch.zeros(2 * B, 8, H, W, dtype=torch.bfloat16, device=cuda)
    want[:B, :c] = x[:, :c]
    want[B:, :c] = x[:, c:]
    assert y.shape == want.shape and y.is_contiguous(memory_format=torch.channels_last) and torch.equal(y, want)


@pytest.mark.parametrize('B,c,H,W', [(2, 3, 10, 14), (1, 1, 5, 3), (3, 4, 4, 6), (8, 3, 64, 96)])
def test_stereo_pair_prepared_from_the_fp32_images_in_one_pass(cuda, B, c, H, W):
    """islam_stereo_pair_prepare_f32 == torch.cat((left, right), 1).to(bfloat16) channels-last + islam_stack_pair_pad8_nhwc_bf16, bit for
    bit (Network/VONet.py:31-34 feeds the concatenated pair to the stereo net; values that round differently to nearest even / by
    truncation are in the sample)."""
    from islam_amd import ops
    g = torch.Generator(device=cuda).manual_seed(7 + B)
    left = torch.randn(B, c, H, W, device=cuda, generator=g)
    right = torch.randn(B, c, H, W, device=cuda, generator=g) * 3.0
    left.view(-1)[:4] = torch.tensor([1.00390625, 1.01171875, -0.0, float('inf')], device=cuda)      # ties, signed zero, infinity
    x6, xs = ops.stereo_pair_prepare(left, right)
    want6 = torch.cat((left, right), 1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert x6.shape == want6.shape and x6.is_contiguous(memory_format=torch.channels_last) and torch.equal(x6.view(torch.int16), want6.view(torch.int16))
    wants = torch.zeros(2 * B, 8, H, W, dtype=torch.bfloat16, device=cuda)
    wants[:B, :c] = want6[:, :c]
    wants[B:, :c] = want6[:, c:]
    assert xs.is_contiguous(memory_format=torch.channels_last) and torch.equal(xs.view(torch.int16), wants.contiguous(memory_format=torch.channels_last).view(torch.int16))
