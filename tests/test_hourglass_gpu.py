"""The fused hourglass Residual (islam_hg_residual_nhwc_bf16; reference Network/PSM/hourglass.py:28-52) against
(a) float64 torch on the same bf16 operands with the same rounding points and (b) the three-launch path it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mods(cin, cout, seed, dev):
    from islam_amd import nets
    torch.manual_seed(seed)
    m = nets._HGResidual(cin, cout)
    for c in (m.conv1, m.conv2, m.conv3, m.skip_layer):
        torch.nn.init.normal_(c.conv.bias, std=0.2)
    return m.to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last).eval()


def _ref64(m, x):
    """hourglass.py:41-52 in float64 on the bf16 operands, rounded where the bf16 execution copy rounds"""
    r = lambda t: t.to(torch.bfloat16).double()
    cv = lambda c, t, p: F.conv2d(t, c.conv.weight.double(), c.conv.bias.double(), padding=p)
    xx = x.double()
    res = r(cv(m.skip_layer, xx, 0)) if m.need_skip else xx
    t1 = F.relu(r(cv(m.conv1, F.relu(xx), 0)))
    t2 = F.relu(r(cv(m.conv2, t1, 1)))
    return r(r(cv(m.conv3, t2, 0)) + res)


@pytest.mark.parametrize('cin,cout,B,H,W', [(64, 64, 2, 24, 40), (128, 192, 1, 13, 21), (192, 192, 2, 14, 20), (256, 256, 2, 7, 10),
                                            (192, 256, 1, 9, 17), (128, 128, 1, 16, 16), (64, 64, 1, 50, 70), (256, 256, 1, 28, 40)])
def test_fused_residual_matches_float64_and_the_layerwise_path(cuda, cin, cout, B, H, W):
    from islam_amd import nets
    m = _mods(cin, cout, cin + cout + H, cuda)
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, cin, H, W, generator=g).to(cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    want = _ref64(m, x)
    old = nets.HG_FUSED
    try:
        nets.HG_FUSED = True
        with torch.no_grad():
            got = m(x)
        nets.HG_FUSED = False
        with torch.no_grad():
            layerwise = m(x)
    finally:
        nets.HG_FUSED = old
    assert got.shape == want.shape and got.dtype == torch.bfloat16 and got.is_contiguous(memory_format=torch.channels_last)
    scale = float(want.abs().max())
    for name, t in (('fused', got), ('layerwise', layerwise)):
        err = (t.double() - want).abs()
        # bf16 has 8 bits of mantissa: a result on the other side of a rounding boundary differs by one ulp of ITS magnitude; the
        # intermediates' boundary flips move the output by a few ulps of the output scale at isolated pixels
        assert float(err.max()) <= 2.0 ** -6 * scale, (name, float(err.max()), scale)
        assert float((err > 2.0 ** -8 * want.abs().clamp_min(2.0 ** -6 * scale)).double().mean()) < 0.02, name
    # fused vs layer-by-layer: same operands, same rounding points; only the fp32 summation order differs
    d = (got.double() - layerwise.double()).abs()
    assert float((d > 0).double().mean()) < 0.02 and float(d.max()) <= 2.0 ** -6 * scale


def test_zero_padding_of_the_middle_convolution(cuda):
    """conv2 pads ITS input (conv1's activated output) with zeros: a patch pixel outside the image must be 0, not relu(b1)."""
    from islam_amd import nets
    m = _mods(64, 64, 3, cuda)
    with torch.no_grad():
        m.conv1.conv.bias.fill_(1.0)                 # relu(b1) = 1 everywhere conv1 sees zeros
    x = torch.zeros(1, 64, 8, 16, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    want = _ref64(m, x)
    with torch.no_grad():
        got = m(x)
    assert float((got.double() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    assert float((want[0, :, 0, 0] - want[0, :, 4, 8]).abs().max()) > 1e-3      # (the border really differs from the interior)
