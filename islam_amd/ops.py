"""Thin tensor-level wrappers over the C ABI (one function per entry point of include/islam_hip.h).

PyTorch is used for device memory and the current HIP stream only.  All functions raise when given
CPU tensors: the product path has no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_double, c_float, c_int, c_int64, c_size_t, c_void_p, check, lib, ptr, require_cuda, stream_ptr


def _f32c(t):
    return t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t


# --------------------------------------------------------------------------- correlation / warp
def corr81_forward(first, second):
    """Network/PWC/correlation.py:281-331.  (B,C,H,W) x2 float32 -> (B,81,H,W)."""
    require_cuda(first, second)
    assert first.is_contiguous() and second.is_contiguous(), 'inputs must be contiguous NCHW (correlation.py:287-288)'
    assert first.dtype == torch.float32 and second.dtype == torch.float32 and first.shape == second.shape
    B, C, H, W = first.shape
    out = torch.empty((B, 81, H, W), dtype=torch.float32, device=first.device)
    nbytes = lib().islam_corr81_scratch_bytes(B, C, H, W)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=first.device) if nbytes else None
    check(lib().islam_corr81_fwd(ptr(first), ptr(second), ptr(out), B, C, H, W, ptr(scratch), stream_ptr(first.device)))
    return out


def corr81_act(first, second, out, coff, slope=0.1):
    """LeakyReLU_slope(corr81(first, second)) written into channels [coff, coff + 81) of ``out`` (B, Ctot, H, W) fp32 -- the head of
    PWC-Net's DenseNet concatenation without the activation and copy passes (islam_corr81_fwd_act).  Inference only."""
    require_cuda(first, second, out)
    assert first.is_contiguous() and second.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32
    assert first.dtype == torch.float32 and second.dtype == torch.float32 and first.shape == second.shape
    B, C, H, W = first.shape
    assert out.shape[0] == B and tuple(out.shape[2:]) == (H, W)
    nbytes = lib().islam_corr81_scratch_bytes(B, C, H, W)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=first.device) if nbytes else None
    check(lib().islam_corr81_fwd_act(ptr(first), ptr(second), ptr(out), int(out.shape[1]), int(coff), float(slope), B, C, H, W, ptr(scratch),
                                     stream_ptr(first.device)))
    return out


def corr81_backward(first, second, grad_out, need_first=True, need_second=True):
    """Network/PWC/correlation.py:334-383."""
    require_cuda(first, second, grad_out)
    assert grad_out.is_contiguous(), 'gradOutput must be contiguous (correlation.py:337)'
    B, C, H, W = first.shape
    g1 = torch.empty_like(first) if need_first else None
    g2 = torch.empty_like(second) if need_second else None
    check(lib().islam_corr81_bwd(ptr(first), ptr(second), ptr(grad_out), ptr(g1), ptr(g2), B, C, H, W,
                                 stream_ptr(first.device)))
    return g1, g2


class _Correlation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second):
        ctx.save_for_backward(first, second)
        return corr81_forward(first, second)

    @staticmethod
    def backward(ctx, grad_out):
        first, second = ctx.saved_tensors
        return corr81_backward(first, second, grad_out.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])


def FunctionCorrelation(tenFirst, tenSecond):
    """Drop-in for Network.PWC.correlation.FunctionCorrelation (correlation.py:386-388)."""
    return _Correlation.apply(tenFirst, tenSecond)


class _WarpMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow, scale):
        ctx.save_for_backward(x, flow)
        ctx.scale = float(scale)
        return warp_mask(x, flow, scale)

    @staticmethod
    def backward(ctx, g):
        x, flow = ctx.saved_tensors
        x, flow, g = _f32c(x), _f32c(flow), _f32c(g)
        B, C, H, W = x.shape
        gx, gflow = torch.zeros_like(x), torch.zeros_like(flow)
        check(lib().islam_warp_mask_bwd(ptr(x), ptr(flow), c_float(ctx.scale), ptr(g), ptr(gx), ptr(gflow), B, C, H, W,
                                        stream_ptr(x.device)))
        return gx, gflow, None


def warp(x, flow, scale=1.0):
    """Differentiable PWCDCNet.warp(x, flow*scale) (Network/PWC/PWCNet.py:170-206)."""
    return _WarpMask.apply(x, flow, scale)


def warp_mask(x, flow, scale=1.0):
    """PWCDCNet.warp(x, flow*scale) (Network/PWC/PWCNet.py:170-206), forward value only."""
    require_cuda(x, flow)
    x, flow = _f32c(x), _f32c(flow)
    B, C, H, W = x.shape
    assert flow.shape == (B, 2, H, W)
    out = torch.empty_like(x)
    check(lib().islam_warp_mask(ptr(x), ptr(flow), c_float(scale), ptr(out), B, C, H, W, stream_ptr(x.device)))
    return out


# --------------------------------------------------------------------------- edge mask
def edge_mask(img0, downscale=True, low=50, high=100):
    """TartanVO.py:145-155 (uint8 conversion, 1/4 resize, Canny(50, 100), 5x5 dilate) in one launch.
    (B,3,H,W) float32 in [0,1] -> (B,H/4,W/4) bool."""
    require_cuda(img0)
    assert img0.dim() == 4 and img0.shape[1] == 3, 'expects (B,3,H,W) images'
    img0 = _f32c(img0)
    B, _, H, W = img0.shape
    h, w = (H // 4, W // 4) if downscale else (H, W)
    out = torch.empty((B, h, w), dtype=torch.bool, device=img0.device)
    check(lib().islam_edge_mask(ptr(img0), ptr(out), B, H, W, int(bool(downscale)), int(low), int(high), stream_ptr(img0.device)))
    return out


# --------------------------------------------------------------------------- scale recovery
def scale_ls(disp, flow, pose7, intr4, baseline, edge, disp_th, depth_input=False):
    """Batched dense_ba.scale_from_disp_flow (dense_ba.py:88-176).  Returns scale (B), z (B,H,W),
    mask, depth_mask (B,H,W) bool, sums (B,18) float64.  depth_input: ``disp`` holds a depth map (the ``depth=`` branch,
    dense_ba.py:125-131; disp_th unused)."""
    require_cuda(disp, flow, pose7)
    dev = disp.device
    disp, flow = _f32c(disp), _f32c(flow)
    B, _, H, W = flow.shape
    pose7 = _f32c(pose7.detach())
    intr4 = _f32c(intr4.to(dev))
    baseline = _f32c(baseline.to(dev))
    disp_th = None if depth_input else _f32c(disp_th.to(dev))
    if edge is not None:
        edge = edge.to(dev).to(torch.uint8).contiguous()
    scale = torch.empty(B, dtype=torch.float32, device=dev)
    z = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    dmask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    sums = torch.empty((B, _lib.SCALE_NSUM), dtype=torch.float64, device=dev)
    partial = torch.empty((B, _lib.SCALE_NBLK, _lib.SCALE_NSUM), dtype=torch.float64, device=dev)
    if depth_input:
        check(lib().islam_scale_ls_depth(ptr(disp), ptr(flow), ptr(pose7), ptr(intr4), ptr(baseline), ptr(edge), ptr(scale),
                                         ptr(z), ptr(mask), ptr(dmask), ptr(sums), ptr(partial), B, H, W, stream_ptr(dev)))
    else:
        check(lib().islam_scale_ls(ptr(disp), ptr(flow), ptr(pose7), ptr(intr4), ptr(baseline), ptr(edge), ptr(disp_th),
                                   ptr(scale), ptr(z), ptr(mask), ptr(dmask), ptr(sums), ptr(partial), B, H, W,
                                   stream_ptr(dev)))
    return scale, z, mask.bool(), dmask.bool(), sums


# --------------------------------------------------------------------------- 3x3 convolution on the matrix cores
def deconv_to2(x, weight, bias, out=None, coff=0):
    """nn.ConvTranspose2d(C, 2, 4, 2, 1)(x) for fp32 NCHW tensors on islam_deconv4x4s2_to2_f32 (PWC-Net's deconv / upfeat layers), written
    into channels [coff, coff + 2) of ``out`` (B, ytot, 2H, 2W) (allocated with two channels when None)."""
    require_cuda(x, weight)
    B, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and tuple(weight.shape) == (C, 2, 4, 4) and weight.dtype == torch.float32
    if out is None:
        out = torch.empty((B, 2, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == B and tuple(out.shape[2:]) == (2 * H, 2 * W)
    check(lib().islam_deconv4x4s2_to2_f32(ptr(x), ptr(weight.contiguous()), ptr(bias), ptr(out), int(out.shape[1]), int(coff), B, C, H, W,
                                          stream_ptr(x.device)))
    return out


def flow_head_up(x, wf, bf, wu=None, bu=None, up_out=None, up_coff=0):
    """(Conv2d(C, 2, 3, 1, 1)(x), ConvTranspose2d(C, 2, 4, 2, 1)(x)) -- PWC-Net's predict_flow and upfeat of one level -- in one pass over
    x (islam_flow_head_up_f32).  wf: the Conv2d weight re-laid out as [C][2][3][3] (``weight.permute(1, 0, 2, 3).contiguous()``);
    wu / bu: the ConvTranspose2d's weight (C,2,4,4) / bias, or None for the head alone (returns (flow, None))."""
    require_cuda(x, wf)
    B, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and tuple(wf.shape) == (C, 2, 3, 3) and wf.is_contiguous() and wf.dtype == torch.float32
    flow = torch.empty((B, 2, H, W), dtype=torch.float32, device=x.device)
    up = None
    if wu is not None:
        assert tuple(wu.shape) == (C, 2, 4, 4) and wu.dtype == torch.float32 and wu.is_contiguous()
        if up_out is not None:              # into channels [up_coff, up_coff + 2) of a larger (B, Ctot, 2H, 2W) tensor
            assert up_out.dtype == torch.float32 and up_out.is_contiguous() and up_out.shape[0] == B and tuple(up_out.shape[2:]) == (2 * H, 2 * W)
            up = up_out
        else:
            up, up_coff = torch.empty((B, 2, 2 * H, 2 * W), dtype=torch.float32, device=x.device), 0
    check(lib().islam_flow_head_up_f32(ptr(x), ptr(wf), ptr(bf), ptr(flow), ptr(wu), ptr(bu), ptr(up), int(up.shape[1]) if up is not None else 0,
                                       int(up_coff), B, C, H, W, stream_ptr(x.device)))
    return flow, up


def pack_pyramid_weight(w):
    """(Cout, Cin, 3, 3) fp32 -> bf16 [Cout][ceil(9 * SC / 32) * 32] with K index = (ky * 3 + kx) * SC + c, SC = 4 for Cin <= 4 else Cin,
    zero padded: the layout islam_flow_pyramid_level holds in registers (include/islam_hip.h)."""
    Cout, Cin = int(w.shape[0]), int(w.shape[1])
    assert tuple(w.shape[2:]) == (3, 3)
    SC = 4 if Cin <= 4 else Cin
    K = (9 * SC + 31) // 32 * 32
    p = torch.zeros((Cout, K), dtype=torch.bfloat16, device=w.device)
    t = torch.zeros((Cout, 9, SC), dtype=torch.bfloat16, device=w.device)
    t[:, :, :Cin] = w.detach().permute(0, 2, 3, 1).reshape(Cout, 9, Cin).to(torch.bfloat16)
    p[:, :9 * SC] = t.reshape(Cout, 9 * SC)
    assert p.numel() == lib().islam_pyramid_packed_elems(Cin, Cout)
    return p.contiguous()


def flow_pyramid_level(x, packed, biases, slope=0.1):
    """conv(k3, s2) + conv(k3) + conv(k3), each + bias + LeakyReLU(slope), of one pyramid level in one launch (fp32 NCHW in and out, bf16
    operands).  packed / biases: the three layers' pack_pyramid_weight tensors / fp32 biases.  Inference only."""
    require_cuda(x, *packed)
    B, Cin, H, W = x.shape
    C = int(biases[0].numel())
    assert x.dtype == torch.float32 and x.is_contiguous() and len(packed) == 3 and len(biases) == 3
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    check(lib().islam_flow_pyramid_level(ptr(x), ptr(packed[0]), ptr(biases[0]), ptr(packed[1]), ptr(biases[1]), ptr(packed[2]), ptr(biases[2]),
                                         ptr(y), B, Cin, H, W, C, float(slope), stream_ptr(x.device)))
    return y


def flow_pyramid_level_pair(x, packed, biases, slope=0.1):
    """flow_pyramid_level (level 1: 3 -> 16 channels) on the two frames of a pair tensor x (B, 6, H, W) fp32: the result for
    torch.cat((x[:, :3], x[:, 3:]), 0) -- (2 B, 16, H/2, W/2), first frames first -- without that copy."""
    require_cuda(x, *packed)
    B, C6, H, W = x.shape
    assert C6 == 6 and x.dtype == torch.float32 and x.is_contiguous() and len(packed) == 3 and int(biases[0].numel()) == 16
    y = torch.empty((2 * B, 16, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    check(lib().islam_flow_pyramid_level_pair(ptr(x), ptr(packed[0]), ptr(biases[0]), ptr(packed[1]), ptr(biases[1]), ptr(packed[2]),
                                              ptr(biases[2]), ptr(y), B, H, W, float(slope), stream_ptr(x.device)))
    return y


def pack_conv3x3_weight(w):
    """(Cout, Cin, 3, 3) fp32 -> bf16 [9][CoutP][CinP] (tap-major, zero padded), the layout islam_conv3x3_mfma stages.
    The kernel walks the input channels in chunks of 16; when Cin > 16 is not a multiple of 16 its last chunk reads the LAST
    16 channels of the tensor (all exist), so that chunk's 16 weight slots hold channels Cin-16 .. Cin-1, zero for the ones
    the previous chunk already covered (include/islam_hip.h, islam_conv3x3_mfma)."""
    Cout, Cin = int(w.shape[0]), int(w.shape[1])
    assert tuple(w.shape[2:]) == (3, 3)
    CinP, CoutP = (Cin + 15) // 16 * 16, (Cout + 63) // 64 * 64
    p = torch.zeros((9, CoutP, CinP), dtype=torch.bfloat16, device=w.device)
    wt = w.detach().permute(2, 3, 0, 1).reshape(9, Cout, Cin).to(torch.bfloat16)
    full = Cin // 16 * 16
    if Cin == full or Cin < 16:
        p[:, :Cout, :Cin] = wt
    else:
        p[:, :Cout, :full] = wt[:, :, :full]
        p[:, :Cout, CinP - (Cin - full):] = wt[:, :, full:]
    assert p.numel() == lib().islam_conv3x3_packed_elems(Cin, Cout)
    return p.contiguous()


def conv3x3_mfma(x, packed, bias, cout, stride=1, dilation=1, slope=0.1, out=None, coff=0, xoff=0, cin=None):
    """y = LeakyReLU_slope(conv3x3(x[:, xoff:xoff+cin]) + bias) with padding = dilation (slope=1.0: linear).
    ``out`` (B, Ctot, Ho, Wo) fp32: the result goes to channels [coff, coff+cout); with ``xoff`` the input is a channel slice
    of a larger buffer -- together a DenseNet block needs no torch.cat.  Inference only (no autograd)."""
    require_cuda(x, packed)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, xtot, H, W = x.shape
    cin = xtot - xoff if cin is None else cin
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == B and tuple(out.shape[2:]) == (Ho, Wo)
    check(lib().islam_conv3x3_mfma(ptr(x), ptr(packed), ptr(bias), ptr(out), B, int(cin), H, W, int(cout), int(stride), int(dilation),
                                   int(xoff), int(xtot), int(coff), int(out.shape[1]), ctypes.c_float(slope), stream_ptr(x.device)))
    return out


def pack_conv_nhwc_weight(w):
    """(Cout, Cin, k, k) -> bf16 [k*k][CoutP][CinP] (tap-major, zero padded; CoutP = Cout up to 64s, CinP = Cin up to 32s), the
    layout islam_conv_nhwc_bf16 stages."""
    Cout, Cin, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
    assert w.shape[2] == w.shape[3] and k in (1, 2, 3)
    CinP, CoutP = (Cin + 31) // 32 * 32, (Cout + 63) // 64 * 64
    p = torch.zeros((k * k, CoutP, CinP), dtype=torch.bfloat16, device=w.device)
    p[:, :Cout, :Cin] = w.detach().permute(2, 3, 0, 1).reshape(k * k, Cout, Cin).to(torch.bfloat16)
    assert p.numel() == lib().islam_conv_nhwc_packed_elems(Cin, Cout, k)
    return p.contiguous()


def conv_nhwc(x, packed, cout, ksize, in_affine=None, bias=None, res=None, relu=False, stats=False, in_relu=False):
    """Channels-last bf16 convolution (k = 1 or 3, stride 1, "same" padding) on islam_conv_nhwc_bf16.  x: (B,Cin,H,W) bf16
    channels_last.  in_affine (2*Cin fp32): the producer's BatchNorm scale | shift, applied with a ReLU while x is staged.
    stats=True: returns (y, folded) with folded = the [256][2][cout] partial sums of the raw output for bn_finalize.
    in_relu: ReLU of the input while it is staged (no separate pass over x)."""
    require_cuda(x, packed)
    B, Cin, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    st = None
    if stats:
        st = torch.empty(lib().islam_conv_nhwc_stats_floats(B, H, W, cout), dtype=torch.float32, device=x.device)
    if res is not None:
        assert res.shape == y.shape and res.dtype == torch.bfloat16 and res.is_contiguous(memory_format=torch.channels_last)
    check(lib().islam_conv_nhwc_bf16(ptr(x), ptr(packed), ptr(in_affine), ptr(bias), ptr(res), ptr(y), ptr(st), B, Cin, H, W, int(cout),
                                     int(ksize), int(bool(relu)) | (2 if in_relu else 0), stream_ptr(x.device)))
    if stats:
        return y, st[st.numel() - 256 * 2 * cout:]
    return y


def conv_nhwc_s2(x, packed, cout, ksize, in_affine=None, bias=None, res=None, relu=False, stats=False, in_relu=False, out_hw=None, x2=None):
    """conv_nhwc at stride 2 with padding ksize // 2 (islam_conv_nhwc_bf16_s2; ksize 1, 2 or 3).  out_hw: fewer output rows / columns than
    the convolution has (the quarter-resolution tail keeps (H/2, W/2) of (H/2 + 1, W/2 + 1)).  x2 (ksize 2, bias / ReLU only): the input is
    torch.cat((x, x2), 1) read from the two tensors where they lie (islam_conv_nhwc_bf16_s2_cat; x.shape[1] % 16 == 0)."""
    require_cuda(x, packed, x2)
    B, Cin, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    P = ksize // 2
    Ho, Wo = ((H + 2 * P - ksize) // 2 + 1, (W + 2 * P - ksize) // 2 + 1) if out_hw is None else out_hw
    y = torch.empty((B, cout, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    if x2 is not None:
        assert in_affine is None and res is None and not stats and ksize == 2
        assert x2.dtype == torch.bfloat16 and x2.is_contiguous(memory_format=torch.channels_last) and x2.shape[0] == B and tuple(x2.shape[2:]) == (H, W)
        check(lib().islam_conv_nhwc_bf16_s2_cat(ptr(x), Cin, ptr(x2), int(x2.shape[1]), ptr(packed), ptr(bias), ptr(y), B, H, W, int(cout), Ho, Wo,
                                                int(ksize), int(bool(relu)) | (2 if in_relu else 0), stream_ptr(x.device)))
        return y
    st = None
    if stats:
        st = torch.empty(lib().islam_conv_nhwc_s2_stats_floats(B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    if res is not None:
        assert res.shape == y.shape and res.dtype == torch.bfloat16 and res.is_contiguous(memory_format=torch.channels_last)
    check(lib().islam_conv_nhwc_bf16_s2(ptr(x), ptr(packed), ptr(in_affine), ptr(bias), ptr(res), ptr(y), ptr(st), B, Cin, H, W, int(cout), Ho, Wo,
                                        int(ksize), int(bool(relu)) | (2 if in_relu else 0), stream_ptr(x.device)))
    if stats:
        return y, st[st.numel() - 256 * 2 * cout:]
    return y


def conv_nhwc_into(x, packed, cout, ksize, out, yoff, in_affine=None, bias=None, relu=False, in_relu=False):
    """conv_nhwc with the result written into out[:, yoff:yoff+cout] of a larger channels-last bf16 tensor (a concatenation under
    construction; islam_conv_nhwc_bf16_into).  Returns ``out``."""
    require_cuda(x, packed, out)
    B, Cin, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    assert out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=torch.channels_last) and out.shape[0] == B
    assert tuple(out.shape[2:]) == (H, W)
    check(lib().islam_conv_nhwc_bf16_into(ptr(x), ptr(packed), ptr(in_affine), ptr(bias), ptr(out), int(out.shape[1]), int(yoff), B, Cin, H, W,
                                          int(cout), int(ksize), int(bool(relu)) | (2 if in_relu else 0), stream_ptr(x.device)))
    return out


_BN_COUNTERS = {}


def _bn_counter(device):
    """A zero-initialised int32 slot for islam_conv_nhwc_bf16_bn's ticket (one ring of 256 per device: two launches that could run at
    the same time -- two streams, two nodes of a captured graph -- never share a slot; the kernel leaves its slot at zero)."""
    key = (device.type, device.index)
    hit = _BN_COUNTERS.get(key)
    if hit is None:
        hit = _BN_COUNTERS[key] = [torch.zeros(256 * 32, dtype=torch.int32, device=device), 0]      # one 128-byte line per slot
    hit[1] = (hit[1] + 1) % 256
    return hit[0][hit[1] * 32:]


def conv_nhwc_bn(x, packed, cout, ksize, bn, in_affine=None, in_relu=False):
    """conv_nhwc(..., stats=True) followed by bn_finalize as two launches instead of three (islam_conv_nhwc_bf16_bn): returns
    (raw output, [scale | shift] of the train-mode BatchNorm ``bn``, whose running statistics are updated).  Same bits."""
    require_cuda(x, packed)
    B, Cin, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and bn.num_features == cout
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    st = torch.empty(lib().islam_conv_nhwc_stats_floats(B, H, W, cout), dtype=torch.float32, device=x.device)
    out = torch.empty(2 * cout, dtype=torch.float32, device=x.device)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    check(lib().islam_conv_nhwc_bf16_bn(ptr(x), ptr(packed), ptr(in_affine), ptr(y), ptr(st), B, Cin, H, W, int(cout), int(ksize),
                                        int(bool(in_relu)), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean) if track else None,
                                        ptr(bn.running_var) if track else None, ptr(bn.num_batches_tracked) if track else None,
                                        c_double(mom), c_double(float(bn.eps)), ptr(out), ptr(_bn_counter(x.device)),
                                        stream_ptr(x.device)))
    return y, out


def pack_hg_residual(w1, w2, w3):
    """Weights of one hourglass Residual (hourglass.py:28-40: conv1 (h,Cin,1,1), conv2 (h,h,3,3), conv3 (Cout,h,1,1)) as the MFMA A
    fragments islam_hg_residual_nhwc_bf16 loads (1 KiB each: 64 lanes x 8 bf16), in the order its waves consume them
    (csrc/hourglass.hip)."""
    h, Cin, Cout = int(w1.shape[0]), int(w1.shape[1]), int(w3.shape[0])
    assert tuple(w2.shape) == (h, h, 3, 3) and int(w3.shape[1]) == h and Cout == 2 * h and h % 32 == 0 and Cin % 64 == 0
    MT = h // 32
    dev = w1.device
    bf = torch.bfloat16
    W1, W3 = w1.detach().reshape(h, Cin), w3.detach().reshape(Cout, h)
    W2 = w2.detach()
    # one MFMA A fragment = [64 lanes][8] bf16: lane (li = lane & 31, kg = lane >> 5) holds row r0 + li, columns k0 + 8 kg ... + 8
    li = torch.arange(64, device=dev) & 31
    kk = (torch.arange(64, device=dev) >> 5)[:, None] * 8 + torch.arange(8, device=dev)[None, :]      # (64, 8): 8 kg + j
    frag = lambda M, r0, k0: M[(r0 + li)[:, None], k0 + kk]
    if Cin == 64 and Cout == 64:
        # hg_residual64_kernel keeps the module's weights in registers: 26 fragments
        fr = [frag(W1, 0, 16 * ks) for ks in range(4)]
        fr += [frag(W2[:, :, t // 3, t % 3], 0, 16 * ks) for t in range(9) for ks in range(2)]
        fr += [frag(W3, 32 * a, 16 * ks) for a in range(2) for ks in range(2)]
    else:
        # hg_residual_kernel: every wave streams the fragments of ITS output-channel tile in consumption order
        fr = [frag(W1, 32 * mg, 16 * ks) for mg in range(MT) for ks in range(Cin // 16)]
        fr += [frag(W2[:, :, t // 3, t % 3], 32 * mg, 32 * c + 16 * ks) for mg in range(MT) for c in range(MT) for t in range(9) for ks in range(2)]
        fr += [frag(W3, 32 * m3, 16 * ks) for m3 in range(2 * MT) for ks in range(h // 16)]
    packed = torch.stack(fr).to(bf).reshape(-1).contiguous()
    assert packed.numel() == lib().islam_hg_residual_packed_elems(Cin, Cout), (packed.numel(), Cin, Cout)
    return packed


def hg_residual(x, packed, bias32, cout, res=None):
    """conv3(relu(conv2(relu(conv1(relu(x)))))) + res of a channels-last bf16 tensor in one launch (islam_hg_residual_nhwc_bf16);
    res = x when None (needs Cin == cout).  bias32 = cat(b1, b2, b3) fp32."""
    require_cuda(x, packed)
    B, Cin, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and bias32.dtype == torch.float32
    assert bias32.numel() == 2 * cout
    if res is None:
        assert Cin == cout
        res = x
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    assert res.shape == y.shape and res.dtype == torch.bfloat16 and res.is_contiguous(memory_format=torch.channels_last)
    check(lib().islam_hg_residual_nhwc_bf16(ptr(x), ptr(res), ptr(y), ptr(packed), ptr(bias32), B, Cin, H, W, int(cout), stream_ptr(x.device)))
    return y


def pack_deconv_nhwc_weight(w):
    """ConvTranspose2d weight (Cin, Cout, 4, 4) -> bf16 [4 classes][4 taps][CoutP][CinP] for islam_deconv4x4s2_nhwc_bf16: output parity
    class (a, c), tap (r, s) of its 2x2 convolution = W[:, :, 3 - 2r - a, 3 - 2s - c] (stride 2, padding 1)."""
    Cin, Cout = int(w.shape[0]), int(w.shape[1])
    assert tuple(w.shape[2:]) == (4, 4)
    CinP, CoutP = (Cin + 31) // 32 * 32, (Cout + 63) // 64 * 64
    p = torch.zeros((4, 4, CoutP, CinP), dtype=torch.bfloat16, device=w.device)
    wt = w.detach().permute(2, 3, 1, 0).to(torch.bfloat16)            # (ky, kx, Cout, Cin)
    for a in range(2):
        for c in range(2):
            for r in range(2):
                for s_ in range(2):
                    p[a * 2 + c, r * 2 + s_, :Cout, :Cin] = wt[3 - 2 * r - a, 3 - 2 * s_ - c]
    assert p.numel() == lib().islam_deconv_nhwc_packed_elems(Cin, Cout)
    return p.contiguous()


def deconv_nhwc(x, packed, bias32, cout, out=None, yoff=0, relu=False, x2=None):
    """act(ConvTranspose2d(k=4, s=2, p=1)(x) + bias) of a channels-last bf16 tensor on islam_deconv4x4s2_nhwc_bf16, written into
    channels [yoff, yoff + cout) of ``out`` (B, ytot, 2H, 2W) channels_last bf16 (allocated dense when None).  x2: the layer's input is
    torch.cat((x, x2), 1), read from the two tensors where they lie (islam_deconv4x4s2_nhwc_bf16_cat; x.shape[1] % 32 == 0)."""
    require_cuda(x, packed, x2)
    B, Cin, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and bias32.dtype == torch.float32
    if out is None:
        out = torch.empty((B, cout, 2 * H, 2 * W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    assert out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=torch.channels_last)
    assert out.shape[0] == B and tuple(out.shape[2:]) == (2 * H, 2 * W)
    if x2 is not None:
        assert x2.dtype == torch.bfloat16 and x2.is_contiguous(memory_format=torch.channels_last) and x2.shape[0] == B and tuple(x2.shape[2:]) == (H, W)
        check(lib().islam_deconv4x4s2_nhwc_bf16_cat(ptr(x), Cin, ptr(x2), int(x2.shape[1]), ptr(packed), ptr(bias32), ptr(out), int(out.shape[1]),
                                                    int(yoff), B, H, W, int(cout), int(bool(relu)), stream_ptr(x.device)))
        return out
    check(lib().islam_deconv4x4s2_nhwc_bf16(ptr(x), ptr(packed), ptr(bias32), ptr(out), int(out.shape[1]), int(yoff), B, Cin, H, W, int(cout),
                                            int(bool(relu)), stream_ptr(x.device)))
    return out


def bn_finalize(folded, bn, count):
    """[256][2][C] partial sums -> (2*C) fp32 [scale | shift] of a train-mode nn.BatchNorm2d; its running statistics are
    updated like nn.BatchNorm2d would (islam_bn_finalize)."""
    C = bn.num_features
    out = torch.empty(2 * C, dtype=torch.float32, device=folded.device)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    check(lib().islam_bn_finalize(ptr(folded), c_double(float(count)), ptr(bn.weight), ptr(bn.bias),
                                  ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                  ptr(bn.num_batches_tracked) if track else None, c_double(mom), c_double(float(bn.eps)), C, ptr(out),
                                  stream_ptr(folded.device)))
    return out


def bn_apply_(x, scale_shift, relu=False, res=None):
    """In place on a channels-last bf16 tensor: x <- act(bf16(x*scale[c] + shift[c]) [+ res]) (islam_bn_apply_nhwc_bf16)."""
    B, C, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    if res is not None:
        assert res.shape == x.shape and res.dtype == torch.bfloat16 and res.is_contiguous(memory_format=torch.channels_last)
    check(lib().islam_bn_apply_nhwc_bf16(ptr(x), ptr(x), ptr(res), ptr(scale_shift), int(bool(relu)), ctypes.c_longlong(B * H * W), C,
                                         stream_ptr(x.device)))
    return x


def resize_bilinear(x, size, align_corners=False):
    """F.interpolate(x, size, mode='bilinear', align_corners=...) -- on the HIP kernel for channels-last bf16 inference
    tensors (the frozen stereo net's execution copy), through torch otherwise."""
    Ho, Wo = int(size[0]), int(size[1])
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0 and not x.requires_grad
            and x.is_contiguous(memory_format=torch.channels_last)):
        B, C, Hi, Wi = x.shape
        y = torch.empty((B, C, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        check(lib().islam_resize_bilinear_nhwc_bf16(ptr(x), ptr(y), B, C, Hi, Wi, Ho, Wo, int(bool(align_corners)),
                                                    stream_ptr(x.device)))
        return y
    return torch.nn.functional.interpolate(x, [Ho, Wo], mode='bilinear', align_corners=align_corners)


def resize_bilinear_into(x, out, coff, align_corners=False):
    """Bilinear resize of a channels-last bf16 tensor to out's spatial size, written into out[:, coff:coff+C] (out: channels-last
    bf16 with a multiple of 8 channels; coff a multiple of 8).  islam_resize_bilinear_nhwc_bf16_into."""
    B, C, Hi, Wi = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    assert out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=torch.channels_last) and out.shape[0] == B
    check(lib().islam_resize_bilinear_nhwc_bf16_into(ptr(x), ptr(out), B, C, Hi, Wi, int(out.shape[2]), int(out.shape[3]),
                                                     int(bool(align_corners)), int(out.shape[1]), int(coff), stream_ptr(x.device)))
    return out


def stack_pair_pad8(x):
    """(B, 2c, H, W) channels-last bf16 stereo pair -> (2B, 8, H, W): left images, then right images, channels [c, 8) zero
    (islam_stack_pair_pad8_nhwc_bf16)."""
    B, C2, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and C2 % 2 == 0 and C2 <= 16
    y = torch.empty((2 * B, 8, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    check(lib().islam_stack_pair_pad8_nhwc_bf16(ptr(x), ptr(y), B, C2, H, W, stream_ptr(x.device)))
    return y


def stereo_pair_prepare(left, right):
    """The stereo pair (two fp32 (B,c,H,W) tensors, c <= 4) as the frozen stereo net's execution copy reads it, in one pass
    (islam_stereo_pair_prepare_f32): (x6, xs) with x6 = torch.cat((left, right), 1).to(bfloat16) channels-last and xs = stack_pair_pad8(x6)."""
    require_cuda(left, right)
    B, c, H, W = left.shape
    assert left.dtype == torch.float32 and right.dtype == torch.float32 and left.shape == right.shape and c <= 4
    assert left.is_contiguous() and right.is_contiguous()
    x6 = torch.empty((B, 2 * c, H, W), dtype=torch.bfloat16, device=left.device, memory_format=torch.channels_last)
    xs = torch.empty((2 * B, 8, H, W), dtype=torch.bfloat16, device=left.device, memory_format=torch.channels_last)
    check(lib().islam_stereo_pair_prepare_f32(ptr(left), ptr(right), ptr(x6), ptr(xs), B, c, H, W, stream_ptr(left.device)))
    return x6, xs


def upsample_cat(pieces, size, tail=None, align_corners=False):
    """torch.cat([F.interpolate(p, size, mode='bilinear') for p in pieces] + [tail], 1) for channels-last bf16 pieces of one shape
    (B, C_k, Hi, Wi), C_k multiples of 8, at most 8 of them; tail: (B, C_t, *size) or None.  One launch
    (islam_upsample_cat_nhwc_bf16); every value is the one resize_bilinear produces."""
    import ctypes
    B, _, Hi, Wi = pieces[0].shape
    Ho, Wo = int(size[0]), int(size[1])
    for t in pieces:
        assert t.dtype == torch.bfloat16 and t.is_contiguous(memory_format=torch.channels_last) and t.shape[0] == B and tuple(t.shape[2:]) == (Hi, Wi)
        assert t.shape[1] % 8 == 0
    ct = 0
    if tail is not None:
        assert tail.dtype == torch.bfloat16 and tail.is_contiguous(memory_format=torch.channels_last) and tuple(tail.shape[2:]) == (Ho, Wo)
        assert tail.shape[0] == B and tail.shape[1] % 8 == 0
        ct = int(tail.shape[1])
    n = len(pieces)
    assert 1 <= n <= 8
    out = torch.empty((B, sum(int(t.shape[1]) for t in pieces) + ct, Ho, Wo), dtype=torch.bfloat16, device=pieces[0].device,
                      memory_format=torch.channels_last)
    srcs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in pieces])
    chans = (ctypes.c_int * n)(*[int(t.shape[1]) for t in pieces])
    check(lib().islam_upsample_cat_nhwc_bf16(srcs, chans, n, ptr(tail) if tail is not None else None, ct, ptr(out), B, Hi, Wi, Ho, Wo,
                                             int(bool(align_corners)), stream_ptr(out.device)))
    return out


def nchw_to_nhwc_mirror(src, soff, C, dst, doff):
    """fp32 NCHW channels [soff, soff+C) of ``src`` -> bf16 channels [doff, doff+C) of the channels-last tensor ``dst`` (a
    (B, Ctot, H, W) bf16 tensor in torch.channels_last memory format); the channels up to the next multiple of 8 are zeroed."""
    B, stot, H, W = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.dtype == torch.bfloat16
    assert dst.is_contiguous(memory_format=torch.channels_last) and dst.shape[0] == B and tuple(dst.shape[2:]) == (H, W)
    check(lib().islam_nchw_f32_to_nhwc_bf16(ptr(src), stot, int(soff), ptr(dst), int(dst.shape[1]), int(doff), B, int(C), H, W,
                                            stream_ptr(src.device)))
    return dst


def conv_nhwc_flow(xm, xoff, cin, packed, bias, y32, coff, cout, slope, ymir=None, moff=0, dilation=1):
    """islam_conv_nhwc_flow: act(conv3x3(xm[:, xoff:xoff+cin], dilation) + bias) -> y32[:, coff:coff+cout] (fp32 NCHW, may be None)
    and / or ymir[:, moff:moff+cout] (bf16 channels-last mirror).  xm / ymir: bf16 tensors in torch.channels_last memory format."""
    B, xtot, H, W = xm.shape
    assert xm.dtype == torch.bfloat16 and xm.is_contiguous(memory_format=torch.channels_last)
    assert y32 is not None or ymir is not None
    if y32 is not None:
        assert y32.dtype == torch.float32 and y32.is_contiguous() and tuple(y32.shape[2:]) == (H, W) and y32.shape[0] == B
    assert packed.dtype == torch.bfloat16 and packed.numel() == lib().islam_conv_nhwc_packed_elems(int(cin), int(cout), 3)
    if ymir is not None:
        assert ymir.dtype == torch.bfloat16 and ymir.is_contiguous(memory_format=torch.channels_last) and ymir.shape[0] == B
        assert tuple(ymir.shape[2:]) == (H, W)
    check(lib().islam_conv_nhwc_flow(ptr(xm), int(xtot), int(xoff), int(cin), ptr(packed), ptr(bias), ptr(y32),
                                     int(y32.shape[1]) if y32 is not None else 0, int(coff), ptr(ymir),
                                     int(ymir.shape[1]) if ymir is not None else 0, int(moff), B, H, W, int(cout), int(dilation),
                                     c_float(slope), stream_ptr(xm.device)))
    return y32 if y32 is not None else ymir


def resize_bilinear_add(x, add, align_corners=False, out=None, coff=0):
    """add + F.interpolate(x, add's size, mode='bilinear') in one pass over channels-last bf16 tensors
    (islam_resize_bilinear_add_nhwc_bf16: the hourglass's `up1 + up2(low)`).  ``out``: write into out[:, coff:coff+C] of a larger
    channels-last bf16 tensor (a concatenation under construction) and return ``out``."""
    B, C, Hi, Wi = x.shape
    assert fusable_nhwc_bf16(x, C) and fusable_nhwc_bf16(add, C) and add.shape[:2] == x.shape[:2]
    if out is not None:
        assert out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=torch.channels_last) and out.shape[0] == B
        assert tuple(out.shape[2:]) == tuple(add.shape[2:])
        check(lib().islam_resize_bilinear_add_nhwc_bf16_into(ptr(x), ptr(add), ptr(out), B, C, Hi, Wi, int(add.shape[2]), int(add.shape[3]),
                                                             int(bool(align_corners)), int(out.shape[1]), int(coff), stream_ptr(x.device)))
        return out
    y = torch.empty_like(add)
    check(lib().islam_resize_bilinear_add_nhwc_bf16(ptr(x), ptr(add), ptr(y), B, C, Hi, Wi, int(add.shape[2]), int(add.shape[3]),
                                                    int(bool(align_corners)), stream_ptr(x.device)))
    return y


def half_image_into(x, out, yoff):
    """F.interpolate(x, scale_factor=0.5, mode='bilinear') of a channels-last bf16 image with at most 8 channels, written (zero padded to
    8 channels) into out[:, yoff:yoff+8] of a larger channels-last bf16 tensor (islam_half_image_into_nhwc_bf16).  Returns ``out``."""
    require_cuda(x, out)
    B, C, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    assert out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=torch.channels_last)
    assert tuple(out.shape[2:]) == (H // 2, W // 2) and out.shape[0] == B
    check(lib().islam_half_image_into_nhwc_bf16(ptr(x), ptr(out), int(out.shape[1]), int(yoff), B, C, H, W, stream_ptr(x.device)))
    return out


def maxpool2(x, relu=False):
    """F.max_pool2d([relu](x), 2): islam_maxpool2_nhwc_bf16 for channels-last bf16 inference tensors, torch otherwise."""
    if fusable_nhwc_bf16(x, x.shape[1]) and x.shape[2] >= 2 and x.shape[3] >= 2:
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        check(lib().islam_maxpool2_nhwc_bf16(ptr(x), ptr(y), B, C, H, W, int(bool(relu)), stream_ptr(x.device)))
        return y
    return torch.nn.functional.max_pool2d(torch.relu(x) if relu else x, kernel_size=2)


def avgpool(x, k):
    """AvgPool2d((k, k), stride=(k, k)) of a channels-last bf16 inference tensor (islam_avgpool_nhwc_bf16)."""
    B, C, H, W = x.shape
    assert fusable_nhwc_bf16(x, C) and H >= k and W >= k
    y = torch.empty((B, C, H // k, W // k), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    check(lib().islam_avgpool_nhwc_bf16(ptr(x), ptr(y), B, C, H, W, int(k), stream_ptr(x.device)))
    return y


def fusable_nhwc_bf16(x, channels):
    """True for the tensors the channels-last bf16 epilogue / resize kernels take (frozen execution copies, no autograd)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and channels % 8 == 0 and not x.requires_grad
            and x.is_contiguous(memory_format=torch.channels_last))


def bias_act_add_(y, bias32, res=None, relu=False):
    """In place on a channels-last bf16 conv output: y <- act(bf16(y + bias) [+ res]) (islam_bias_act_add_nhwc_bf16)."""
    B, C, H, W = y.shape
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last) and bias32.dtype == torch.float32
    if res is not None:
        assert res.shape == y.shape and res.dtype == torch.bfloat16 and res.is_contiguous(memory_format=torch.channels_last)
    check(lib().islam_bias_act_add_nhwc_bf16(ptr(y), ptr(bias32), ptr(res), ctypes.c_longlong(B * H * W), C, int(bool(relu)),
                                             stream_ptr(y.device)))
    return y


class _BiasActF32(torch.autograd.Function):
    """y = act(x + bias[c] (+ res)) on fp32 channels-last conv outputs, one launch each way (islam_bias_act_f32_nhwc /
    islam_bias_act_bwd_f32_nhwc): the elementwise tail of the trainable pose head's convolutions (Network/VOFlowNet.py:42-157)."""

    @staticmethod
    def forward(ctx, x, bias, res, relu):
        B, C, H, W = x.shape
        y = torch.empty_like(x, memory_format=torch.channels_last)
        check(lib().islam_bias_act_f32_nhwc(ptr(x), ptr(bias), ptr(res), ptr(y), ctypes.c_longlong(B * H * W), C, int(bool(relu)),
                                            stream_ptr(x.device)))
        ctx.relu, ctx.has_res = bool(relu), res is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        B, C, H, W = y.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(gy, memory_format=torch.channels_last)
        gb = torch.empty(C, dtype=torch.float32, device=y.device)
        npix = B * H * W
        scratch = torch.empty(int(lib().islam_bias_act_bwd_scratch_floats(ctypes.c_longlong(npix), C)), dtype=torch.float32, device=y.device)
        check(lib().islam_bias_act_bwd_f32_nhwc(ptr(gy), ptr(y), ptr(gx), ptr(gb), ptr(scratch), ptr(_ticket(y.device)),
                                                ctypes.c_longlong(npix), C, int(ctx.relu), stream_ptr(y.device)))
        return gx, gb, (gx if ctx.has_res else None), None


_TICKETS = {}


def _ticket(device):
    """One zero-initialised word per device for the kernels that fold partial sums in their last workgroup (they leave it at zero;
    launches on one stream are ordered, so they can share it)."""
    t = _TICKETS.get(device)
    if t is None:
        t = _TICKETS[device] = torch.zeros(4, dtype=torch.int32, device=device)
    return t


def bias_act(x, bias, res=None, relu=True):
    """act(x + bias[c] (+ res)) for an fp32 channels-last conv output, differentiable w.r.t. x, bias and res."""
    assert x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 4 == 0
    if res is not None:
        assert res.shape == x.shape and res.dtype == torch.float32 and res.is_contiguous(memory_format=torch.channels_last)
    return _BiasActF32.apply(x, bias, res, relu)


def bn_train_(x, bn, relu=False, res=None):
    """nn.BatchNorm2d in training mode on a channels-last bf16 conv output, IN PLACE, with the ReLU and / or residual add
    that follows it folded in (islam_bn_train_nhwc_bf16).  ``bn``: the fp32 BatchNorm2d module (weight, bias, running
    statistics, momentum, eps); its buffers are updated like nn.BatchNorm2d would."""
    B, C, H, W = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last)
    if res is not None:
        assert res.shape == x.shape and res.dtype == torch.bfloat16 and res.is_contiguous(memory_format=torch.channels_last)
    scratch = torch.empty(lib().islam_bn_scratch_floats(C), dtype=torch.float32, device=x.device)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    check(lib().islam_bn_train_nhwc_bf16(ptr(x), ptr(x), ptr(res), ptr(bn.weight), ptr(bn.bias),
                                         ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                         ptr(bn.num_batches_tracked) if track else None, c_double(mom), c_double(bn.eps),
                                         int(bool(relu)), ctypes.c_longlong(B * H * W), C, ptr(scratch), stream_ptr(x.device)))
    return x


# --------------------------------------------------------------------------- IMU
def _imu_preint_raw(dt, gyro, acc, seg, seg_host, init_pos, init_rot, init_vel, gravity, motion_mode):
    require_cuda(dt, gyro, acc, seg)
    dtype = dt.dtype
    code = {torch.float32: 0, torch.float64: 1}[dtype]
    dev = dt.device
    nframes = int(seg_host.shape[0]) - 1
    S = int(dt.shape[0])
    maxF = int(np.max(np.diff(seg_host))) if nframes > 0 else 0
    rows = nframes if motion_mode else nframes + 1
    pos = torch.empty((rows, 3), dtype=dtype, device=dev)
    rot = torch.empty((rows, 4), dtype=dtype, device=dev)
    vel = torch.empty((rows, 3), dtype=dtype, device=dev)
    nbytes = lib().islam_imu_scratch_bytes(S, nframes, code)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().islam_imu_preint(ptr(dt), ptr(gyro), ptr(acc), ptr(seg), nframes, S, maxF, ptr(init_pos), ptr(init_rot),
                                 ptr(init_vel), c_double(gravity), 1 if motion_mode else 0, ptr(pos), ptr(rot), ptr(vel),
                                 ptr(scratch), code, stream_ptr(dev)))
    return pos, rot, vel, scratch


def imu_preint_both(dt, gyro, acc, seg, seg_host, init_pos, init_rot, init_vel, gravity):
    """World-mode and motion-mode outputs of one frame range from ONE pass (islam_imu_preint_both): ((pos, rot, vel) with
    nframes + 1 rows, (pos, rot, vel) with nframes rows), bit-identical to two imu_preint calls.  Forward values only."""
    require_cuda(dt, gyro, acc, seg)
    dtype = dt.dtype
    code = {torch.float32: 0, torch.float64: 1}[dtype]
    dev = dt.device
    nframes = int(seg_host.shape[0]) - 1
    S = int(dt.shape[0])
    maxF = int(np.max(np.diff(seg_host))) if nframes > 0 else 0
    # one allocation for the six outputs: [world pos | rot | vel | motion pos | rot | vel], 10 columns per row
    out = torch.empty((2 * nframes + 1) * 10, dtype=dtype, device=dev)
    w, m = nframes + 1, nframes
    views, o = [], 0
    for rows, cols in ((w, 3), (w, 4), (w, 3), (m, 3), (m, 4), (m, 3)):
        views.append(out[o:o + rows * cols].view(rows, cols))
        o += rows * cols
    nbytes = lib().islam_imu_scratch_bytes(S, nframes, code)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().islam_imu_preint_both(ptr(dt), ptr(gyro), ptr(acc), ptr(seg), nframes, S, maxF, ptr(init_pos), ptr(init_rot),
                                      ptr(init_vel), c_double(gravity), *[ptr(v) for v in views], ptr(scratch), code, stream_ptr(dev)))
    return tuple(views[:3]), tuple(views[3:]), out


class _ImuPreint(torch.autograd.Function):
    """islam_imu_preint with islam_imu_preint_bwd: gradients w.r.t. the gyro / accelerometer samples (the denoiser's outputs
    in the IMU-target epoch).  The gradient of the quaternion output is read in PyPose's convention (left tangent, padded)."""

    @staticmethod
    def forward(ctx, gyro, acc, dt, seg, seg_host, init_pos, init_rot, init_vel, gravity, motion_mode):
        pos, rot, vel, scratch = _imu_preint_raw(dt, gyro, acc, seg, seg_host, init_pos, init_rot, init_vel, gravity, motion_mode)
        ctx.save_for_backward(gyro, acc, dt, seg, scratch)
        ctx.meta = (int(seg_host.shape[0]) - 1, float(gravity), bool(motion_mode))
        return pos, rot, vel

    @staticmethod
    def backward(ctx, g_pos, g_rot, g_vel):
        gyro, acc, dt, seg, scratch = ctx.saved_tensors
        nframes, gravity, motion = ctx.meta
        code = {torch.float32: 0, torch.float64: 1}[dt.dtype]
        c = lambda g: None if g is None else g.to(dt.dtype).contiguous()
        g_pos, g_rot, g_vel = c(g_pos), c(g_rot), c(g_vel)
        g_gyro, g_acc = torch.zeros_like(gyro), torch.zeros_like(acc)
        ws = torch.empty(lib().islam_imu_preint_bwd_scratch_bytes(nframes), dtype=torch.uint8, device=dt.device)
        check(lib().islam_imu_preint_bwd(ptr(dt), ptr(gyro), ptr(acc), ptr(seg), nframes, int(dt.shape[0]), c_double(gravity),
                                         1 if motion else 0, ptr(scratch), ptr(g_pos), ptr(g_rot), ptr(g_vel), ptr(g_gyro), ptr(g_acc),
                                         ptr(ws), code, stream_ptr(dt.device)))
        return g_gyro, g_acc, None, None, None, None, None, None, None, None


def imu_preint(dt, gyro, acc, seg, seg_host, init_pos, init_rot, init_vel, gravity, motion_mode):
    """One IMUModule.integrate frame loop (imu_integrator.py:116-158).  dt (S), gyro/acc (S,3) on device,
    float32 or float64; seg int64 device tensor (nframes+1), seg_host the same on the host.  Differentiable w.r.t. gyro / acc
    when either requires grad."""
    if torch.is_grad_enabled() and (gyro.requires_grad or acc.requires_grad):
        return _ImuPreint.apply(gyro, acc, dt, seg, seg_host, init_pos, init_rot, init_vel, gravity, motion_mode)
    return _imu_preint_raw(dt, gyro, acc, seg, seg_host, init_pos, init_rot, init_vel, gravity, motion_mode)[:3]


# --------------------------------------------------------------------------- PVGO
def pvgo_default_params(loss_weight=(1, 1, 1, 1), radius=1e4, seg_len=(0, 0)):
    p = _lib.PvgoParams()
    lib().islam_pvgo_default_params(ctypes.byref(p))
    for i in range(4):
        p.w[i] = float(loss_weight[i]) ** 2            # pvgo.py:125-129
    p.radius = float(radius)
    p.seg_len[0], p.seg_len[1] = int(seg_len[0]), int(seg_len[1])
    return p


def pvgo_workspace(N, device):
    nbytes = lib().islam_pvgo_workspace_bytes(N)
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def pvgo_reproj_struct(points, targets, K4, rgb2imu, weight, compat_first_motion=True):
    """islam_pvgo_reproj over float64 device tensors points (M,K,3), targets (M,K,2); keeps them alive on the struct."""
    require_cuda(points, targets)
    assert points.dtype == torch.float64 and targets.dtype == torch.float64 and points.is_contiguous() and targets.is_contiguous()
    assert points.shape[:2] == targets.shape[:2] and points.shape[2] == 3 and targets.shape[2] == 2
    r = _lib.PvgoReproj()
    r.points, r.targets, r.K = points.data_ptr(), targets.data_ptr(), int(points.shape[1])
    r.fx, r.fy, r.cx, r.cy = [float(v) for v in K4]
    for i, v in enumerate(rgb2imu):
        r.rgb2imu[i] = float(v)
    r.weight = float(weight)
    r.compat_first_motion = int(bool(compat_first_motion))
    r._keep = (points, targets)
    return r


def pvgo_reproj_reduce(nodes, reproj, dx=None):
    """Per-link keypoint reduction (M, 32): [J^T J upper (21) | J^T r (6) | r^T r | pad] at nodes or Exp(dx)*nodes."""
    require_cuda(nodes, dx)
    N = nodes.shape[0]
    red = torch.empty((N - 1, _lib.REPROJ_REC), dtype=torch.float64, device=nodes.device)
    check(lib().islam_pvgo_reproj_reduce(ptr(nodes), ptr(dx), N, ctypes.byref(reproj), ptr(red), stream_ptr(nodes.device)))
    return red


def pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, params, workspace=None, trace_cap=0, reproj=None):
    """In-place LM on float64 device tensors.  Returns (PvgoResult, trace ndarray (trials,3) or None)."""
    require_cuda(nodes, vels, poses, drots, dtrans, dvels, dts)
    for t in (nodes, vels, poses, drots, dtrans, dvels, dts):
        assert t.dtype == torch.float64 and t.is_contiguous()
    N = nodes.shape[0]
    if reproj is not None:
        assert reproj._keep[0].shape[0] == N - 1, 'reprojection factor needs one keypoint set per link'
    if workspace is None:
        workspace = pvgo_workspace(N, nodes.device)
    ws, nbytes = workspace
    res = _lib.PvgoResult()
    trace = np.zeros((trace_cap, 3), dtype=np.float64) if trace_cap > 0 else None
    tp = trace.ctypes.data_as(c_void_p) if trace is not None else c_void_p(0)
    check(lib().islam_pvgo_run_chain_reproj(ptr(nodes), ptr(vels), ptr(poses), ptr(drots), ptr(dtrans), ptr(dvels), ptr(dts), N,
                                            ctypes.byref(params), ctypes.byref(reproj) if reproj is not None else None,
                                            ptr(ws), c_size_t(nbytes), ctypes.byref(res), tp, trace_cap,
                                            stream_ptr(nodes.device)))
    if trace is not None:
        trace = trace[:min(res.trials, trace_cap)]
    return res, trace


def pvgo_linearize(nodes, vels, poses, drots, dtrans, dvels, dts):
    N = nodes.shape[0]
    M = N - 1
    lin = torch.empty((42, M), dtype=torch.float64, device=nodes.device)
    part = torch.empty(((M + 63) // 64,), dtype=torch.float64, device=nodes.device)
    check(lib().islam_pvgo_linearize(ptr(nodes), ptr(vels), ptr(poses), ptr(drots), ptr(dtrans), ptr(dvels), ptr(dts), N,
                                     ptr(lin), ptr(part), stream_ptr(nodes.device)))
    return lin, part


def pvgo_build_normal(lin, dts, N, w4, vmin=1e-4, vmax=1e32):
    dev = lin.device
    Hd = torch.zeros((N, 9, 9), dtype=torch.float64, device=dev)
    Ho = torch.zeros((N, 9, 9), dtype=torch.float64, device=dev)
    rhs = torch.zeros((N, 9), dtype=torch.float64, device=dev)
    w = (c_double * 4)(*[float(x) for x in w4])
    check(lib().islam_pvgo_build_normal(ptr(lin), ptr(dts), N, w, c_double(vmin), c_double(vmax), ptr(Hd), ptr(Ho),
                                        ptr(rhs), stream_ptr(dev)))
    return Hd, Ho, rhs


def pvgo_solve_chain(Hd, Ho, rhs, damping, seg_len=(0, 0), workspace=None):
    N = Hd.shape[0]
    if workspace is None:
        workspace = pvgo_workspace(N, Hd.device)
    ws, nbytes = workspace
    dx = torch.empty((N, 9), dtype=torch.float64, device=Hd.device)
    sl = (c_int * 2)(int(seg_len[0]), int(seg_len[1]))
    check(lib().islam_pvgo_solve_chain(ptr(Hd), ptr(Ho), ptr(rhs), c_double(damping), N, sl, ptr(ws), c_size_t(nbytes),
                                       ptr(dx), stream_ptr(Hd.device)))
    return dx


def pvgo_solve_chain_enqueue(Hd, Ho, rhs, workspace, seg_len=(0, 0), damping=0.0):
    """Stream-ordered islam_pvgo_solve_chain (the diagonal of Hd is damped in place, cumulatively): no read-back, no synchronisation; pvgo_solve_status() tells later
    whether any of the enqueued solves met a non-positive pivot."""
    N = Hd.shape[0]
    ws, nbytes = workspace
    dx = torch.empty((N, 9), dtype=torch.float64, device=Hd.device)
    sl = (c_int * 2)(int(seg_len[0]), int(seg_len[1]))
    check(lib().islam_pvgo_solve_chain_enqueue(ptr(Hd), ptr(Ho), ptr(rhs), c_double(damping), N, sl, ptr(ws), c_size_t(nbytes),
                                               ptr(dx), stream_ptr(Hd.device)))
    return dx


def pvgo_solve_status(N, workspace, device):
    """Synchronises; raises IslamHipError (ISLAM_ENOTPD) if a solve enqueued since the last call was not positive definite."""
    ws, nbytes = workspace
    check(lib().islam_pvgo_solve_status(N, ptr(ws), c_size_t(nbytes), stream_ptr(device)))


def pvgo_solve_chain_timed(Hd, Ho, rhs, damping, seg_len=(0, 0), workspace=None):
    """One solve with HIP events around each launch.  Returns (dx, {launch name: ms}, [(n, m, P) per level])."""
    N = Hd.shape[0]
    if workspace is None:
        workspace = pvgo_workspace(N, Hd.device)
    ws, nbytes = workspace
    dx = torch.empty((N, 9), dtype=torch.float64, device=Hd.device)
    sl = (c_int * 2)(int(seg_len[0]), int(seg_len[1]))
    ms = (c_float * 16)()
    plan = (c_int * (3 * _lib.MAX_LEVELS + 1))()
    nl = c_int(0)
    check(lib().islam_pvgo_solve_chain_timed(ptr(Hd), ptr(Ho), ptr(rhs), c_double(damping), N, sl, ptr(ws), c_size_t(nbytes),
                                             ptr(dx), ms, plan, ctypes.byref(nl), stream_ptr(Hd.device)))
    levels = [(plan[3 * l], plan[3 * l + 1], plan[3 * l + 2]) for l in range(_lib.MAX_LEVELS) if plan[3 * l] > 0]
    top = plan[3 * _lib.MAX_LEVELS]
    if nl.value == top + 1:         # root solve + whole down-sweep in one launch (bt_downsweep_kernel)
        names = ['eliminate_L%d' % l for l in range(top)] + ['root_L%d+downsweep' % top]
    else:
        names = ['eliminate_L%d' % l for l in range(top)] + ['top_L%d-%d' % (top, len(levels) - 1)] + \
            ['backsub_L%d' % l for l in range(top - 1, -1, -1)]
    return dx, dict(zip(names, [ms[i] for i in range(nl.value)])), levels


def pvgo_retract(nodes, vels, dx, sign=1.0):
    N = nodes.shape[0]
    no, vo = torch.empty_like(nodes), torch.empty_like(vels)
    check(lib().islam_pvgo_retract(ptr(nodes), ptr(vels), ptr(dx), c_double(sign), N, ptr(no), ptr(vo),
                                   stream_ptr(nodes.device)))
    return no, vo


def pvgo_align(nodes, vels, target7):
    N = nodes.shape[0]
    no, vo = torch.empty_like(nodes), torch.empty_like(vels)
    check(lib().islam_pvgo_align(ptr(nodes), ptr(vels), ptr(target7), N, ptr(no), ptr(vo), stream_ptr(nodes.device)))
    return no, vo


class _VoLoss(torch.autograd.Function):
    """graph.vo_loss(edges, poses) (pvgo.py:67-78): nodes detached, gradient to ``poses`` only, in PyPose's
    convention (left-tangent 6-vector stored in a 7-vector with a trailing 0)."""

    @staticmethod
    def forward(ctx, nodes, edges, poses):
        E = edges.shape[0]
        p64 = poses.detach().to(torch.float64).contiguous()
        err = torch.empty((E, 6), dtype=torch.float64, device=nodes.device)
        tl = torch.empty(E, dtype=torch.float64, device=nodes.device)
        rl = torch.empty(E, dtype=torch.float64, device=nodes.device)
        check(lib().islam_pvgo_vo_loss_fwd(ptr(nodes), ptr(edges), ptr(p64), E, ptr(err), ptr(tl), ptr(rl),
                                           stream_ptr(nodes.device)))
        ctx.save_for_backward(p64, err)
        ctx.out_dtype = poses.dtype
        return tl.to(poses.dtype), rl.to(poses.dtype)

    @staticmethod
    def backward(ctx, g_tl, g_rl):
        p64, err = ctx.saved_tensors
        E = p64.shape[0]
        grad = torch.empty((E, 7), dtype=torch.float64, device=p64.device)
        gt = g_tl.to(torch.float64).contiguous()
        gr = g_rl.to(torch.float64).contiguous()
        check(lib().islam_pvgo_vo_loss_bwd(ptr(p64), ptr(err), ptr(gt), ptr(gr), E, ptr(grad), stream_ptr(p64.device)))
        return None, None, grad.to(ctx.out_dtype)


def pvgo_vo_loss(nodes64, edges, poses):
    return _VoLoss.apply(nodes64, edges, poses)


class ClockProbe:
    """Shader clock the chip sustains while it is busy (islam_clock_probe): ``sample()`` queues a one-wavefront dependent-FMA kernel of
    ~40 us on a stream of this object's own -- it runs beside whatever the other streams are doing -- and ``mhz()`` reads the samples
    back (synchronises that stream only).  bench.py samples it before, during and after the stereo_vio measurement, so that a slow
    run names its cause (a box that clocks lower under load vs. anything else)."""

    def __init__(self, device, capacity=64, iters=20000):
        from ._lib import lib
        self.device, self.iters, self.n = torch.device(device), iters, 0
        self.buf = torch.zeros(capacity, 3, dtype=torch.int64, device=self.device)
        self.stream = torch.cuda.Stream(self.device)
        self.khz = lib().islam_wall_clock_khz(self.device.index or 0)

    def sample(self):
        from ._lib import check, lib, ptr, stream_ptr
        if self.n >= self.buf.shape[0]:
            return
        with torch.cuda.stream(self.stream):
            check(lib().islam_clock_probe(ptr(self.buf[self.n]), self.iters, stream_ptr(self.device)))
        self.n += 1

    def mhz(self):
        self.stream.synchronize()
        h = self.buf[:self.n].cpu().numpy().astype(float)
        return [float(c / w * self.khz * 1e-3) for w, c, _ in h if w > 0]
