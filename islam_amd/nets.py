"""Network definitions of the TartanVO front-end for PyTorch-ROCm (MIOpen / hipBLASLt convolutions).

These are re-implementations (table-driven builders) of the reference's module graph with IDENTICAL
state-dict keys and forward arithmetic, so the released checkpoints load unchanged
(SURVEY.md section 5 "Checkpoint / resume": 765 keys = flowNet 128 + stereoNet 517 + flowPoseNet 120):
  PWCDCNet     <- Network/PWC/PWCNet.py:58-292     (correlation / warp run on the HIP kernels)
  StereoNet7   <- Network/StereoNet7.py:13-146, Network/PSM/submodule.py:10-43,66-155,
                  Network/PSM/hourglass.py:6-77
  VOFlowRes    <- Network/VOFlowNet.py:7-39,42-194  (config=1, intrinsic, down_scale, stereo=0)
  VONet        <- Network/VONet.py:5-39
  IMUCorrector_CNN_GRU_WO_COV <- Network/IMUDenoiseNet.py:9-62
tests/test_nets_cpu.py checks key sets, shapes and forward outputs against golden vectors generated
by importing the reference modules (tests/golden/make_net_golden.py).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

# the two native ops of the flow network; tests may substitute CPU stand-ins for them
corr_fn = ops.FunctionCorrelation


def warp_fn(x, flow, scale):
    """PWCDCNet.warp on the HIP kernels, differentiable w.r.t. features and flow (islam_warp_mask / _bwd)."""
    return ops.warp(x, flow, scale)


# ------------------------------------------------------------------------------------------ PWC-DC-Net
def _conv_lrelu(cin, cout, stride=1, dilation=1):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, dilation, dilation, bias=True), nn.LeakyReLU(0.1))


class PWCDCNet(nn.Module):
    PYR = (3, 16, 32, 64, 96, 128, 196)             # channels of pyramid levels 0..6
    DENSE = (128, 128, 96, 64, 32)                  # DenseNet decoder widths
    WARP_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}

    def __init__(self, md=4, flow_norm=20.0, uncertainty=False):
        super().__init__()
        assert not uncertainty, 'VONet builds the flow net with uncertainty=False (Network/VONet.py:10)'
        self.flow_norm, self.uncertainty = flow_norm, uncertainty
        for l in range(1, 7):
            cin, cout = self.PYR[l - 1], self.PYR[l]
            first, second = ('a', 'aa') if l < 6 else ('aa', 'a')       # level 6 is declared aa -> a -> b
            setattr(self, 'conv%d%s' % (l, first), _conv_lrelu(cin, cout, stride=2))
            setattr(self, 'conv%d%s' % (l, second), _conv_lrelu(cout, cout))
            setattr(self, 'conv%db' % l, _conv_lrelu(cout, cout))
        self.leakyRELU = nn.LeakyReLU(0.1)
        nd = (2 * md + 1) ** 2
        for l in range(6, 1, -1):
            od = nd if l == 6 else nd + self.PYR[l] + 4
            c = od
            for i, w in enumerate(self.DENSE):
                setattr(self, 'conv%d_%d' % (l, i), _conv_lrelu(c, w))
                c += w
            setattr(self, 'predict_flow%d' % l, nn.Conv2d(c, 2, 3, 1, 1, bias=True))
            setattr(self, 'deconv%d' % l, nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True))
            if l > 2:
                setattr(self, 'upfeat%d' % l, nn.ConvTranspose2d(c, 2, 4, 2, 1, bias=True))
        c2 = nd + self.PYR[2] + 4 + sum(self.DENSE)
        for i, (cin, cout, dil) in enumerate([(c2, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16),
                                               (64, 32, 1)], 1):
            setattr(self, 'dc_conv%d' % i, _conv_lrelu(cin, cout, dilation=dil))
        self.dc_conv7 = nn.Conv2d(32, 2, 3, 1, 1, bias=True)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight.data, mode='fan_in')
                if m.bias is not None:
                    m.bias.data.zero_()

    def _pyramid(self, im):
        feats, x = [], im
        for l in range(1, 7):
            order = ('a', 'aa', 'b') if l < 6 else ('aa', 'a', 'b')
            for s in order:
                x = getattr(self, 'conv%d%s' % (l, s))(x)
            feats.append(x)
        return feats                                   # levels 1..6

    def _dense(self, l, x):
        for i in range(5):
            x = torch.cat((getattr(self, 'conv%d_%d' % (l, i))(x), x), 1)
        return x

    def _corr(self, a, b):
        return self.leakyRELU(corr_fn(a.float().contiguous(), b.float().contiguous()).to(a.dtype))

    def forward(self, x):
        p1, p2 = self._pyramid(x[:, 0:3]), self._pyramid(x[:, 3:6])
        x = self._dense(6, self._corr(p1[5], p2[5]))
        flows = {}
        for l in range(5, 1, -1):
            flows[l + 1] = getattr(self, 'predict_flow%d' % (l + 1))(x)
            up_flow = getattr(self, 'deconv%d' % (l + 1))(flows[l + 1])
            up_feat = getattr(self, 'upfeat%d' % (l + 1))(x)
            warped = warp_fn(p2[l - 1].float(), up_flow.float(), self.WARP_SCALE[l]).to(x.dtype)
            x = torch.cat((self._corr(p1[l - 1], warped), p1[l - 1], up_flow, up_feat), 1)
            x = self._dense(l, x)
        flow2 = self.predict_flow2(x)
        x = self.dc_conv4(self.dc_conv3(self.dc_conv2(self.dc_conv1(x))))
        flow2 = flow2 + self.dc_conv7(self.dc_conv6(self.dc_conv5(x)))
        return (flow2, flows[3], flows[4], flows[5], flows[6]), (None, None, None, None, None)

    # ---- inference path on the HIP matrix-core convolution (frozen flow net, BASELINE config 2 "bf16 nets") ----
    def _packed(self, name):
        """bf16 tap-major weights of a 3x3 conv for islam_conv3x3_mfma, re-packed when the fp32 master changes."""
        mod = getattr(self, name)
        conv = mod[0] if isinstance(mod, nn.Sequential) else mod
        cache = self.__dict__.setdefault('_mfma_cache', {})
        key = (conv.weight._version, conv.weight.data_ptr())
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = cache[name] = (key, ops.pack_conv3x3_weight(conv.weight), conv.bias.detach(), conv)
        return hit[1], hit[2], hit[3]

    def _c(self, name, x, out=None, coff=0, xoff=0):
        """One `conv()` block (PWCNet.py:16-20).  Layers with dilation <= 8 and >= 16 input channels run on the HIP kernel (1.3-4.5x
        MIOpen's fp32 Winograd on the stride-1 layers, scripts/conv_bench.py; the stride-2 heads of pyramid levels 3-6 for the launches
        they save: MIOpen's kernel + layout transposes + a bias and a LeakyReLU launch each, and because MIOpen's choice of kernel for
        them differs between machines); what is left for MIOpen: the dilation-16 layer when the sub-grid path does not apply."""
        packed, bias, conv = self._packed(name)
        act = isinstance(getattr(self, name), nn.Sequential)
        if (conv.stride[0] == 1 or (FLOW_S2_HIP and conv.stride[0] == 2 and conv.dilation[0] == 1)) and conv.dilation[0] <= 8 \
                and conv.in_channels >= 16:
            return ops.conv3x3_mfma(x, packed, bias, conv.out_channels, conv.stride[0], conv.dilation[0], 0.1 if act else 1.0, out, coff, xoff)
        xin = x if xoff == 0 else x[:, xoff:]
        y = conv(xin)
        y = F.leaky_relu(y, 0.1) if act else y
        if out is None:
            return y
        out[:, coff:coff + y.shape[1]].copy_(y)
        return out

    def _packed_flow(self, name, cin_eff):
        """bf16 tap-major weights of a 3x3 conv for islam_conv_nhwc_flow, with zero rows for the mirror's padding channels."""
        mod = getattr(self, name)
        conv = mod[0] if isinstance(mod, nn.Sequential) else mod
        cache = self.__dict__.setdefault('_flow_cache', {})
        key = (conv.weight._version, conv.weight.data_ptr(), cin_eff)
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            w = conv.weight.detach()
            if cin_eff > w.shape[1]:
                w = F.pad(w, (0, 0, 0, 0, 0, cin_eff - w.shape[1]))
            hit = cache[name] = (key, ops.pack_conv_nhwc_weight(w.to(torch.bfloat16)), conv.bias.detach().float().contiguous(), conv)
        return hit[1], hit[2], hit[3]

    def _dense_mfma(self, l, pieces):
        """The DenseNet block of one level without torch.cat: the buffer holds [conv4 | conv3 | conv2 | conv1 | conv0 |
        input] (PWCNet.py:237-292 prepend the newest features); every layer reads a suffix and writes the slice before it.
        FLOW_NHWC: the convolutions read a bf16 channels-last MIRROR of the buffer (islam_conv_nhwc_flow: 16-byte loads, 32-channel
        chunks) and write both the fp32 NCHW slice -- for the correlation / warp / transposed-convolution / flow-head consumers --
        and the mirror's slice for the next convolution.  The values a convolution multiplies are the same either way: the fp32
        path rounds its operands to bf16 when it stages them.  Returns (buffer, mirror or None)."""
        od = sum(p.shape[1] for p in pieces)
        B, _, H, W = pieces[0].shape
        tot = od + sum(self.DENSE)
        buf = torch.empty((B, tot, H, W), dtype=torch.float32, device=pieces[0].device)
        o = tot - od
        for p in pieces:
            buf[:, o:o + p.shape[1]].copy_(p)
            o += p.shape[1]
        return self._dense_run(l, buf, od)

    def _dense_run(self, l, buf, od, keep32=True):
        """The five convolutions of level l's DenseNet block on a buffer whose last ``od`` channels (the block's input) are filled.
        keep32=False: the convolutions write the bf16 mirror only (no fp32 NCHW copy of their outputs: nobody reads it when the
        flow head and the up-sampled features are taken from the mirror, _head_up_mirror)."""
        B, tot, H, W = buf.shape
        off = tot - od
        if FLOW_NHWC and off % 8 == 0 and all(w % 8 == 0 for w in self.DENSE):
            totp = off + (od + 7) // 8 * 8
            mir = torch.empty((B, totp, H, W), dtype=torch.bfloat16, device=buf.device, memory_format=torch.channels_last)
            ops.nchw_to_nhwc_mirror(buf, off, od, mir, off)            # the block's input (correlation, features, flow, up-features)
            for i, w in enumerate(self.DENSE):
                packed, bias, conv = self._packed_flow('conv%d_%d' % (l, i), totp - off)
                ops.conv_nhwc_flow(mir, off, totp - off, packed, bias, buf if keep32 else None, off - w, w, 0.1, ymir=mir, moff=off - w)
                off -= w
            return buf, mir
        for i, w in enumerate(self.DENSE):
            self._c('conv%d_%d' % (l, i), buf, out=buf, coff=off - w, xoff=off)
            off -= w
        return buf, None

    def _pyramid_level(self, l, f, pair=False):
        """conv{l}a (stride 2), conv{l}aa, conv{l}b (PWCNet.py:78-83, :240-243) on islam_flow_pyramid_level; weights re-packed when the
        fp32 masters change (ISLAM_FLOW_PYR=0: layer by layer)."""
        convs = [getattr(self, 'conv%d%s' % (l, s))[0] for s in ('a', 'aa', 'b')]
        cache = self.__dict__.setdefault('_pyr_cache', {})
        key = tuple((c.weight._version, c.weight.data_ptr(), c.bias._version) for c in convs)
        hit = cache.get(l)
        if hit is None or hit[0] != key:
            hit = cache[l] = (key, [ops.pack_pyramid_weight(c.weight) for c in convs], [c.bias.detach().float().contiguous() for c in convs])
        if pair:                                              # f = the (B, 6, H, W) frame pair: both frames as one batch, no concatenated copy
            return ops.flow_pyramid_level_pair(f, hit[1], hit[2], 0.1)
        return ops.flow_pyramid_level(f, hit[1], hit[2], 0.1)

    def _head_up(self, l, x, up, up_out=None, up_coff=0):
        """(predict_flow{l}(x), upfeat{l}(x) or None): the level's flow head and its up-sampled features (PWCNet.py:216-222) in one pass
        over the DenseNet buffer (islam_flow_head_up_f32, exact fp32).  ISLAM_FLOW_UP2=0: matrix-core head + MIOpen."""
        head = getattr(self, 'predict_flow%d' % l)
        dc = getattr(self, 'upfeat%d' % l) if up else None
        ok = (FLOW_UP2 and x.is_cuda and x.dtype == torch.float32 and head.weight.dtype == torch.float32 and head.kernel_size == (3, 3)
              and head.stride == (1, 1) and head.padding == (1, 1) and head.dilation == (1, 1) and head.out_channels == 2
              and (dc is None or (dc.out_channels == 2 and dc.kernel_size == (4, 4) and dc.stride == (2, 2) and dc.padding == (1, 1)
                                  and dc.output_padding == (0, 0) and dc.weight.dtype == torch.float32)))
        if not ok:
            uf = self._up2('upfeat%d' % l, x) if up else None
            if up_out is not None:
                up_out[:, up_coff:up_coff + 2].copy_(uf)
            return self._c('predict_flow%d' % l, x), uf
        cache = self.__dict__.setdefault('_head_cache', {})
        key = (head.weight._version, head.weight.data_ptr())
        hit = cache.get(l)
        if hit is None or hit[0] != key:
            hit = cache[l] = (key, head.weight.detach().permute(1, 0, 2, 3).contiguous())
        return ops.flow_head_up(x.contiguous(), hit[1], head.bias.detach(), dc.weight.detach() if up else None,
                                dc.bias.detach() if up and dc.bias is not None else None, up_out=up_out, up_coff=up_coff)

    def _head_up_mirror(self, l, mir, tot, up, up_out=None, up_coff=0):
        """predict_flow{l} and upfeat{l} (PWCNet.py:216-222) as ONE 3x3 convolution of the level's bf16 channels-last mirror on the
        matrix-core kernel: the transposed convolution's output pixel (2y+a, 2x+c) is a 2x2 sub-window of the 3x3 neighbourhood of input
        pixel (y, x) (K[a+r, c+s] = W[:, o, 3-2r-a, 3-2s-c], zero elsewhere), so its 2 channels x 4 parity classes are 8 output channels
        of a stride-1 3x3 convolution, the flow head's two are channels 8, 9 (padded to 16), and a pixel shuffle puts the 8 back on the
        2H x 2W grid.  bf16 operands (the mirror is what the DenseNet convolutions consumed; the head ran with bf16 operands before, the
        transposed convolution's weights are rounded here for the first time).  Returns (flow (B,2,H,W) fp32, up-sampled features or None)."""
        head = getattr(self, 'predict_flow%d' % l)
        dc = getattr(self, 'upfeat%d' % l) if up else None
        cache = self.__dict__.setdefault('_headm_cache', {})
        bkey = lambda m: None if m is None or m.bias is None else (m.bias._version, m.bias.data_ptr())
        key = (head.weight._version, head.weight.data_ptr(), bkey(head), None if dc is None else (dc.weight._version, dc.weight.data_ptr()),
               bkey(dc), mir.shape[1])
        hit = cache.get(l)
        if hit is None or hit[0] != key:
            C = head.weight.shape[1]
            w = torch.zeros((16, mir.shape[1], 3, 3), dtype=torch.float32, device=mir.device)     # (zero rows for the mirror's padding channels)
            b = torch.zeros(16, dtype=torch.float32, device=mir.device)
            w[8:10, :C] = head.weight.detach()
            b[8:10] = head.bias.detach()
            if dc is not None:
                wu = dc.weight.detach()                                                             # (C, 2, 4, 4)
                for o in range(2):
                    for a in range(2):
                        for c in range(2):
                            for r in range(2):
                                for q in range(2):
                                    w[o * 4 + a * 2 + c, :C, a + r, c + q] = wu[:, o, 3 - 2 * r - a, 3 - 2 * q - c]
                            if dc.bias is not None:
                                b[o * 4 + a * 2 + c] = dc.bias.detach()[o]
            hit = cache[l] = (key, ops.pack_conv_nhwc_weight(w.to(torch.bfloat16)), b)
        B, _, H, W = mir.shape
        y = torch.empty((B, 16, H, W), dtype=torch.float32, device=mir.device)
        ops.conv_nhwc_flow(mir, 0, mir.shape[1], hit[1], hit[2], y, 0, 16, 1.0)
        flow = y[:, 8:10].contiguous()
        upf = None
        if up:
            src = y[:, :8].view(B, 2, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3)                        # (B, o, H, a, W, c)
            if up_out is not None:
                up_out[:, up_coff:up_coff + 2].view(B, 2, H, 2, W, 2).copy_(src)
                upf = up_out
            else:
                upf = src.reshape(B, 2, 2 * H, 2 * W)
        return flow, upf

    def _up2(self, name, t):
        """The 4x4 stride-2 transposed convolutions with TWO output channels (deconv / upfeat): a memory-bound channel reduction on
        islam_deconv4x4s2_to2_f32 instead of MIOpen's backward-data kernels (ISLAM_FLOW_UP2=0: MIOpen)."""
        dc = getattr(self, name)
        if (FLOW_UP2 and t.is_cuda and t.dtype == torch.float32 and dc.out_channels == 2 and dc.kernel_size == (4, 4) and dc.stride == (2, 2)
                and dc.padding == (1, 1) and dc.output_padding == (0, 0) and dc.groups == 1 and dc.weight.dtype == torch.float32):
            return ops.deconv_to2(t.contiguous(), dc.weight.detach(), dc.bias.detach() if dc.bias is not None else None)
        return dc(t)

    def forward_mfma(self, x):
        """Same network as forward(), no autograd: both images go through the pyramid as one batch, convolutions through
        _c(), concatenations through channel slices.  fp32 activations; bf16-rounded operands inside the convolutions."""
        B = x.shape[0]
        x = x.float()
        pair = FLOW_PYR and x.is_cuda and x.shape[1] == 6 and x.is_contiguous()      # level 1 reads the two frames where they lie
        f, feats = (x if pair else torch.cat((x[:, 0:3], x[:, 3:6]), 0).contiguous()), []
        for l in range(1, 7):
            if FLOW_PYR and l <= 2 and f.is_cuda:                # levels 1, 2: the three layers in one launch, intermediates in LDS
                f = self._pyramid_level(l, f, pair=pair and l == 1)
            else:
                for s in (('a', 'aa', 'b') if l < 6 else ('aa', 'a', 'b')):
                    f = self._c('conv%d%s' % (l, s), f)
            feats.append(f)
        p1, p2 = [t[:B] for t in feats], [t[B:] for t in feats]
        lrelu = lambda t: F.leaky_relu(t, 0.1)
        direct = FLOW_UP2 and x.is_cuda                     # producers write straight into the level's concatenation buffer
        nd = sum(self.DENSE)
        if direct:
            a6 = p1[5].contiguous()
            buf = torch.empty((a6.shape[0], nd + 81, a6.shape[2], a6.shape[3]), dtype=torch.float32, device=x.device)
            ops.corr81_act(a6, p2[5].contiguous(), buf, nd, 0.1)
            x, mir = self._dense_run(6, buf, 81, keep32=not FLOW_HEAD_MIRROR)
        else:
            x, mir = self._dense_mfma(6, [lrelu(corr_fn(p1[5].contiguous(), p2[5].contiguous()))])
        flows = {}
        for l in range(5, 1, -1):
            a, b2 = p1[l - 1].contiguous(), p2[l - 1].contiguous()
            if direct:
                # [conv4 .. conv0 | corr 81 | features ca | up_flow 2 | up_feat 2] (PWCNet.py:227 torch.cat order)
                ca = a.shape[1]
                buf = torch.empty((a.shape[0], nd + 81 + ca + 4, a.shape[2], a.shape[3]), dtype=torch.float32, device=x.device)
                if FLOW_HEAD_MIRROR and mir is not None:
                    flows[l + 1], _ = self._head_up_mirror(l + 1, mir, x.shape[1], True, up_out=buf, up_coff=nd + 81 + ca + 2)
                else:
                    flows[l + 1], _ = self._head_up(l + 1, x, True, up_out=buf, up_coff=nd + 81 + ca + 2)
                up_flow = self._up2('deconv%d' % (l + 1), flows[l + 1])
                warped = warp_fn(b2, up_flow, self.WARP_SCALE[l])
                ops.corr81_act(a, warped, buf, nd, 0.1)
                buf[:, nd + 81:nd + 81 + ca].copy_(a)
                buf[:, nd + 81 + ca:nd + 81 + ca + 2].copy_(up_flow)
                x, mir = self._dense_run(l, buf, 81 + ca + 4, keep32=not FLOW_HEAD_MIRROR)
                continue
            flows[l + 1], up_feat = self._head_up(l + 1, x, True)
            up_flow = self._up2('deconv%d' % (l + 1), flows[l + 1])
            warped = warp_fn(b2, up_flow.contiguous(), self.WARP_SCALE[l])
            x, mir = self._dense_mfma(l, [lrelu(corr_fn(a, warped)), a, up_flow, up_feat])
        flow2 = self._head_up_mirror(2, mir, x.shape[1], False)[0] if (direct and FLOW_HEAD_MIRROR and mir is not None) else self._head_up(2, x, False)[0]
        Hc, Wc = x.shape[2], x.shape[3]
        if mir is not None and all(Hc % getattr(self, 'dc_conv%d' % i)[0].dilation[0] == 0 and Wc % getattr(self, 'dc_conv%d' % i)[0].dilation[0] == 0
                                   for i in range(1, 7)):
            # the context network on the channels-last kernel, bf16 between its layers (what the next layer's operands are rounded
            # to anyway); a dilated layer runs as d*d dense convolutions on the sub-grids of the map (islam_conv_nhwc_flow)
            cur, cin = mir, mir.shape[1]
            for i in range(1, 7):
                packed, bias, conv = self._packed_flow('dc_conv%d' % i, cin)
                co = conv.out_channels
                if i < 6:
                    out = torch.empty((x.shape[0], co, Hc, Wc), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
                    ops.conv_nhwc_flow(cur, 0, cin, packed, bias, None, 0, co, 0.1, ymir=out, moff=0, dilation=conv.dilation[0])
                else:
                    out = torch.empty((x.shape[0], co, Hc, Wc), dtype=torch.float32, device=x.device)
                    ops.conv_nhwc_flow(cur, 0, cin, packed, bias, out, 0, co, 0.1, dilation=conv.dilation[0])
                cur, cin = out, co
            return (flow2 + self._c('dc_conv7', cur), flows[3], flows[4], flows[5], flows[6]), (None, None, None, None, None)
        if mir is not None:                               # dc_conv1 (565 -> 128, the largest convolution of the net) reads the mirror
            packed, bias, conv = self._packed_flow('dc_conv1', mir.shape[1])
            d1 = torch.empty((x.shape[0], conv.out_channels, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
            d1 = ops.conv_nhwc_flow(mir, 0, mir.shape[1], packed, bias, d1, 0, conv.out_channels, 0.1)
        else:
            d1 = self._c('dc_conv1', x)
        x = self._c('dc_conv4', self._c('dc_conv3', self._c('dc_conv2', d1)))
        flow2 = flow2 + self._c('dc_conv7', self._c('dc_conv6', self._c('dc_conv5', x)))
        return (flow2, flows[3], flows[4], flows[5], flows[6]), (None, None, None, None, None)


# ------------------------------------------------------------------------------------------ StereoNet7
def _convbn(cin, cout, k, stride, pad, dilation):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, dilation if dilation > 1 else pad, dilation, bias=False),
                         nn.BatchNorm2d(cout))


# Which convolutions of the frozen stereo net's bf16 channels-last execution copy run on islam_conv_nhwc_bf16 (hand-written
# implicit GEMM, round-to-nearest-even, BatchNorm statistics from the epilogue, the producer's BatchNorm + ReLU applied on load,
# bias / residual / ReLU fused) instead of MIOpen.  MIOpen's kernels for the 32- and 64-channel 3x3 shapes TRUNCATE the fp32
# accumulator to bf16 (scripts/calib/bf16_rounding_probe.py); the hand-written kernel rounds to nearest even, delivers the
# BatchNorm statistics from its epilogue and is faster than MIOpen's pick on every 3x3 shape of the net but one (64->128: 69 vs
# 65 us; 128->128: 108 vs 123 us, 352->128: 1.00 vs 1.20 ms), so every stride-1 3x3 convolution runs on it (HIP_CONV_MAX_C).
# ISLAM_HIP_CONV: 0 = all MIOpen (round-1 path), 1 = 3x3 up to HIP_CONV_MAX_C channels, 2 (default) = also the 1x1 convolutions of
# the hourglass modules with their fused epilogue.  MIOpen's 1x1 kernels are ~25 % faster (12.40 vs 12.48 ms per forward), but some
# of them truncate as well: against the reference-generated vectors level 1 leaves rms 3.6e-2 / bias 2.6e-2, level 2 rms 1.2e-2 /
# bias 1.6e-3 (profiles/r02/golden_errors.txt) -- parity decides.
import os as _os

HIP_CONV_LEVEL = int(_os.environ.get('ISLAM_HIP_CONV', '2'))
# the flow net's DenseNet convolutions on the channels-last kernel through a bf16 mirror of the concatenation buffer (0: fp32 NCHW kernel)
FLOW_NHWC = _os.environ.get('ISLAM_FLOW_NHWC', '1') == '1'
HIP_CONV_MAX_C = int(_os.environ.get('ISLAM_HIP_CONV_MAX_C', '512'))
FLOW_UP2 = _os.environ.get('ISLAM_FLOW_UP2', '1') == '1'
FLOW_PYR = _os.environ.get('ISLAM_FLOW_PYR', '1') == '1'
# stride-2 layers of pyramid levels 3-6 on islam_conv3x3_mfma (bias + LeakyReLU in the launch) instead of MIOpen + bias + activation launches
FLOW_S2_HIP = _os.environ.get('ISLAM_FLOW_S2_HIP', '1') == '1'
# flow head + up-sampled features of a level as one 3x3 convolution of the bf16 mirror; the DenseNet convolutions then skip their fp32 copies
FLOW_HEAD_MIRROR = _os.environ.get('ISLAM_FLOW_HEAD_MIRROR', '1') == '1'
# capture the frozen flow and stereo nets as two parallel branches of the HIP graph (0: one after the other): forward-only +11 %,
# sequential bilevel step +6 %, pipelined step unchanged (scripts/vio_only.py, five runs each)
FROZEN_FORK = _os.environ.get('ISLAM_FROZEN_FORK', '1') == '1'
# the stereo decoder's 4x4 stride-2 transposed convolutions on the channels-last kernel (0: MIOpen + torch.cat, for A/B runs)
HIP_DECONV = _os.environ.get('ISLAM_HIP_DECONV', '1') == '1'
# hourglass Residual modules as one launch each (0: three convolution launches, for A/B runs); only for maps of at most
# ISLAM_HG_FUSED_MAX_PIXELS pixels per image
# 1: stride-1 convbn layers through islam_conv_nhwc_bf16_bn -- convolution + [scale | shift] of its train-mode BatchNorm in one C call, same
# bits as convolution(stats) + partial_fold + bn_finalize.  Behind the persistent kernels (20 of the stereo net's 29 such layers at the
# benched size: one row of partial sums per workgroup) the finalize launch reads the rows directly, no fold launch; layers on the tile
# kernel keep fold + finalize (ISLAM_BN_FINALIZE=1: the round-4 variant, fold + finalize in one ticketed launch).  Default off: neither
# moves the replay -- round 4: 6.80 / 6.80 ms per stereo replay with the ticketed launch, 6.76 / 6.75 without; round 6, 20 launches fewer
# per forward: 8.02 / 7.99 ms per replay of both nets against 7.96 / 7.98 (profiles/r06/bn_finalize_ab_r06.txt).  Launches of a few
# microseconds on one branch of the replay are filled by the other branch's kernels; what the replay costs is its large kernels.
BN_FOLD_FINALIZE = os.environ.get('ISLAM_BN_FOLD_FINALIZE', '0') == '1'
# 1x1 stride-1 convbn layers (layer3's 64 -> 128 downsample, the four SPP branch convolutions) on the channels-last kernel with the batch
# statistics from its epilogue instead of CK + a statistics pass over the output (0: as through round 5)
CONV1X1_BN = os.environ.get('ISLAM_CONV1X1_BN', '1') == '1'
# the stereo pair from its two fp32 images to the execution copy's bf16 inputs in one kernel (0: torch.cat + cast + layout copy + stacking)
STEREO_PAIR_PREPARE = os.environ.get('ISLAM_STEREO_PAIR_PREPARE', '1') == '1'
# the stereo decoder's torch.cat((previous stage, skip tensor), 1) in front of every transposed convolution (StereoNet7.py:121-138) kept as a
# PAIR of dense tensors that the consumer reads where they lie (islam_deconv4x4s2_nhwc_bf16_cat / islam_conv_nhwc_bf16_s2_cat) instead of a
# concatenation buffer the skip tensor is copied into.  Bit-identical, 0.3 GB less traffic per forward at B = 8 (five copies, 79 us when run
# alone) -- and no faster where it counts: replay 7.872 / 7.876 ms with the pairs against 7.873 / 7.907 without (alternating runs): the
# copies are HBM-bound and ride beside the other branch's matrix-core kernels.  Default off.
CAT_PAIRS = os.environ.get('ISLAM_CAT_PAIRS', '0') == '1'


class _Cat:
    """torch.cat((a, b), 1) of two dense channels-last bf16 tensors that has not been written: a consumer that can reads the two halves
    (ops.deconv_nhwc / ops.conv_nhwc_s2 with x2=), any other calls materialize()."""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def materialize(self):
        return torch.cat((self.a, self.b), 1)

    def readable(self, cin, mult):
        a, b = self.a, self.b
        return (a.shape[1] + b.shape[1] == cin and a.shape[1] % mult == 0 and ops.fusable_nhwc_bf16(a, a.shape[1])
                and ops.fusable_nhwc_bf16(b, b.shape[1]) and a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:])

# stride-2 convolutions of the frozen stereo net (layer2's first block, the quarter-resolution tail) on islam_conv_nhwc_bf16_s2; 0: MIOpen / CK
HIP_CONV_S2 = os.environ.get('ISLAM_HIP_CONV_S2', '1') != '0'
HIP_FIRST_LAYER = os.environ.get('ISLAM_HIP_FIRST_LAYER', '1') != '0'   # the 3 -> 32 stride-2 first layer of the stereo net on the channels-last kernel
UPSAMPLE_CAT = os.environ.get('ISLAM_UPSAMPLE_CAT', '1') != '0'     # the feature extractor's six up-samplings + concatenation as one launch
# the feature extractor's last convolution writes conv_c0's input in place (left images first in the batch); 0: dense features + two copies
STEREO_DIRECT_CAT = os.environ.get('ISLAM_STEREO_DIRECT_CAT', '1') != '0'
HG_FUSED = _os.environ.get('ISLAM_HG_FUSED', '1') == '1'
HG_FUSED_MAX_PIXELS = int(_os.environ.get('ISLAM_HG_FUSED_MAX_PIXELS', str(1 << 30)))


def execution_description():
    """What the frozen nets' execution copies run on, built from the switches above (bench.py prints it with the stereo_vio figure:
    a hand-written literal there went stale when the stride-2 convolutions moved onto the HIP kernel)."""
    on = lambda flag, a, b: a if flag else b
    stereo = ['stereo net: bf16 NHWC execution copy']
    if HIP_CONV_LEVEL >= 1:
        stereo.append('stride-1 3x3%s convolutions on the HIP implicit-GEMM kernel conv_nhwc_kernel (BatchNorm statistics in the epilogue, '
                      'BatchNorm + ReLU on load)' % on(HIP_CONV_LEVEL >= 2, on(CONV1X1_BN, ', the hourglass 1x1 and the 1x1 convbn', ' and the hourglass 1x1'), ''))
        from ._lib import lib
        if lib().islam_conv_ws_mode(-1) > 0:
            stereo.append('the twelve 3x3 layers with 128 output channels of layer3 / layer4 on the weight-stationary persistent kernel conv3x3_ws_kernel (weights in the register file), '
                          'the eight 32->32 ones on the persistent kernel conv3x3_ws32_kernel'
                          + (' (layers whose image is whole tiles and has >= 1024 of them, as at the benched size; the tile kernel otherwise)'
                             if lib().islam_conv_ws_mode(-1) == 1 else ' (every layer whose image is whole tiles)'))
    else:
        stereo.append('all convolutions on MIOpen')
    stereo.append(on(HIP_CONV_S2, 'stride-2 convolutions on islam_conv_nhwc_bf16_s2' + on(HIP_FIRST_LAYER, ' including the 3->32 first layer', ' (first layer on MIOpen)'),
                     'stride-2 convolutions on MIOpen / CK'))
    stereo.append(on(HG_FUSED, 'the 35 hourglass Residual modules as one launch each (islam_hg_residual_nhwc_bf16)', 'hourglass Residual modules as three launches'))
    stereo.append(on(HIP_DECONV, 'decoder transposed convolutions on the convolution kernel', 'decoder transposed convolutions on MIOpen'))
    stereo.append(on(UPSAMPLE_CAT, 'SPP up-samplings + concatenation as one launch', 'SPP up-samplings as separate launches'))
    stereo.append(on(CONV1X1_BN and HIP_CONV_LEVEL >= 2, 'biased convolutions of the 384 / 512-channel levels, the second SPP and the two output convolutions (conv_c12, the one-channel conv_c13) on MIOpen / CK',
                     'biased 1x1 / SPP 1x1 convolutions on MIOpen / CK'))
    flow = ['flow net: ' + on(FLOW_NHWC, 'DenseNet blocks / context network on the channels-last kernel through a bf16 mirror', 'DenseNet blocks on islam_conv3x3_mfma (fp32 NCHW)')]
    flow.append(on(FLOW_PYR, 'pyramid levels 1-2 as one fused three-layer launch each', 'pyramid levels 1-2 layer by layer'))
    flow.append(on(FLOW_S2_HIP, 'stride-2 layers of levels 3-6 on islam_conv3x3_mfma', 'stride-2 layers of levels 3-6 on MIOpen'))
    flow.append(on(FLOW_HEAD_MIRROR, 'flow head + up-sampled features of a level as one convolution of the mirror', 'flow heads on islam_conv3x3_mfma'))
    flow.append('81-channel correlation at four pixels per lane, warp + mask kernel')
    return '; '.join(stereo) + ' | ' + '; '.join(flow) + on(FROZEN_FORK, ' | the two nets as parallel branches of one HIP graph', ' | the two nets one after the other')


def _hip_conv_ok(conv, x, fused_1x1=False, strided=False):
    """strided: the caller can also run Conv2d(stride = 2) on the channels-last kernel (islam_conv_nhwc_bf16_s2)."""
    if HIP_CONV_LEVEL < 1 or not isinstance(conv, nn.Conv2d) or not isinstance(x, torch.Tensor):
        return False
    k = conv.kernel_size[0]
    s2 = conv.stride == (2, 2)
    if conv.kernel_size != (k, k) or k not in (1, 3) or not (conv.stride == (1, 1) or (s2 and strided and HIP_CONV_S2)) \
            or conv.padding != (k // 2, k // 2) or conv.dilation != (1, 1) or conv.groups != 1 or conv.in_channels % 8 or conv.out_channels % 8:
        return False
    if not ops.fusable_nhwc_bf16(x, conv.in_channels) or conv.weight.dtype != torch.bfloat16:
        return False
    if k == 1:
        return HIP_CONV_LEVEL >= 2 and (fused_1x1 or s2)
    return max(conv.in_channels, conv.out_channels) <= HIP_CONV_MAX_C


def _packed_nhwc(conv, cin=None):
    """bf16 tap-major weights of a convolution for islam_conv_nhwc_bf16, re-packed when the execution copy is rebuilt.
    ``cin`` > conv.in_channels: zero weights for the extra (zero-filled) input channels of a padded buffer."""
    key = (conv.weight._version, conv.weight.data_ptr(), cin)
    hit = conv.__dict__.get('_nhwc_packed')
    if hit is None or hit[0] != key:
        w = conv.weight
        if cin is not None and cin > w.shape[1]:
            w = F.pad(w, (0, 0, 0, 0, 0, cin - w.shape[1]))
        hit = conv.__dict__['_nhwc_packed'] = (key, ops.pack_conv_nhwc_weight(w))
    return hit[1]


class _Pending:
    """A convbn output whose BatchNorm (+ ReLU) has not been applied yet: the raw convolution output and the [scale | shift] of
    its batch statistics.  The consumer applies it while staging its input tile (islam_conv_nhwc_bf16 `in_affine`), or
    materialize() does (one in-place pass)."""

    def __init__(self, raw, affine):
        self.raw, self.affine = raw, affine

    def materialize(self):
        return ops.bn_apply_(self.raw, self.affine, relu=True)


def _cbn(convbn, x, relu=False, res=None, defer=False):
    """A `convbn` block (Conv2d, BatchNorm2d; submodule.py:10-13) together with the residual add and / or ReLU that follows
    it.  On the frozen bf16 channels-last execution copy in training mode: the convolution on islam_conv_nhwc_bf16 with the batch
    statistics from its epilogue where _hip_conv_ok says so (``x`` may be a _Pending producer; ``defer`` returns one), MIOpen +
    ONE fused BatchNorm op (ops.bn_train_) otherwise; everywhere else plain torch."""
    conv, bn = convbn[0], convbn[1]
    plain_bn = type(bn) is nn.BatchNorm2d and bn.training and bn.weight.dtype == torch.float32
    xin = x.raw if isinstance(x, _Pending) else x
    if plain_bn and bn.num_features <= 256 and _hip_conv_ok(conv, xin, fused_1x1=CONV1X1_BN, strided=True):
        if conv.stride == (2, 2):     # layer2's first block (submodule.py:76-85): stride-2 3x3 convbn and its stride-2 1x1 downsample
            y, folded = ops.conv_nhwc_s2(xin, _packed_nhwc(conv), conv.out_channels, conv.kernel_size[0],
                                         in_affine=x.affine if isinstance(x, _Pending) else None, stats=True)
            affine = ops.bn_finalize(folded, bn, y.shape[0] * y.shape[2] * y.shape[3])
        elif BN_FOLD_FINALIZE:          # statistics folded and finalized by ONE launch behind the convolution (islam_conv_nhwc_bf16_bn)
            y, affine = ops.conv_nhwc_bn(xin, _packed_nhwc(conv), conv.out_channels, conv.kernel_size[0], bn,
                                         in_affine=x.affine if isinstance(x, _Pending) else None)
        else:
            y, folded = ops.conv_nhwc(xin, _packed_nhwc(conv), conv.out_channels, conv.kernel_size[0],
                                      in_affine=x.affine if isinstance(x, _Pending) else None, stats=True)
            affine = ops.bn_finalize(folded, bn, y.shape[0] * y.shape[2] * y.shape[3])
        if defer and relu and res is None:
            return _Pending(y, affine)
        if res is not None and not res.is_contiguous(memory_format=torch.channels_last):
            res = res.contiguous(memory_format=torch.channels_last)
        return ops.bn_apply_(y, affine, relu, res)
    if isinstance(x, _Pending):
        x = x.materialize()
    y = conv(x)
    c = y.shape[1]
    # the fused op normalises with the statistics of THIS rank's tensor: only a plain nn.BatchNorm2d may take it (a
    # dist_train.ShardedBatchNorm2d must run its own forward, which all-reduces [sum | sum of squares | count] first)
    if plain_bn and ops.fusable_nhwc_bf16(y, c) and c <= 256 and 256 % (c // 8) == 0:
        if res is not None and not res.is_contiguous(memory_format=torch.channels_last):
            res = res.contiguous(memory_format=torch.channels_last)
        return ops.bn_train_(y, bn, relu, res)
    y = bn(y)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


class _PSMBlock(nn.Module):
    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        self.conv1 = nn.Sequential(_convbn(cin, cout, 3, stride, 1, 1), nn.ReLU(inplace=True))
        self.conv2 = _convbn(cout, cout, 3, 1, 1, 1)
        self.downsample = downsample

    def forward(self, x):
        y = _cbn(self.conv1[0], x, relu=True, defer=True)      # its BatchNorm + ReLU may ride on conv2's input staging
        res = x if self.downsample is None else _cbn(self.downsample, x)
        return _cbn(self.conv2, y, res=res)


def _block_mean(x, k):
    """AvgPool2d((k, k), stride=(k, k)) (floor mode, no padding) as a reshape + mean: one pass over x."""
    B, C, H, W = x.shape
    h, w = H // k, W // k
    if ops.fusable_nhwc_bf16(x, C) and h >= 1 and w >= 1:
        return ops.avgpool(x, k)
    return x[:, :, :h * k, :w * k].reshape(B, C, h, k, w, k).mean((3, 5))


def _spp_pools(x):
    """The four non-overlapping average pools of the SPP branches (kernels 8/16/32/64, submodule.py:103-122): the 8x8
    pool reads x once, every coarser pool is the 2x2 mean of the previous one (identical blocks: floor mode drops the same
    rows / columns either way).  torch's avg_pool2d kernel takes ~0.4 ms per call on these shapes."""
    p8 = _block_mean(x, 8)
    p16 = _block_mean(p8, 2)
    p32 = _block_mean(p16, 2)
    return {8: p8, 16: p16, 32: p32, 64: _block_mean(p32, 2)}


class feature_extraction(nn.Module):
    def __init__(self, last_planes=32, bigger=False, middleblock=16):
        super().__init__()
        self.bigger = bigger
        relu = lambda: nn.ReLU(inplace=True)
        self.firstconv = nn.Sequential(_convbn(3, 32, 3, 2, 1, 1), relu(), _convbn(32, 32, 3, 1, 1, 1), relu(),
                                       _convbn(32, 32, 3, 1, 1, 1), relu())
        self._c = 32
        self.layer1 = self._layer(32, 3, 1)
        self.layer2 = self._layer(64, middleblock, 2)
        self.layer3 = self._layer(128, 3, 1)
        self.layer4 = self._layer(128, 3, 1)
        for i, k in enumerate((64, 32, 16, 8), 1):
            setattr(self, 'branch%d' % i, nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)), _convbn(128, 32, 1, 1, 0, 1), relu()))
        self.lastconv = nn.Sequential(_convbn(320 + (32 if bigger else 0), 128, 3, 1, 1, 1), relu(),
                                      nn.Conv2d(128, last_planes, 1, 1, 0, bias=False))

    def _layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self._c != planes:
            down = nn.Sequential(nn.Conv2d(self._c, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        mods = [_PSMBlock(self._c, planes, stride, down)]
        self._c = planes
        mods += [_PSMBlock(planes, planes, 1, None) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, x, into=None):
        """into: callable (images, channels, h, w, dtype, device) -> None | (buffer, [(first image, images, channel offset), ...]):
        when it returns a buffer, the result is written into those channel slices of it (batch ranges of the result, in order)
        and the buffer is returned instead of the feature tensor -- StereoNet7 assembles conv_c0's input this way."""
        f = x
        c0, bn0 = self.firstconv[0][0], self.firstconv[0][1]
        if (x.shape[1] == 8 and c0.in_channels < 8 and HIP_CONV_S2 and HIP_CONV_LEVEL >= 1 and ops.fusable_nhwc_bf16(x, 8)
                and c0.weight.dtype == torch.bfloat16 and c0.kernel_size == (3, 3) and c0.stride == (2, 2) and c0.padding == (1, 1)
                and c0.bias is None and type(bn0) is nn.BatchNorm2d and bn0.training and bn0.weight.dtype == torch.float32):
            # x = ops.stack_pair_pad8 of the stereo pair (StereoNet7.forward): the 3 -> 32 stride-2 first layer on the channels-last
            # kernel with its weights zero-padded to 8 input channels, statistics from the epilogue, BatchNorm + ReLU applied by the second
            # convolution's load (MIOpen: a layout pass, the convolution, a statistics pass and an apply pass)
            key = (c0.weight._version, c0.weight.data_ptr())
            hit = self.__dict__.get('_first8')
            if hit is None or hit[0] != key:
                hit = self.__dict__['_first8'] = (key, ops.pack_conv_nhwc_weight(F.pad(c0.weight.detach(), (0, 0, 0, 0, 0, 8 - c0.in_channels))))
            y, folded = ops.conv_nhwc_s2(x, hit[1], c0.out_channels, 3, stats=True)
            f = _Pending(y, ops.bn_finalize(folded, bn0, y.shape[0] * y.shape[2] * y.shape[3]))
        else:
            if x.shape[1] == 8 and c0.in_channels < 8:
                f = x[:, :c0.in_channels].contiguous(memory_format=torch.channels_last)
            f = _cbn(self.firstconv[0], f, relu=True)
        for i in (2, 4):                                     # firstconv = (convbn, ReLU) x 3; the middle one's BatchNorm + ReLU
            f = _cbn(self.firstconv[i], f, relu=True, defer=(i == 2))      # is applied by the third convolution's load
        o0 = self.layer1(f)
        raw = self.layer2(o0)
        skip = self.layer4(self.layer3(raw))
        hw = [skip.shape[2], skip.shape[3]]
        pools = _spp_pools(skip)
        br = [ops.resize_bilinear(_cbn(getattr(self, 'branch%d' % i)[1], pools[k], relu=True), hw, align_corners=True)
              for i, k in ((4, 8), (3, 16), (2, 32), (1, 64))]
        pieces = [raw, skip] + br
        if self.bigger and ops.fusable_nhwc_bf16(skip, skip.shape[1]) and all(ops.fusable_nhwc_bf16(t, t.shape[1]) for t in pieces + [o0]):
            # bilinear resizing is per channel: up(cat(pieces)) = cat(up(piece)) -- every piece is up-sampled straight into its
            # channel slice of the 352-channel input of lastconv (no torch.cat of the pieces, no copy of the up-sampled 320
            # channels into the second torch.cat: ~2 GB of traffic per forward at B=8)
            if UPSAMPLE_CAT and all(tuple(t.shape[2:]) == tuple(hw) for t in pieces) and tuple(o0.shape[2:]) == (hw[0] * 2, hw[1] * 2) \
                    and o0.is_contiguous(memory_format=torch.channels_last):
                # ... and in ONE launch that writes whole 352-channel pixels, layer1's 32 channels included (six launches writing
                # 64 .. 256-byte pieces 704 bytes apart + a copy: 0.40 ms; ISLAM_UPSAMPLE_CAT=0)
                feat = ops.upsample_cat(pieces, (hw[0] * 2, hw[1] * 2), tail=o0, align_corners=True)
            else:
                ctot = sum(t.shape[1] for t in pieces) + o0.shape[1]
                feat = torch.empty((skip.shape[0], ctot, hw[0] * 2, hw[1] * 2), dtype=skip.dtype, device=skip.device,
                                   memory_format=torch.channels_last)
                off = 0
                for t in pieces:
                    ops.resize_bilinear_into(t, feat, off, align_corners=True)
                    off += t.shape[1]
                feat[:, off:].copy_(o0)
        else:
            feat = torch.cat(pieces, 1)
            if self.bigger:
                feat = torch.cat((ops.resize_bilinear(feat, [hw[0] * 2, hw[1] * 2], align_corners=True), o0), 1)
        last = self.lastconv[2]                               # Conv2d(128, 32, 1, bias=False) behind (convbn, ReLU)
        y = _cbn(self.lastconv[0], feat, relu=True, defer=True)
        if isinstance(y, _Pending) and last.bias is None and _hip_conv_ok(last, y.raw, fused_1x1=True):
            # BatchNorm + ReLU of the 352 -> 128 convolution applied while the 1x1 convolution stages its input: no apply pass over the
            # 128-channel half-resolution tensor (94 us) and no MIOpen / CK launch (94 us) -- one launch of the channels-last kernel
            dest = into(y.raw.shape[0], last.out_channels, y.raw.shape[2], y.raw.shape[3], y.raw.dtype, y.raw.device) if into else None
            if dest is not None:
                buf, parts = dest
                for b0, nb, off in parts:                     # (a batch range of a channels-last tensor is itself dense channels-last)
                    ops.conv_nhwc_into(y.raw[b0:b0 + nb], _packed_nhwc(last), last.out_channels, 1, buf, off, in_affine=y.affine)
                return buf
            return ops.conv_nhwc(y.raw, _packed_nhwc(last), last.out_channels, 1, in_affine=y.affine)
        return last(y.materialize() if isinstance(y, _Pending) else y)


class _HGConv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.inp_dim = cin
        self.conv = nn.Conv2d(cin, cout, k, 1, (k - 1) // 2, bias=True)

    def forward(self, x):
        assert x.shape[1] == self.inp_dim, '{} {}'.format(x.shape[1], self.inp_dim)
        return self.conv(x)

    def run(self, x, relu=False, res=None, in_relu=False):
        """act(conv([relu] x) + bias [+ res]): on islam_conv_nhwc_bf16 with the bias add, residual add and ReLU in its epilogue, or --
        for the shapes MIOpen is faster on -- a bias-free MIOpen convolution and ONE in-place pass over its output (frozen
        bf16 channels-last execution copy only; see _HGResidual.forward)."""
        c = self.conv
        b = self.__dict__.get('_b32')
        key = (c.bias._version, c.bias.data_ptr())
        if b is None or b[0] != key:
            b = self.__dict__['_b32'] = (key, c.bias.detach().float().contiguous())
        if _hip_conv_ok(c, x, fused_1x1=True):
            if res is not None and not res.is_contiguous(memory_format=torch.channels_last):
                res = res.contiguous(memory_format=torch.channels_last)
            return ops.conv_nhwc(x, _packed_nhwc(c), c.out_channels, c.kernel_size[0], bias=b[1], res=res, relu=relu, in_relu=in_relu)
        if in_relu:
            x = F.relu(x)
        y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
        return ops.bias_act_add_(y, b[1], res, relu)


class _HGResidual(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.relu = nn.ReLU()
        h = int(cout / 2)
        self.conv1, self.conv2, self.conv3 = _HGConv(cin, h, 1), _HGConv(h, h, 3), _HGConv(h, cout, 1)
        self.skip_layer = _HGConv(cin, cout, 1)
        self.need_skip = cin != cout

    def _fused_pack(self):
        """packed weights + [b1 | b2 | b3] for islam_hg_residual_nhwc_bf16, rebuilt when the execution copy's parameters change"""
        cs = (self.conv1.conv, self.conv2.conv, self.conv3.conv)
        key = tuple((c.weight._version, c.weight.data_ptr(), c.bias._version, c.bias.data_ptr()) for c in cs)
        hit = self.__dict__.get('_fused')
        if hit is None or hit[0] != key:
            hit = self.__dict__['_fused'] = (key, ops.pack_hg_residual(*(c.weight for c in cs)),
                                             torch.cat([c.bias.detach().float() for c in cs]).contiguous())
        return hit[1], hit[2]

    def forward(self, x):
        c1, c3 = self.conv1.conv, self.conv3.conv
        if (HG_FUSED and ops.fusable_nhwc_bf16(x, c1.in_channels) and c1.weight.dtype == torch.bfloat16 and c1.in_channels % 64 == 0
                and c1.in_channels <= 256 and c3.out_channels % 64 == 0 and c3.out_channels <= 256 and c1.out_channels * 2 == c3.out_channels
                and (c3.out_channels > 64 or c1.in_channels == 64)
                and x.shape[2] * x.shape[3] <= HG_FUSED_MAX_PIXELS):
            # the whole module in ONE launch, intermediates in LDS (islam_hg_residual_nhwc_bf16); the 1x1 skip convolution of the two
            # modules that change the channel count stays a launch of its own and comes in as the residual
            res = self.skip_layer.run(x) if self.need_skip else None
            packed, b = self._fused_pack()
            return ops.hg_residual(x, packed, b, c3.out_channels, res)
        if ops.fusable_nhwc_bf16(x, self.conv1.conv.out_channels) and self.conv3.conv.out_channels % 8 == 0:
            res = self.skip_layer.run(x) if self.need_skip else x
            y = self.conv2.run(self.conv1.run(x, relu=True, in_relu=True), relu=True)      # relu(x) rides on conv1's input staging
            return self.conv3.run(y, relu=False, res=res)
        res = self.skip_layer(x) if self.need_skip else x
        y = self.conv3(self.relu(self.conv2(self.relu(self.conv1(self.relu(x))))))
        return y + res


class Hourglass(nn.Module):
    def __init__(self, n, f, increase=0):
        super().__init__()
        nf = f + increase
        self.up1 = _HGResidual(f, nf)
        self.pool1 = nn.MaxPool2d(2, 2)
        self.n = n
        self.low2 = Hourglass(n - 1, nf, 0) if n > 1 else _HGResidual(nf, nf)
        self.low3 = _HGResidual(nf, nf)
        self.up2 = nn.Upsample(scale_factor=2, mode='bilinear')

    def forward(self, x, cat_with=None, pair=False):
        """cat_with: a tensor the caller concatenates behind the result (torch.cat((hourglass(x), cat_with), 1)); on the bf16
        channels-last path the result is written straight into its half of that concatenation, which is returned -- or, with ``pair``,
        the concatenation is returned as a _Cat of the dense result and cat_with (no buffer, no copy of cat_with)."""
        u = self.up1(x)
        fused = ops.fusable_nhwc_bf16(u, u.shape[1])
        low = self.low3(self.low2(ops.maxpool2(u) if fused else self.pool1(u)))
        # self.up2 = nn.Upsample(scale_factor=2, mode='bilinear') (align_corners=False); HIP kernels for bf16 channels-last:
        # pooling, and the up-sampling with the addition of u in the same pass
        if fused and ops.fusable_nhwc_bf16(low, low.shape[1]) and (low.shape[2] * 2, low.shape[3] * 2) == tuple(u.shape[2:]):
            if cat_with is not None and ops.fusable_nhwc_bf16(cat_with, cat_with.shape[1]) and cat_with.shape[2:] == u.shape[2:]:
                if pair:
                    return _Cat(ops.resize_bilinear_add(low, u), cat_with)
                c = u.shape[1]
                out = torch.empty((u.shape[0], c + cat_with.shape[1], u.shape[2], u.shape[3]), dtype=u.dtype, device=u.device,
                                  memory_format=torch.channels_last)
                out[:, c:].copy_(cat_with)
                return ops.resize_bilinear_add(low, u, out=out, coff=0)
            y = ops.resize_bilinear_add(low, u)
        else:
            y = u + ops.resize_bilinear(low, (low.shape[2] * 2, low.shape[3] * 2), align_corners=False)
        return y if cat_with is None else torch.cat((y, cat_with), 1)


class SSP(nn.Module):
    def __init__(self, c):
        super().__init__()
        for i, k in enumerate((64, 32, 16, 8), 1):
            setattr(self, 'branch%d' % i, nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)), nn.Conv2d(c, int(c / 4), 1, 1, 0),
                                                        nn.ReLU(inplace=True)))

    def forward(self, x):
        hw = [x.shape[2], x.shape[3]]
        pools = _spp_pools(x)
        return torch.cat([x] + [ops.resize_bilinear(getattr(self, 'branch%d' % i)[1:](pools[k]), hw)
                                for i, k in ((4, 8), (3, 16), (2, 32), (1, 64))], 1)


class StereoNet7(nn.Module):
    def __init__(self, version=0, uncertainty=False, act_fun='relu'):
        super().__init__()
        self.version = version
        self.feature_extraction = feature_extraction(last_planes=64, bigger=True, middleblock=3)
        self.actfun = F.selu if act_fun == 'selu' else F.relu
        self.conv_c0 = nn.Conv2d(134, 64, 3, padding=1)
        self.conv_c1 = Hourglass(2, 64, 0)
        self.conv_c2 = Hourglass(2, 64, 0)
        self.conv_c2_SSP = SSP(64)
        self.conv_c3 = Hourglass(2, 128, 64)
        self.conv_c4 = Hourglass(2, 192, 64)
        self.conv_c5 = nn.Conv2d(256, 384, 3, padding=1)
        self.conv_c6 = nn.Conv2d(384, 512, 3, padding=1)
        self.deconv_c7 = nn.ConvTranspose2d(896, 320, 4, 2, 1)
        self.deconv_c8 = nn.ConvTranspose2d(576, 192, 4, 2, 1)
        self.conv_c8 = Hourglass(2, 192, 0)
        self.deconv_c9 = nn.ConvTranspose2d(384, 128, 4, 2, 1)
        self.conv_c9 = Hourglass(2, 128, 0)
        self.deconv_c10 = nn.ConvTranspose2d(256, 64, 4, 2, 1)
        self.conv_c10 = Hourglass(2, 64, 0)
        self.deconv_c11 = nn.ConvTranspose2d(128, 64, 4, 2, 1)
        self.conv_c12 = nn.Conv2d(64, 16, 1, padding=0)
        self.conv_c13 = nn.Conv2d(16, 1, 1, padding=0)
        self.conv_c6_2 = nn.Conv2d(512, 512, 3, padding=1)
        self.deconv_c7_2 = nn.ConvTranspose2d(512, 512, 4, 2, 1)

    def forward(self, x, quarter=False):
        """quarter=True returns only the pixels VONet keeps (Network/VONet.py:33-34 `F.interpolate(disp, scale_factor=0.25,
        mode='nearest')` = disp[..., ::4, ::4]): conv_c12 / conv_c13 are 1x1 and an output pixel (4y, 4x) of the 4x4 stride-2
        deconv_c11 depends on a 2x2 input patch only, so the full-resolution tail (a 294 MB tensor at B=8, 16x the needed work)
        collapses to a 2x2 stride-2 convolution at half resolution -- the same arithmetic for the pixels that are used."""
        assert x.shape[1] % 2 == 0
        B, C2, H, W = x.shape
        c0 = self.conv_c0
        cin = (c0.in_channels + 7) // 8 * 8
        direct = {}

        def into(n, cf, h, w, dtype, device):
            # conv_c0's input is torch.cat((left features, right features, half-resolution image)) (StereoNet7.py:103-105).  With the
            # left images first and the right images behind them in the batch, the feature extractor's last convolution writes the two
            # halves of its result straight into their channel slices of that (padded) buffer: no dense feature tensor, no copies of it
            if not (STEREO_DIRECT_CAT and HIP_CONV_LEVEL >= 1 and n == 2 * B and dtype == torch.bfloat16 and c0.weight.dtype == torch.bfloat16
                    and cf % 8 == 0 and 2 * cf + C2 == c0.in_channels):
                return None
            direct['buf'] = torch.empty((B, cin, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)
            return direct['buf'], [(0, B, 0), (B, B, cf)]
        stacked = STEREO_DIRECT_CAT and x.dtype == torch.bfloat16
        if stacked and HIP_FIRST_LAYER and HIP_CONV_S2 and C2 // 2 < 8 and x.is_contiguous(memory_format=torch.channels_last):
            xs = getattr(x, '_islam_stacked', None)          # (VONet._run_frozen_stereo_pair: built together with x from the fp32 pair)
            if xs is None or tuple(xs.shape) != (2 * B, 8, H, W):
                xs = ops.stack_pair_pad8(x)                  # [left images; right images], three channels padded to eight
        else:
            xs = torch.cat((x[:, :C2 // 2], x[:, C2 // 2:]), 0) if stacked else x.reshape(B * 2, C2 // 2, H, W)
        f2 = self.feature_extraction(xs, into=into if stacked else None)      # left / right images stacked along the batch
        half = None
        act = self.actfun
        pool = ops.maxpool2                                                   # F.max_pool2d(t, kernel_size=2); HIP kernel for bf16 channels-last
        relu_pool = (lambda t: ops.maxpool2(t, relu=True)) if act is F.relu else (lambda t: pool(act(t)))      # pool(relu(t)) in one pass
        if 'buf' in direct:
            buf = direct['buf']
            cf = (c0.in_channels - C2) // 2
        else:
            if stacked:                                      # (the direct path did not apply: back to the reference's interleaved order)
                f2 = torch.stack((f2[:B], f2[B:]), 1).reshape(B * 2, f2.shape[1], f2.shape[2], f2.shape[3])
            cf = f2.shape[1]
            buf = None
        if buf is not None or (HIP_CONV_LEVEL >= 1 and ops.fusable_nhwc_bf16(f2, cf) and c0.weight.dtype == torch.bfloat16 and cf % 8 == 0):
            # 134 = 128 + 6 input channels: the concatenation is built with 136 channels (two zero ones) so that conv_c0 runs on
            # islam_conv_nhwc_bf16 -- MIOpen's bf16 kernel for this shape truncates its output (scripts/calib/bf16_rounding_probe.py)
            # and this is the first layer of the un-normalised hourglass path, where that loss of magnitude is never renormalised
            if buf is None:
                buf = torch.empty((B, cin, f2.shape[2], f2.shape[3]), dtype=f2.dtype, device=f2.device, memory_format=torch.channels_last)
                buf[:, :cf].copy_(f2[0::2])                  # f.reshape(B, 2*cf, h, w) of the reference: [left | right] features per
                buf[:, cf:2 * cf].copy_(f2[1::2])            # image, written straight into the padded buffer (one copy, not two)
            if (STEREO_DIRECT_CAT and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and C2 <= 8 and C2 % 2 == 0
                    and H % 2 == 0 and W % 2 == 0 and (2 * cf) % 8 == 0 and cin == 2 * cf + 8 and buf.shape[2:] == (H // 2, W // 2)):
                ops.half_image_into(x, buf, 2 * cf)          # the half-resolution pair + the zero channels, one launch
            else:
                buf[:, 2 * cf:c0.in_channels].copy_(F.interpolate(x, scale_factor=0.5, mode='bilinear'))
                buf[:, c0.in_channels:].zero_()
            bkey = (c0.bias._version, c0.bias.data_ptr())
            b32 = self.__dict__.get('_c0_b32')
            if b32 is None or b32[0] != bkey:
                b32 = self.__dict__['_c0_b32'] = (bkey, c0.bias.detach().float().contiguous())
            x0 = ops.conv_nhwc(buf, _packed_nhwc(c0, cin), c0.out_channels, 3, bias=b32[1])
        else:
            f = f2.reshape(B, cf * 2, f2.shape[2], f2.shape[3])
            x0 = c0(torch.cat((f, F.interpolate(x, scale_factor=0.5, mode='bilinear')), 1))
        cat0 = self.conv_c1(x0)                                               # 1/2, 64
        cat1 = self.conv_c2_SSP(pool(self.conv_c2(cat0)))                     # 1/4, 128
        cat2 = pool(self.conv_c3(cat1))                                       # 1/8, 192
        cat3 = pool(self.conv_c4(cat2))                                       # 1/16, 256
        cat4 = relu_pool(self.conv_c5(cat3))                                  # 1/32, 384
        x = act(self.conv_c6_2(relu_pool(self.conv_c6(cat4))))                # 1/64, 512
        x = self._deconv_act(self.deconv_c7_2, x, cat4)
        x = self._deconv_act(self.deconv_c7, x, cat3)
        x = self.conv_c8(self._deconv_act(self.deconv_c8, x), cat_with=cat2, pair=CAT_PAIRS)     # = torch.cat((conv_c8(...), cat2), 1), StereoNet7.py:129-138
        x = self.conv_c9(self._deconv_act(self.deconv_c9, x), cat_with=cat1, pair=CAT_PAIRS)
        x = self.conv_c10(self._deconv_act(self.deconv_c10, x), cat_with=cat0, pair=CAT_PAIRS)
        if quarter:
            x = self._deconv_c11_quarter(x, act)
        else:
            x = act(self.deconv_c11(x.materialize() if isinstance(x, _Cat) else x))
        return self.conv_c13(act(self.conv_c12(x))), None

    def _deconv_act(self, dc, x, skip=None):
        """act(dc(x)), concatenated with ``skip`` when given (StereoNet7.py:121-136).  On the frozen bf16 channels-last execution
        copy the 4x4 stride-2 transposed convolution runs on islam_deconv4x4s2_nhwc_bf16 -- four 2x2 convolutions on the matrix
        cores, bias + ReLU in the epilogue, written straight into its channel slice of the concatenation (MIOpen ran these as
        bf16 backward-data kernels whose output was then activated and copied by torch.cat)."""
        act = self.actfun
        co, ci = dc.out_channels, dc.in_channels
        x2 = None
        if isinstance(x, _Cat):                              # the previous stage's pair: read where its halves lie, if this layer can
            if HIP_DECONV and act is F.relu and x.readable(ci, 32):
                x, x2 = x.a, x.b
            else:
                x = x.materialize()
        if (HIP_DECONV and act is F.relu and (x2 is not None or ops.fusable_nhwc_bf16(x, ci)) and dc.weight.dtype == torch.bfloat16 and co % 8 == 0
                and dc.kernel_size == (4, 4) and dc.stride == (2, 2) and dc.padding == (1, 1) and dc.output_padding == (0, 0)
                and dc.groups == 1 and dc.dilation == (1, 1)
                and (skip is None or (ops.fusable_nhwc_bf16(skip, skip.shape[1]) and tuple(skip.shape[2:]) == (2 * x.shape[2], 2 * x.shape[3])))):
            key = (dc.weight._version, dc.weight.data_ptr(), dc.bias._version, dc.bias.data_ptr())
            hit = dc.__dict__.get('_nhwc_deconv')
            if hit is None or hit[0] != key:
                hit = dc.__dict__['_nhwc_deconv'] = (key, ops.pack_deconv_nhwc_weight(dc.weight), dc.bias.detach().float().contiguous())
            out = None
            if skip is not None and CAT_PAIRS:
                return _Cat(ops.deconv_nhwc(x, hit[1], hit[2], co, relu=True, x2=x2), skip)
            if skip is not None:
                out = torch.empty((x.shape[0], co + skip.shape[1], 2 * x.shape[2], 2 * x.shape[3]), dtype=x.dtype, device=x.device,
                                  memory_format=torch.channels_last)
                out[:, co:].copy_(skip)
            return ops.deconv_nhwc(x, hit[1], hit[2], co, out=out, yoff=0, relu=True, x2=x2)
        if x2 is not None:
            x = torch.cat((x, x2), 1)
        y = act(dc(x))
        return y if skip is None else torch.cat((y, skip), 1)

    def _deconv_c11_quarter(self, x, act=None):
        """act(deconv_c11(x)[..., ::4, ::4]).  ConvTranspose2d(k=4, s=2, p=1): out[o, oy, ox] = b[o] + sum_i sum_ky,kx x[i, (oy+1-ky)/2,
        (ox+1-kx)/2] W[i, o, ky, kx] over the taps where the division is exact; for oy = 4y: ky = 1 -> row 2y, ky = 3 -> row 2y-1.
        That is a 2x2 convolution with stride 2 and one row / column of zero padding in front: K[o, i, a, b] = W[i, o, 3-2a, 3-2b]."""
        dc = self.deconv_c11
        x2 = None
        if isinstance(x, _Cat):
            if HIP_CONV_S2 and HIP_CONV_LEVEL >= 1 and act is F.relu and dc.bias is not None and x.readable(dc.in_channels, 16):
                x, x2 = x.a, x.b
            else:
                x = x.materialize()
        key = (dc.weight._version, dc.weight.data_ptr())
        hit = self.__dict__.get('_c11q')
        if hit is None or hit[0] != key:
            K = dc.weight.detach()[:, :, [3, 1]][:, :, :, [3, 1]].permute(1, 0, 2, 3).contiguous()
            if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
                K = K.contiguous(memory_format=torch.channels_last)
            hit = self.__dict__['_c11q'] = (key, K)
        h, w = x.shape[2] // 2, x.shape[3] // 2
        co, ci = hit[1].shape[0], hit[1].shape[1]
        if (HIP_CONV_S2 and HIP_CONV_LEVEL >= 1 and act is F.relu and (x2 is not None or ops.fusable_nhwc_bf16(x, ci)) and hit[1].dtype == torch.bfloat16
                and ci % 8 == 0 and co % 8 == 0 and dc.bias is not None):
            # the 2x2 stride-2 convolution, its bias, the ReLU and the crop to (h, w) in one launch of the channels-last kernel
            pk = self.__dict__.get('_c11q_packed')
            if pk is None or pk[0] != key:
                pk = self.__dict__['_c11q_packed'] = (key, ops.pack_conv_nhwc_weight(hit[1]), dc.bias.detach().float().contiguous())
            return ops.conv_nhwc_s2(x, pk[1], co, 2, bias=pk[2], relu=True, out_hw=(h, w), x2=x2)
        if x2 is not None:
            x = torch.cat((x, x2), 1)
        y = F.conv2d(x, hit[1], dc.bias, stride=2, padding=1)[:, :, :h, :w]
        return y if act is None else act(y)


# ------------------------------------------------------------------------------------------ VOFlowRes
def _conv_relu(cin, cout, stride):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, 1), nn.ReLU(inplace=True))


def _linear_relu(cin, cout):
    return nn.Sequential(nn.Linear(cin, cout), nn.ReLU(inplace=True))


def _conv_nobias(conv, x):
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def _fused_tail_ok(x):
    """The one-launch elementwise tail (ops.bias_act) serves fp32 channels-last activations on the device."""
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) \
        and not torch.is_autocast_enabled()


def _conv_relu_fused(seq, x):
    """`_conv_relu` (conv + bias + ReLU) with the bias add and the ReLU in ONE launch forward and ONE backward (ops.bias_act)."""
    from . import ops
    conv = seq[0]
    y = _conv_nobias(conv, x)
    if not _fused_tail_ok(y) or conv.out_channels % 4:
        return F.relu(y + conv.bias.view(1, -1, 1, 1))
    return ops.bias_act(y, conv.bias, None, True)


class _PoseBlock(nn.Module):
    fused_tail = False           # set by VOFlowRes.set_fused_tail: same parameters, same state dict, fewer launches

    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        self.conv1 = _conv_relu(cin, cout, stride)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, 1)
        self.downsample = downsample

    def forward(self, x):
        if self.fused_tail and _fused_tail_ok(x):
            from . import ops
            h = _conv_relu_fused(self.conv1, x)
            y = _conv_nobias(self.conv2, h)
            sc = x if self.downsample is None else self.downsample(x)
            if _fused_tail_ok(y) and _fused_tail_ok(sc):
                return ops.bias_act(y, self.conv2.bias, sc, True)           # relu((conv2 + bias) + shortcut): one launch
            return F.relu(y + self.conv2.bias.view(1, -1, 1, 1) + sc)
        y = self.conv2(self.conv1(x))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)), inplace=True)


class VOFlowRes(nn.Module):
    def __init__(self, intrinsic=True, down_scale=True, config=1, stereo=0, fix_parts=(), **_unused):
        super().__init__()
        assert intrinsic and down_scale and config == 1 and stereo == 0, \
            'only the configuration VONet instantiates is built (Network/VONet.py:16)'
        mods = [_conv_relu(4, 32, 2), _conv_relu(32, 32, 1), _conv_relu(32, 32, 1)]
        c = 32
        for planes, blocks in ((64, 3), (128, 4), (128, 6), (256, 7), (256, 3)):
            layer = [_PoseBlock(c, planes, 2, nn.Conv2d(c, planes, 1, 2))]
            layer += [_PoseBlock(planes, planes, 1, None) for _ in range(1, blocks)]
            mods.append(nn.Sequential(*layer))
            c = planes
        self.feat_net = nn.Sequential(*mods)
        dim = 256 * 6
        self.voflow_trans = nn.Sequential(_linear_relu(dim, 128), _linear_relu(128, 32), nn.Linear(32, 3))
        self.voflow_rot = nn.Sequential(_linear_relu(dim, 128), _linear_relu(128, 32), nn.Linear(32, 3))
        for name, part in (('feat', self.feat_net), ('rot', self.voflow_rot), ('trans', self.voflow_trans)):
            if name in fix_parts:
                for p in part.parameters():
                    p.requires_grad = False

    def set_fused_tail(self, on=True):
        """Bias add + ReLU (+ shortcut add) behind every convolution of the encoder as one HIP launch forward and one backward
        (islam_amd/csrc/pose_ops.hip) instead of 2-4 ATen launches each way; fp32 channels-last activations only (other inputs
        take the plain modules).  Same parameters and state dict."""
        self.fused_tail = bool(on)
        for m in self.modules():
            if isinstance(m, _PoseBlock):
                m.fused_tail = bool(on)

    def set_hip_head(self, on=True, graphs=False):
        """Forward and backward of the WHOLE module on the hand-written fp32 kernels of csrc/pose_head.hip (islam_amd/pose_head.py): one C
        call each way instead of ~110 / ~430 third-party launches (CK / MIOpen convolutions, ATen elementwise, zero-fills).  Serves fp32
        channels-last (B,4,H,W) device inputs that carry no gradient; anything else takes the modules below.  graphs: the two calls
        replay from captured HIP graphs.  Same parameters and state dict; the backward node adds the parameter gradients to .grad itself
        and hangs on ``self.hip_head().leaf`` (see pose_head.py)."""
        self.use_hip_head, self.hip_graphs = bool(on), bool(graphs)
        self.__dict__.pop('_hip_head', None)

    def hip_head(self):
        hh = self.__dict__.get('_hip_head')
        if hh is None:
            from . import pose_head
            hh = self.__dict__['_hip_head'] = pose_head.PoseHeadHip(self, graphs=getattr(self, 'hip_graphs', False))
        return hh

    def forward(self, x, extrinsic=None):
        if getattr(self, 'use_hip_head', False) and not (torch.is_grad_enabled() and x.requires_grad):
            from . import pose_head
            if pose_head.supported_input(x):
                return self.hip_head()(x)
        if getattr(self, 'fused_tail', False) and _fused_tail_ok(x):
            for i, m in enumerate(self.feat_net):
                x = _conv_relu_fused(m, x) if i < 3 else m(x)
        else:
            x = self.feat_net(x)
        x = x.reshape(x.shape[0], -1)          # (= .view for the reference's NCHW activations; channels_last ones are re-laid out)
        return torch.cat((self.voflow_trans(x), self.voflow_rot(x)), 1)


# ------------------------------------------------------------------------------------------ VONet
class VONet(nn.Module):
    def __init__(self, fix_parts=('flow', 'stereo')):
        super().__init__()
        self.flowNet = PWCDCNet(uncertainty=False)
        self.stereoNet = StereoNet7()
        self.flowPoseNet = VOFlowRes(intrinsic=True, down_scale=True, stereo=0, fix_parts=fix_parts)
        for name, net in (('flow', self.flowNet), ('stereo', self.stereoNet)):
            if name in fix_parts:
                for p in net.parameters():
                    p.requires_grad = False
        # BASELINE config 2 ("bf16 nets"): the frozen nets run through a reduced-precision channels_last EXECUTION COPY
        # (_HalfExec) whose conv weights are cast once; autocast re-casts ~150 weight tensors and as many activations per
        # forward and keeps interpolate / cat in fp32.  The fp32 master modules keep the checkpoint (765 keys, fp32).
        self.frozen_dtype = None        # stereo net (77 % of the FLOPs): bf16 execution copy on MIOpen
        self.flow_dtype = None          # flow net: not None -> PWCDCNet.forward_mfma (HIP implicit-GEMM convolutions)
        self._exec = {}
        self._graphs = {}
        self.graph_frozen = False
        self.pose_channels_last = False
        self.graph_pose = False             # set before the first training forward; fixed input shape from then on
        self._pose_graphed = None
        # BASELINE config 2 ("bf16 nets / fp64 LM") for the TRAINABLE pose head as well: fp32 master weights and optimizer
        # state, forward + backward under bf16 autocast (standard mixed precision; the reference ran TF32)
        self.pose_dtype = None

    def set_pose_channels_last(self, on=True):
        """Run the trainable pose head on channels_last (NHWC) fp32 tensors: MIOpen's fp32 implicit-GEMM kernels are NHWC
        and otherwise wrap every convolution (forward, data- and weight-gradient) in layout-transposing launches
        (~270 per step).  Same parameters, same state dict, same arithmetic up to MIOpen's kernel choice."""
        self.pose_channels_last = bool(on)
        self._drop_pose_graph()                      # a captured pose-head graph holds the old layout
        self.flowPoseNet.to(memory_format=torch.channels_last if on else torch.contiguous_format)
        self.flowPoseNet.set_fused_tail(on and os.environ.get('ISLAM_POSE_NO_FUSED_TAIL') != '1')     # (channels-last activations: ops.bias_act applies)

    def set_frozen_dtype(self, dtype, flow_dtype=None):
        self.frozen_dtype, self.flow_dtype = dtype, flow_dtype
        self._exec = {}
        self._graphs = {}
        self.__dict__.pop('_frozen_param_list', None)

    def _run_frozen(self, name, master, dtype, x, quarter=False):
        if dtype is None or any(p.requires_grad for p in master.parameters()):
            return master(x, quarter=True) if quarter else master(x)      # trainable parts keep their fp32 autograd path
        if name == 'flow':                   # fp32 activations, bf16 operands inside the HIP matrix-core convolutions
            with torch.no_grad():
                return master.forward_mfma(x)
        ex = self._exec.get(name)
        if ex is None or ex.dtype != dtype:
            ex = self._exec[name] = _HalfExec(master, dtype)
        xin = x.to(dtype).contiguous(memory_format=torch.channels_last)
        return ex.module()(xin, quarter=True) if quarter else ex.module()(xin)

    def _run_frozen_stereo_pair(self, left, right, quarter=True):
        """_run_frozen('stereo', ..., torch.cat((left, right), 1)) with the concatenation, the bf16 cast, the channels-last copy and the
        stacking of the pair done by ONE kernel (ops.stereo_pair_prepare) when the execution copy takes the stacked batch anyway."""
        dtype, master = self.frozen_dtype, self.stereoNet
        if (STEREO_PAIR_PREPARE and dtype == torch.bfloat16 and not any(p.requires_grad for p in master.parameters()) and left.is_cuda
                and left.dtype == torch.float32 and right.dtype == torch.float32 and left.shape == right.shape and left.shape[1] <= 3
                and left.is_contiguous() and right.is_contiguous() and STEREO_DIRECT_CAT and HIP_FIRST_LAYER and HIP_CONV_S2):
            ex = self._exec.get('stereo')
            if ex is None or ex.dtype != dtype:
                ex = self._exec['stereo'] = _HalfExec(master, dtype)
            x6, xs = ops.stereo_pair_prepare(left, right)
            x6._islam_stacked = xs
            return ex.module()(x6, quarter=True) if quarter else ex.module()(x6)
        return self._run_frozen('stereo', master, dtype, torch.cat((left, right), 1), quarter=quarter)

    def set_graph_frozen(self, on=True):
        """Replay the frozen flow + disparity forward (~750 launches) from a captured HIP graph instead of enqueueing it
        launch by launch: same kernels and results, the host is free during the replay.  Graphs are keyed by input shape and
        train / eval mode and dropped by reset_graphs() (call it after changing frozen weights: the graph holds the bf16
        execution copies and packed weights it was captured with)."""
        self.graph_frozen = bool(on)
        self.reset_graphs()

    def reset_graphs(self):
        if getattr(self, '_graphs', None):
            torch.cuda.synchronize()            # no replay may still be in flight when the graphs and their memory pool go
        self._graphs = {}
        self.__dict__.pop('_frozen_param_list', None)
        self._drop_pose_graph()

    def frozen_nets_trainable(self):
        """True if any parameter of the flow / stereo nets requires grad.  Asked once or twice per batch on the host's critical path
        (TartanVO.prefetch, frozen_forward): the flat parameter list is cached -- walking two module trees of ~900 modules through
        nn.Module.parameters() cost ~1 ms per call -- and dropped by reset_graphs() / set_frozen_dtype()."""
        ps = self.__dict__.get('_frozen_param_list')
        if ps is None:
            ps = self.__dict__['_frozen_param_list'] = [p for n in (self.flowNet, self.stereoNet) for p in n.parameters()]
        return any(p.requires_grad for p in ps)

    def set_mode(self, is_train):
        """vonet.train() / vonet.eval() (TartanVO.py:91) without walking ~900 modules when nothing changes: nn.Module.train()
        re-assigns `training` on every module of the tree (~1.5 ms of host time per call, twice per batch, in front of everything
        the batch enqueues).  The walk is skipped only if the net and its three sub-nets already are in the mode this method
        applied last; anything else -- a first call, a mode change, a sub-net switched by hand -- takes the full walk."""
        is_train = bool(is_train)
        if (self.__dict__.get('_mode_applied') is is_train and self.training == is_train and self.flowNet.training == is_train
                and self.stereoNet.training == is_train and self.flowPoseNet.training == is_train):
            return
        self.train(is_train)
        self.__dict__['_mode_applied'] = is_train

    def _frozen_graphed(self, imgs):
        key = (tuple(imgs[0].shape), imgs[0].device, self.stereoNet.training, self.flowNet.training, self.frozen_dtype, self.flow_dtype)
        ring = self._graphs.get(key)
        if ring is None:
            ring = self._graphs[key] = {'inst': [], 'next': 0}
        # graph_instances (default 1) captured copies of the same forward, used round-robin: with two, the replay of batch k+2 can be
        # QUEUED while batch k+1's is still running (TartanVO.prefetch two batches ahead) -- the replays then run back to back on the
        # side stream instead of waiting for the host's turn-around (fence + input copies + graph launch: ~0.85 ms per batch).  Every
        # instance has its own static inputs / outputs, memory pool and fork stream; BatchNorm running statistics are shared module
        # buffers and are updated in stream order.
        want = max(1, int(getattr(self, 'graph_instances', 1)))
        if len(ring['inst']) < want:
            static_in = [t.detach().clone() for t in imgs]
            if not ring['inst']:
                with torch.no_grad():
                    for _ in range(2):                  # MIOpen's kernel search, lazy initialisation: outside the capture
                        self._frozen_eager(*static_in)
            torch.cuda.synchronize(imgs[0].device)
            g = torch.cuda.CUDAGraph()
            fork = torch.cuda.Stream(imgs[0].device) if FROZEN_FORK else None        # (created outside the capture)
            with torch.no_grad(), torch.cuda.graph(g):
                static_out = self._frozen_eager(*static_in, fork=fork)
            ring['inst'].append({'g': g, 'in': static_in, 'out': static_out, 'done': None, 'fork': fork})
            st = ring['inst'][-1]
        else:
            st = ring['inst'][ring['next'] % len(ring['inst'])]
            ring['next'] += 1
        g, static_in, static_out = st['g'], st['in'], st['out']
        # One launch of a captured graph in flight at a time: the host waits for THIS INSTANCE's previous replay (and the clones of its
        # outputs) before it queues the next one.  With two batches ahead on ONE instance the second replay used to be queued on the side
        # stream while the first was still running, and on this stack (ROCm 7.0 hipGraphLaunch) that intermittently corrupted a few pixels
        # of the FIRST replay's flow -- in about half of all fresh processes the stereo scale of one frame of the second batch moved by
        # 3-4 % (scripts/debug/flaky_once.py; tests/test_benched_frontend_gpu.py was flaky for it).  With one instance and one batch
        # ahead, or N instances and N batches ahead, the wait returns at once.
        prev = st['done']
        if prev is not None and os.environ.get('ISLAM_GRAPH_FENCE', '1') == '1':
            prev.synchronize()
        for d, t in zip(static_in, imgs):
            d.copy_(t, non_blocking=True)
        g.replay()
        out = tuple(t.clone() for t in static_out)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(imgs[0].device))
        st['done'] = ev
        return out

    def frozen_forward(self, img0, img1, img0_norm, img0_r_norm):
        """Flow + disparity (Network/VONet.py:28-34).  With both nets frozen this part carries no autograd state, so
        TartanVO.prefetch can run it for the NEXT batch on a side stream while the current batch is optimised."""
        if getattr(self, 'graph_frozen', False) and self.frozen_dtype is not None and self.flow_dtype is not None and \
                not self.frozen_nets_trainable():
            return self._frozen_graphed((img0, img1, img0_norm, img0_r_norm))
        return self._frozen_eager(img0, img1, img0_norm, img0_r_norm)

    def _frozen_eager(self, img0, img1, img0_norm, img0_r_norm, fork=None):
        if fork is not None:
            # the two nets are independent: captured as two parallel branches of the graph (fork / join on a second stream), the
            # launch-bound stretches of one (pyramid levels 3-6 and the small decoder levels of the flow net, the hourglass stack of the
            # stereo net) run beside the large convolutions of the other
            cur = torch.cuda.current_stream(img0.device)
            fork.wait_stream(cur)
            with torch.cuda.stream(fork):
                disp = self._run_frozen_stereo_pair(img0_norm, img0_r_norm, quarter=True)[0]
                disp = disp.float().contiguous()
            flow = self._run_frozen('flow', self.flowNet, self.flow_dtype, torch.cat([img0, img1], 1))[0][0].float().contiguous()
            cur.wait_stream(fork)
            return flow, disp
        flow = self._run_frozen('flow', self.flowNet, self.flow_dtype, torch.cat([img0, img1], 1))[0][0]
        # Network/VONet.py:33-34 keeps disp[..., ::4, ::4] (nearest, scale 1/4): only those pixels are computed (StereoNet7.forward)
        disp = self._run_frozen_stereo_pair(img0_norm, img0_r_norm, quarter=True)[0]
        return flow.float().contiguous(), disp.float().contiguous()

    def forward(self, img0, img1, img0_norm, img0_r_norm, intrinsic, frozen=None):
        flow, disp = frozen if frozen is not None else self.frozen_forward(img0, img1, img0_norm, img0_r_norm)
        x = torch.cat([flow, intrinsic], 1)
        if self.pose_channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        if self.graph_pose == 'hip' and x.is_cuda and not x.requires_grad:
            # the whole head -- forward now, backward when the loss is back-propagated -- on the kernels of csrc/pose_head.hip, each way one
            # HIP graph replay; the backward node adds the parameter gradients to .grad itself (pose_graph_leaf() as for 'accumulate')
            if not getattr(self.flowPoseNet, 'use_hip_head', False):
                self.flowPoseNet.set_hip_head(True, graphs=True)
            pose = self.flowPoseNet(x)
        elif self.graph_pose == 'accumulate' and self.flowPoseNet.training and torch.is_grad_enabled() and x.is_cuda and not x.requires_grad:
            # forward and backward as two HIP graphs whose backward node adds the parameter gradients to .grad itself (_PoseGraph).
            # Only for an input that carries no gradient (frozen flow net: the reference detaches flow behind the pose head,
            # TartanVO.py:109, so with a TRAINABLE flow net the pose loss must reach it -- that case takes the eager branch below).
            with self._pose_autocast():
                pg = self.__dict__.get('_pose_graph')
                plist = pg.all_params if pg is not None else list(self.flowPoseNet.parameters())
                key = (tuple(x.shape), self.pose_dtype, self.pose_channels_last, tuple(p.requires_grad for p in plist))
                if pg is not None and pg.key != key and tuple(x.shape) == pg.key[0]:
                    pg = None                                  # pose_dtype / layout / requires_grad changed: capture again
                if pg is None:
                    pg = self.__dict__['_pose_graph'] = _PoseGraph(self.flowPoseNet, x)
                    pg.key = key
                    self._pose_graphed = tuple(x.shape)
                pose = _PoseGraphFn.apply(x, pg.leaf, pg) if tuple(x.shape) == self._pose_graphed else self.flowPoseNet(x)
        elif self.graph_pose and self.graph_pose != 'accumulate' and self.flowPoseNet.training and torch.is_grad_enabled():
            # forward AND backward of the trainable pose head replay from two captured HIP graphs
            # (torch.cuda.make_graphed_callables patches the module's forward; eval mode keeps the eager path, and so does a
            # batch of another shape, e.g. the last one of an epoch)
            with self._pose_autocast():
                if self._pose_graphed is None:
                    self._pose_eager = self.flowPoseNet.forward
                    torch.cuda.make_graphed_callables(self.flowPoseNet, (x.detach().clone(),))
                    self._pose_graphed = tuple(x.shape)
                pose = self.flowPoseNet(x) if tuple(x.shape) == self._pose_graphed else self._pose_eager(x)
        else:
            with self._pose_autocast():
                pose = self.flowPoseNet(x)
        return flow, disp, pose.float()

    def pose_graph_leaf(self):
        """The zero-dimensional leaf that keeps the graphed pose head's backward node alive (graph_pose='accumulate'): callers of
        torch.autograd.grad(loss, inputs) list it among ``inputs`` or the engine prunes the node and the head gets no gradient.
        None while no graph has been captured."""
        if self.graph_pose == 'hip':
            hh = self.flowPoseNet.__dict__.get('_hip_head')
            return None if hh is None else hh.leaf
        pg = self.__dict__.get('_pose_graph')
        return None if pg is None else pg.leaf

    def _drop_pose_graph(self):
        self.__dict__.pop('_pose_graph', None)
        if self.graph_pose == 'accumulate':
            self._pose_graphed = None

    def _pose_autocast(self):
        import contextlib
        if self.pose_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast('cuda', dtype=self.pose_dtype, cache_enabled=False)   # (no weight-cast cache: graph capture)


class _PoseGraph:
    """Forward and backward of the trainable pose head as two captured HIP graphs (what torch.cuda.make_graphed_callables builds), driven
    by an autograd node that ACCUMULATES the parameter gradients itself instead of handing ~110 tensors back to the autograd engine: per
    returned gradient the engine does stream bookkeeping (event record + wait) and runs a capture hook or an AccumulateGrad node -- 1-2 ms
    of host time per backward, at the end of the batch where nothing else keeps the GPU busy (scripts/vio_gpu_busy.py: 1.3 ms without a
    running kernel per pipelined step).  Same arithmetic as AccumulateGrad: p.grad <- p.grad + g in fp32 (first time: a copy).
    ``leaf`` is a zero-dimensional tensor that requires grad: it keeps the node in the graph, and a caller that asks
    torch.autograd.grad for explicit inputs must list it (BilevelLoop._accumulate_gradients does) or the engine prunes the node."""

    def __init__(self, net, x):
        dev = x.device
        self.all_params = list(net.parameters())
        self.params = [p for p in self.all_params if p.requires_grad]
        self.static_x = x.detach().clone()
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                    # MIOpen's kernel choice and lazy initialisation: outside the capture
            for _ in range(3):
                y = net(self.static_x)
                torch.autograd.grad(y, self.params, torch.ones_like(y), allow_unused=True)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        pool = torch.cuda.graph_pool_handle()
        self.fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=pool):
            self.static_y = net(self.static_x)
        self.static_gy = torch.zeros_like(self.static_y)
        self.bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd, pool=pool):
            self.static_grads = torch.autograd.grad(self.static_y, self.params, self.static_gy, allow_unused=True)
        self.leaf = torch.zeros((), device=dev, requires_grad=True)
        self.zero = torch.zeros((), device=dev)
        self.gen = 0                                     # serial number of the latest forward replay (see _PoseGraphFn)
        self.key = None


class _PoseGraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, leaf, pg):
        # the backward graph reads the activations the LAST forward replay left in the graph's static buffers.  A forward that never
        # gets a backward (metrics pass, an exception between the VO forward and the gradient step) is harmless; what must not happen is
        # forward A, forward B, backward of A -- A's activations are gone.  Every replay takes a generation number and the backward
        # checks that its forward is still the latest one (the reference loop, train.py:212-283, has no such restriction because eager
        # autograd keeps every forward's activations alive).
        pg.gen += 1
        ctx.gen = pg.gen
        pg.static_x.copy_(x)
        pg.fwd.replay()
        ctx.pg = pg
        return pg.static_y.detach().clone()

    @staticmethod
    def backward(ctx, gy):
        pg = ctx.pg
        if ctx.gen != pg.gen:
            raise RuntimeError('graph_pose="accumulate": backward of a forward whose activations a later forward replay has overwritten '
                               '(forward #%d, latest #%d); run forwards that need no gradient under torch.no_grad() / '
                               'TartanVO(..., need_grad=False), or back-propagate before the next forward' % (ctx.gen, pg.gen))
        pg.static_gy.copy_(gy)
        pg.bwd.replay()
        acc, new = [], []
        for p, g in zip(pg.params, pg.static_grads):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g.detach().clone()              # (static buffer of the graph: never keep it)
            else:
                acc.append(p.grad)
                new.append(g)
        if acc:
            torch._foreach_add_(acc, new)
        return None, pg.zero, None


class _HalfExec:
    """Reduced-precision channels_last execution copy of a frozen network.  Conv / deconv / linear parameters are cast ONCE
    (re-cast only when the master's parameters change: load_state_dict, .to(device)); BatchNorm layers stay fp32 (MIOpen's
    mixed mode: bf16 activations, fp32 scale / bias / statistics) and SHARE parameters and running-statistic buffers with the
    master, so train-mode updates land in the checkpointed module."""

    def __init__(self, master, dtype):
        self.master, self.dtype = master, dtype
        self._copy, self._key = None, None

    def module(self):
        params = list(self.master.parameters())
        key = (params[0].device, tuple(p._version for p in params), tuple(p.data_ptr() for p in params[:4]))
        if self._copy is None or key != self._key:
            import copy
            # process groups (dist_train.ShardedBatchNorm2d.group) are shared, not copied
            memo = {id(m.group): m.group for m in self.master.modules() if getattr(m, 'group', None) is not None}
            c = copy.deepcopy(self.master, memo)
            for p in c.parameters():
                p.requires_grad_(False)
            c = c.to(self.dtype).to(memory_format=torch.channels_last)
            masters = dict(self.master.named_modules())
            for n, m in c.named_modules():
                if isinstance(m, nn.BatchNorm2d):
                    src = masters[n]
                    m.float()
                    m.weight, m.bias = src.weight, src.bias
                    m.running_mean, m.running_var, m.num_batches_tracked = src.running_mean, src.running_var, src.num_batches_tracked
            self._copy, self._key = c, key
        self._copy.train(self.master.training)
        return self._copy


# ------------------------------------------------------------------------------------------ IMU denoiser
class IMUCorrector_CNN_GRU_WO_COV(nn.Module):
    def __init__(self, in_channel=6, out_channel=64, hidden_size=128, kernel_size=10, num_layers=1):
        super().__init__()
        self.hidden_size, self.num_layers = hidden_size, num_layers
        self.conv1 = nn.Conv1d(in_channel, out_channel, kernel_size=kernel_size, stride=10)
        self.gelu = nn.GELU()
        self.gru = nn.GRU(out_channel, hidden_size, num_layers, batch_first=True)
        self.encoder = nn.Sequential(self.conv1, nn.GELU(), self.gru)      # aliases kept for state-dict key parity
        self.pose_decoder = nn.Sequential(nn.Linear(hidden_size, 64), nn.GELU(), nn.Linear(64, 6), nn.GELU())

    def forward(self, data, eval=True):
        self.train() if not eval else self.eval()
        with torch.set_grad_enabled(not eval):
            acc, gyro = data['acc'], data['gyro']
            if acc.dim() == 2:
                acc, gyro = acc.unsqueeze(0), gyro.unsqueeze(0)
            x = self.gelu(self.conv1(torch.cat([acc, gyro], -1).permute(0, 2, 1))).permute(0, 2, 1)
            out = self.pose_decoder(self.gru(x)[0])
            rep = torch.full((out.shape[1],), 10, dtype=torch.long, device=acc.device)
            rep[-1] = acc.shape[1] - 10 * out.shape[1] + 10       # the last step also covers the remainder (Q14)
            out = torch.repeat_interleave(out, rep, dim=1)
            return (out[..., 0:3] + acc).squeeze(0), (out[..., 3:6] + gyro).squeeze(0), None, None
