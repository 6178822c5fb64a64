"""The trainable pose head (reference Network/VOFlowNet.py:42-194, VOFlowRes of config 1) on the hand-written fp32 kernels of
csrc/pose_head.hip: forward and backward of the WHOLE module are one C call each (islam_pose_head_forward / _backward).

``PoseHeadHip`` wraps a ``nets.VOFlowRes`` module -- same parameters, same state dict, nothing copied: the kernels read the weights where
torch keeps them (channels_last convolution weights) -- and owns what the calls need: the activation workspace, the parameter /
gradient pointer tables, one flat buffer the backward writes this batch's gradients into, and (optionally) two captured HIP graphs.

Autograd: ``head(x)`` returns a (B,6) tensor whose backward node ACCUMULATES the parameter gradients into ``p.grad`` itself (like
``nets._PoseGraphFn``: handing 120 tensors back to the engine costs 1-2 ms of host time per step).  The node hangs on ``head.leaf``, a
zero-dimensional tensor that requires grad: ``loss.backward()`` reaches it by itself; a caller of ``torch.autograd.grad(loss, inputs)``
lists it among ``inputs`` (``VONet.pose_graph_leaf()``; ``BilevelLoop._accumulate_gradients`` does).  No gradient flows to ``x``: a pose
head fed by a TRAINABLE flow net takes the eager modules (``VONet.forward``).
"""
import ctypes

import torch

from ._lib import c_size_t, c_void_p, check, lib, stream_ptr

N_PARAMS = 120


def supported_input(x):
    """True if the kernels serve this input: an fp32 (B,4,H,W) channels-last device tensor whose 1/64-size feature map has 6 pixels, B <= 16."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 4 and x.shape[0] <= 16):
        return False
    if not x.is_contiguous(memory_format=torch.channels_last) or torch.is_autocast_enabled():
        return False
    return lib().islam_pose_head_workspace_bytes(int(x.shape[0]), int(x.shape[2]), int(x.shape[3])) != 0


def supported(module, x):
    """supported_input(x) and parameters the kernels can read in place: fp32 on x's device, convolution weights channels_last."""
    if not supported_input(x):
        return False
    for p in module.parameters():
        if p.dtype != torch.float32 or p.device != x.device:
            return False
        if p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last):
            return False
        if p.dim() != 4 and not p.is_contiguous():
            return False
    return True


class PoseHeadHip:
    def __init__(self, module, graphs=False):
        self.module = module
        self.graphs = bool(graphs)
        self.gen = 0                    # serial number of the latest forward (the backward reads ITS activations from the workspace)
        self._key = None
        self._ws = None
        self._graph = {}
        self.leaf = None
        self.x_saved = None

    # ---- tables -------------------------------------------------------------------------------------------------------------------
    def _prepare(self, x):
        params = list(self.module.parameters())
        if len(params) != N_PARAMS:
            raise RuntimeError('PoseHeadHip: the module has %d parameters, VOFlowRes of config 1 has %d' % (len(params), N_PARAMS))
        B, _, H, W = x.shape
        key = (x.device, int(B), int(H), int(W), tuple(p.data_ptr() for p in params))
        if key == self._key:
            return
        dev = x.device
        if not supported(self.module, x):
            raise RuntimeError('PoseHeadHip: fp32 parameters on %s with channels_last convolution weights and an fp32 channels-last (B<=16,4,H,W) '
                               'input are needed (VONet.set_pose_channels_last(True))' % (dev,))
        nbytes = int(lib().islam_pose_head_workspace_bytes(int(B), int(H), int(W)))
        if nbytes == 0:
            raise RuntimeError('PoseHeadHip: unsupported input shape %s' % (tuple(x.shape),))
        if self._ws is None or self._ws.numel() * 4 < nbytes or self._ws.device != dev:
            self._ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev)       # (ticket words must start at zero)
        self.params = params
        self._ptab = (c_void_p * N_PARAMS)(*[p.data_ptr() for p in params])
        # this batch's gradients: one flat buffer, one view per parameter in the parameter's own memory layout
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        if getattr(self, 'gflat', None) is None or self.gflat.numel() != total or self.gflat.device != dev:
            self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)       # (zeros: the alignment gaps between the views are never written)
            self.acc_flat = None
        self._offs, self._total = offs, total
        self.gviews = self._views(self.gflat)
        self._gtab = (c_void_p * N_PARAMS)(*[g.data_ptr() for g in self.gviews])
        if self.leaf is None or self.leaf.device != dev:
            self.leaf = torch.zeros((), device=dev, requires_grad=True)
            self.zero = torch.zeros((), device=dev)
        self._graph = {}
        self._key = key

    def _views(self, flat):
        out = []
        for p, o in zip(self.params, self._offs):
            v = flat[o:o + p.numel()]
            if p.dim() == 4:
                O, I, kh, kw = p.shape
                v = v.view(O, kh, kw, I).permute(0, 3, 1, 2)          # shape (O,I,kh,kw), channels_last strides: the parameter's layout
            else:
                v = v.view(p.shape)
            out.append(v)
        return out

    # ---- raw calls ----------------------------------------------------------------------------------------------------------------
    def _c_forward(self, x, out):
        B, _, H, W = x.shape
        check(lib().islam_pose_head_forward(c_void_p(x.data_ptr()), ctypes.cast(self._ptab, c_void_p), c_void_p(out.data_ptr()),
                                            c_void_p(self._ws.data_ptr()), c_size_t(self._ws.numel() * 4), int(B), int(H), int(W),
                                            stream_ptr(x.device)))

    def _c_backward(self, x, gy):
        B, _, H, W = x.shape
        check(lib().islam_pose_head_backward(c_void_p(x.data_ptr()), ctypes.cast(self._ptab, c_void_p), ctypes.cast(self._gtab, c_void_p),
                                             c_void_p(gy.data_ptr()), c_void_p(self._ws.data_ptr()), c_size_t(self._ws.numel() * 4),
                                             int(B), int(H), int(W), 0, stream_ptr(x.device)))

    def _graphed(self, x):
        """Two captured HIP graphs (forward, backward) over static input / output buffers: the ~50 / ~100 launches of a call cost the host
        one graph launch."""
        g = self._graph
        if not g:
            dev = x.device
            g['x'] = x.detach().clone(memory_format=torch.channels_last)
            g['y'] = torch.empty(x.shape[0], 6, dtype=torch.float32, device=dev)
            g['gy'] = torch.zeros(x.shape[0], 6, dtype=torch.float32, device=dev)
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):                     # first calls outside the capture (function attributes, lazy module load)
                self._c_forward(g['x'], g['y'])
                self._c_backward(g['x'], g['gy'])
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            g['fwd'] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g['fwd']):
                self._c_forward(g['x'], g['y'])
            g['bwd'] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g['bwd']):
                self._c_backward(g['x'], g['gy'])
        return g

    def forward_raw(self, x):
        self._prepare(x)
        self.gen += 1
        if self.graphs:
            g = self._graphed(x)
            g['x'].copy_(x)
            g['fwd'].replay()
            self.x_saved = g['x']
            return g['y'].clone()
        out = torch.empty(x.shape[0], 6, dtype=torch.float32, device=x.device)
        self._c_forward(x, out)
        self.x_saved = x
        return out

    def backward_raw(self, gy):
        """This batch's parameter gradients into self.gviews (overwritten), then added to p.grad."""
        gy = gy.to(torch.float32).contiguous()
        if self.graphs:
            g = self._graph
            g['gy'].copy_(gy)
            g['bwd'].replay()
        else:
            self._c_backward(self.x_saved, gy)
        self._accumulate()

    def _accumulate(self):
        ps = self.params
        want = [p.requires_grad for p in ps]
        if self.acc_flat is not None and all((not w) or (p.grad is not None and p.grad.data_ptr() == a.data_ptr()) for p, w, a in zip(ps, want, self.acc_views)) \
                and all(want):
            self.acc_flat.add_(self.gflat)                     # every .grad is our view of one flat buffer: ONE launch
            return
        if all(want) and all(p.grad is None for p in ps):
            self.acc_flat = self.gflat.clone()
            self.acc_views = self._views(self.acc_flat)
            for p, a in zip(ps, self.acc_views):
                p.grad = a
            return
        acc, new = [], []
        for p, w, g in zip(ps, want, self.gviews):
            if not w:
                continue
            if p.grad is None:
                p.grad = g.detach().clone(memory_format=torch.preserve_format)
            else:
                acc.append(p.grad)
                new.append(g)
        if acc:
            torch._foreach_add_(acc, new)

    # ---- autograd -----------------------------------------------------------------------------------------------------------------
    def __call__(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.module.parameters()):
            if x.requires_grad:
                raise RuntimeError('PoseHeadHip produces no gradient w.r.t. its input; use the eager modules for a trainable flow net')
            self._prepare(x)
            return _PoseHeadFn.apply(x, self.leaf, self)
        return self.forward_raw(x)


class _PoseHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, leaf, head):
        y = head.forward_raw(x)
        ctx.head, ctx.gen = head, head.gen
        return y

    @staticmethod
    def backward(ctx, gy):
        head = ctx.head
        if ctx.gen != head.gen:
            raise RuntimeError('PoseHeadHip: backward of a forward whose activations a later forward has overwritten (forward #%d, latest #%d); '
                               'run forwards that need no gradient under torch.no_grad(), or back-propagate before the next forward'
                               % (ctx.gen, head.gen))
        head.backward_raw(gy)
        return None, head.zero, None
