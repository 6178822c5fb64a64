"""Minimal LieTensor shim: the handful of PyPose names iSLAM's driver and modules touch
(SURVEY.md section 8b "Lie shim"), with PyPose's storage and autograd conventions:

  * SO3 = [qx,qy,qz,qw], SE3 = [t, q], so3 = phi, se3 = [rho, phi]
  * group-op gradients are LEFT-perturbation tangent vectors stored in the first 6 (SE3) / 3 (SO3)
    slots of a storage-shaped gradient, last slot 0
  * ``.translation()`` / ``.rotation()`` / ``.tensor()`` are raw slices / views

Used for the O(batch) glue on (B,7) tensors (train.py:214-240, Datasets/transformation.py:72-124);
the heavy paths (PVGO, IMU, scale recovery) run in the HIP library and never go through this file.
Works on CPU and device tensors alike, exactly like the PyPose calls it stands in for.
"""
import numpy as np
import torch
from torch.utils._pytree import tree_flatten, tree_map


class LieType:
    def __init__(self, name, dim, tangent, group):
        self.name, self.dim, self.tangent, self.group = name, dim, tangent, group

    def __repr__(self):
        return self.name + '_type'


SO3_type = LieType('SO3', 4, 3, True)
SE3_type = LieType('SE3', 7, 6, True)
so3_type = LieType('so3', 3, 3, False)
se3_type = LieType('se3', 6, 6, False)


# ------------------------------------------------------------------------------------------ plain math
# Host fast path.  The bilevel loop keeps its O(batch) pose algebra on the host in float64 (TartanVO(host_glue=True),
# BilevelLoop): ~200 group operations per batch on (8, 7) tensors, each of them a handful of torch CPU ops at 5-10 us of
# dispatch apiece -- 4 ms of a 12 ms step went into them (scripts/vio_hostprof.py).  The group primitives below are only ever
# called inside the autograd.Functions of this file (no graph is recorded through them), so for small host tensors they
# compute on numpy views instead: the same formulas, the same IEEE operations, ~1 us per operation.
_NP_MAX = 4096


def _host(*ts):
    for t in ts:
        if t.is_cuda or t.numel() > _NP_MAX or t.dtype not in (torch.float64, torch.float32):
            return False
    return True


def _np(t):
    return t.detach().numpy()


def _qmul_np(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    o = np.empty(np.broadcast_shapes(a.shape, b.shape), dtype=np.result_type(a, b))
    o[..., 0] = aw * bx + ax * bw + ay * bz - az * by
    o[..., 1] = aw * by - ax * bz + ay * bw + az * bx
    o[..., 2] = aw * bz + ax * by - ay * bx + az * bw
    o[..., 3] = aw * bw - ax * bx - ay * by - az * bz
    return o


def _cross_np(a, b):
    o = np.empty(np.broadcast_shapes(a.shape, b.shape), dtype=np.result_type(a, b))
    o[..., 0] = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    o[..., 1] = a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2]
    o[..., 2] = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    return o


def _qact_np(q, p):
    u, w = q[..., :3], q[..., 3:]
    uv = 2.0 * _cross_np(u, p)
    return p + w * uv + _cross_np(u, uv)


def _qmat_np(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def _skew_np(v):
    z = np.zeros_like(v[..., 0])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1), np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def _skew(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], -1), torch.stack([v[..., 2], z, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def _qmul(a, b):
    if _host(a, b):
        return torch.from_numpy(_qmul_np(_np(a), _np(b)))
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)


def _qinv(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _qact(q, p):
    if _host(q, p) and q.dtype == p.dtype:
        return torch.from_numpy(_qact_np(_np(q), _np(p)))
    u, w = q[..., :3], q[..., 3:]
    uv = 2.0 * torch.linalg.cross(u.expand(torch.broadcast_shapes(u.shape, p.shape)), p.expand(torch.broadcast_shapes(u.shape, p.shape)), dim=-1)
    return p + w * uv + torch.linalg.cross(u.expand(uv.shape), uv, dim=-1)


def _qmat(q):
    if _host(q):
        return torch.from_numpy(_qmat_np(_np(q)))
    x, y, z, w = q.unbind(-1)
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                        torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def _so3_exp(phi):
    th = torch.linalg.norm(phi, dim=-1, keepdim=True)
    th2 = th * th
    th4 = th2 * th2
    big = th > torch.finfo(phi.dtype).eps
    ths = torch.where(big, th, torch.ones_like(th))
    imag = torch.where(big, torch.sin(0.5 * ths) / ths, 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4)
    real = torch.where(big, torch.cos(0.5 * ths), 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4)
    return torch.cat([phi * imag, real], -1)


def _so3_log(q):
    v, w = q[..., :3], q[..., 3:]
    vn = torch.linalg.norm(v, dim=-1, keepdim=True)
    big = vn > torch.finfo(q.dtype).eps
    vns = torch.where(big, vn, torch.ones_like(vn))
    f = torch.where(big, 2.0 * torch.atan(vns / w) / vns, 2.0 / w - (2.0 / 3.0) * (vn * vn) / (w * w * w))
    return f * v


def _coefs(th, fn_big, small):
    big = th > torch.finfo(th.dtype).eps
    ths = torch.where(big, th, torch.ones_like(th))
    return torch.where(big, fn_big(ths), torch.full_like(th, small))


def _so3_Jl(phi):
    K = _skew(phi)
    th = torch.linalg.norm(phi, dim=-1)[..., None, None]
    c1 = _coefs(th, lambda t: (1 - torch.cos(t)) / t ** 2, 0.5)
    c2 = _coefs(th, lambda t: (t - torch.sin(t)) / t ** 3, 1.0 / 6.0)
    return torch.eye(3, dtype=phi.dtype, device=phi.device) + c1 * K + c2 * (K @ K)


def _so3_Jl_inv(phi):
    K = _skew(phi)
    th = torch.linalg.norm(phi, dim=-1)[..., None, None]
    c = _coefs(th, lambda t: (1 - t * torch.cos(0.5 * t) / (2 * torch.sin(0.5 * t))) / t ** 2, 1.0 / 12.0)
    return torch.eye(3, dtype=phi.dtype, device=phi.device) - 0.5 * K + c * (K @ K)


def _se3_Q(xi):
    T, P = _skew(xi[..., :3]), _skew(xi[..., 3:])
    th = torch.linalg.norm(xi[..., 3:], dim=-1)[..., None, None]
    c1 = _coefs(th, lambda t: (t - torch.sin(t)) / t ** 3, 1.0 / 6.0)
    c2 = _coefs(th, lambda t: (t * t + 2 * torch.cos(t) - 2) / (2 * t ** 4), 1.0 / 24.0)
    c3 = _coefs(th, lambda t: (2 * t - 3 * torch.sin(t) + t * torch.cos(t)) / (2 * t ** 5), 1.0 / 120.0)
    PT, TP = P @ T, T @ P
    PTP = PT @ P
    return 0.5 * T + c1 * (PT + TP + PTP) + c2 * (P @ PT + TP @ P - 3 * PTP) + c3 * (PTP @ P + P @ PTP)


def _se3_Jl(xi):
    J, Q = _so3_Jl(xi[..., 3:]), _se3_Q(xi)
    Z = torch.zeros_like(J)
    return torch.cat([torch.cat([J, Q], -1), torch.cat([Z, J], -1)], -2)


def _se3_Jl_inv(xi):
    Ji, Q = _so3_Jl_inv(xi[..., 3:]), _se3_Q(xi)
    Z = torch.zeros_like(Ji)
    return torch.cat([torch.cat([Ji, -Ji @ Q @ Ji], -1), torch.cat([Z, Ji], -1)], -2)


def _se3_adj(X):
    if _host(X):
        x = _np(X)
        R = _qmat_np(x[..., 3:])
        top = np.concatenate([R, _skew_np(x[..., :3]) @ R], -1)
        return torch.from_numpy(np.concatenate([top, np.concatenate([np.zeros_like(R), R], -1)], -2))
    R = _qmat(X[..., 3:])
    Z = torch.zeros_like(R)
    return torch.cat([torch.cat([R, _skew(X[..., :3]) @ R], -1), torch.cat([Z, R], -1)], -2)


def _se3_mul(X, Y):
    if _host(X, Y) and X.dtype == Y.dtype:
        x, y = _np(X), _np(Y)
        t = x[..., :3] + _qact_np(x[..., 3:], y[..., :3])
        q = _qmul_np(x[..., 3:], y[..., 3:])
        return torch.from_numpy(np.concatenate([t, q], -1))
    return torch.cat([X[..., :3] + _qact(X[..., 3:], Y[..., :3]), _qmul(X[..., 3:], Y[..., 3:])], -1)


def _se3_inv(X):
    if _host(X):
        x = _np(X)
        qi = np.concatenate([-x[..., 3:6], x[..., 6:7]], -1)
        return torch.from_numpy(np.concatenate([-_qact_np(qi, x[..., :3]), qi], -1))
    qi = _qinv(X[..., 3:])
    return torch.cat([-_qact(qi, X[..., :3]), qi], -1)


def _pad(g):
    return torch.cat([g, torch.zeros_like(g[..., :1])], -1)


def _unbroadcast(g, shape):
    while g.dim() > len(shape):
        g = g.sum(0)
    for i, s in enumerate(shape):
        if s == 1 and g.shape[i] != 1:
            g = g.sum(i, keepdim=True)
    return g


def _row(g, M):      # row-vector times matrix, batched
    if _host(g, M) and g.dtype == M.dtype:
        return torch.from_numpy(np.einsum('...i,...ij->...j', _np(g), _np(M)))
    return (g.unsqueeze(-2) @ M).squeeze(-2)


# ------------------------------------------------------------------------------------------ autograd (PyPose conventions)
class _SE3Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y):
        ctx.save_for_backward(X, Y)
        return _se3_mul(X, Y)

    @staticmethod
    def backward(ctx, g):
        X, Y = ctx.saved_tensors
        g6 = g[..., :6]
        gY = _row(g6, _se3_adj(X).expand(g6.shape[:-1] + (6, 6)))
        return _unbroadcast(_pad(g6), X.shape), _unbroadcast(_pad(gY), Y.shape)


class _SE3Inv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        Y = _se3_inv(X)
        ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, g):
        (Y,) = ctx.saved_tensors
        return _pad(-_row(g[..., :6], _se3_adj(Y)))


class _SE3Log(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        phi = _so3_log(X[..., 3:])
        out = torch.cat([(_so3_Jl_inv(phi) @ X[..., :3, None]).squeeze(-1), phi], -1)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return _pad(_row(g, _se3_Jl_inv(out)))


class _se3Exp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xi):
        ctx.save_for_backward(xi)
        return torch.cat([(_so3_Jl(xi[..., 3:]) @ xi[..., :3, None]).squeeze(-1), _so3_exp(xi[..., 3:])], -1)

    @staticmethod
    def backward(ctx, g):
        (xi,) = ctx.saved_tensors
        return _row(g[..., :6], _se3_Jl(xi))


class _SE3Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, p):
        out = _qact(X[..., 3:], p) + X[..., :3]
        ctx.save_for_backward(X, out)
        ctx.pshape = p.shape
        return out

    @staticmethod
    def backward(ctx, g):
        X, out = ctx.saved_tensors
        gX = torch.cat([g, -_row(g, _skew(out))], -1)          # d out / d delta = [I, -[out]x]
        gp = _row(g, _qmat(X[..., 3:]).expand(g.shape[:-1] + (3, 3)))
        return _unbroadcast(_pad(gX), X.shape), _unbroadcast(gp, ctx.pshape)


class _SO3Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y):
        ctx.save_for_backward(X, Y)
        return _qmul(X, Y)

    @staticmethod
    def backward(ctx, g):
        X, Y = ctx.saved_tensors
        g3 = g[..., :3]
        gY = _row(g3, _qmat(X).expand(g3.shape[:-1] + (3, 3)))
        return _unbroadcast(_pad(g3), X.shape), _unbroadcast(_pad(gY), Y.shape)


class _SO3Inv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        Y = _qinv(X)
        ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, g):
        (Y,) = ctx.saved_tensors
        return _pad(-_row(g[..., :3], _qmat(Y)))


class _SO3Log(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        out = _so3_log(X)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return _pad(_row(g, _so3_Jl_inv(out)))


class _so3Exp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, phi):
        ctx.save_for_backward(phi)
        return _so3_exp(phi)

    @staticmethod
    def backward(ctx, g):
        (phi,) = ctx.saved_tensors
        return _row(g[..., :3], _so3_Jl(phi))


class _SO3Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, p):
        out = _qact(X, p)
        ctx.save_for_backward(X, out)
        ctx.pshape = p.shape
        return out

    @staticmethod
    def backward(ctx, g):
        X, out = ctx.saved_tensors
        gX = -_row(g, _skew(out))
        gp = _row(g, _qmat(X).expand(g.shape[:-1] + (3, 3)))
        return _unbroadcast(_pad(gX), X.shape), _unbroadcast(gp, ctx.pshape)


# ------------------------------------------------------------------------------------------ the tensor subclass
_KEEP = {'__getitem__', 'to', 'cpu', 'cuda', 'detach', 'clone', 'contiguous', 'float', 'double', 'unsqueeze', 'squeeze',
         'stack', 'cat', 'concatenate', 'unbind', 'index_select', 'requires_grad_', 'view', 'reshape', 'expand',
         'repeat', 'type', 'flip', '__iter__', 'select', 'narrow'}


def _plain(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, LieTensor) else t


class LieTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, data, ltype):
        if not isinstance(data, torch.Tensor):
            data = torch.as_tensor(np.asarray(data))
            if not data.dtype.is_floating_point:
                data = data.to(torch.get_default_dtype())
        assert data.shape[-1] == ltype.dim, 'last dimension %d does not fit %s' % (data.shape[-1], ltype.name)
        t = _plain(data).as_subclass(cls)
        t.ltype = ltype
        return t

    def __init__(self, *a, **k):
        pass

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        # the Lie type of the first LieTensor among the arguments (one level into lists / tuples: torch.cat / stack); the generic
        # pytree walk is kept for anything nested deeper -- it cost ~25 us per call, on ~60 calls per bilevel step
        ltype = None
        for a in args:
            if isinstance(a, LieTensor):
                ltype = getattr(a, 'ltype', None)
                break
            if isinstance(a, (list, tuple)):
                for b in a:
                    if isinstance(b, LieTensor):
                        ltype = getattr(b, 'ltype', None)
                        break
                if ltype is not None:
                    break
        if ltype is None and kwargs:
            for a in tree_flatten(kwargs)[0]:
                if isinstance(a, LieTensor):
                    ltype = getattr(a, 'ltype', None)
                    break
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        name = getattr(func, '__name__', '')
        keep = name in _KEEP
        if isinstance(out, torch.Tensor):                    # the common case: one tensor out
            if keep and ltype is not None and out.dim() >= 1 and out.shape[-1] == ltype.dim and out.dtype.is_floating_point:
                return LieTensor(out, ltype)
            return _plain(out)
        if ltype is None or not keep:
            return tree_map(_plain, out) if not keep else out

        def wrap(o):
            if isinstance(o, torch.Tensor) and o.dim() >= 1 and o.shape[-1] == ltype.dim and o.dtype.is_floating_point:
                return LieTensor(o, ltype)
            return _plain(o)
        return tree_map(wrap, out)

    # --- PyPose surface
    def tensor(self):
        return _plain(self)

    def Inv(self):
        d = _plain(self)
        if self.ltype is SE3_type:
            return LieTensor(_SE3Inv.apply(d), SE3_type)
        if self.ltype is SO3_type:
            return LieTensor(_SO3Inv.apply(d), SO3_type)
        return LieTensor(-d, self.ltype)

    def Exp(self):
        d = _plain(self)
        if self.ltype is se3_type:
            return LieTensor(_se3Exp.apply(d), SE3_type)
        if self.ltype is so3_type:
            return LieTensor(_so3Exp.apply(d), SO3_type)
        raise TypeError('Exp() is defined on so3/se3, not %s' % self.ltype.name)

    def Log(self):
        d = _plain(self)
        if self.ltype is SE3_type:
            return LieTensor(_SE3Log.apply(d), se3_type)
        if self.ltype is SO3_type:
            return LieTensor(_SO3Log.apply(d), so3_type)
        raise TypeError('Log() is defined on SO3/SE3, not %s' % self.ltype.name)

    def rotation(self):
        d = _plain(self)
        if self.ltype is SE3_type:
            return LieTensor(d[..., 3:7], SO3_type)
        if self.ltype is SO3_type:
            return LieTensor(d, SO3_type)
        raise TypeError('rotation() needs SO3/SE3')

    def translation(self):
        d = _plain(self)
        if self.ltype is SE3_type:
            return d[..., 0:3]
        return torch.zeros(d.shape[:-1] + (3,), dtype=d.dtype, device=d.device)

    def matrix(self):
        d = _plain(self)
        if self.ltype is SO3_type:
            return _qmat(d)
        if self.ltype is SE3_type:
            R = _qmat(d[..., 3:])
            top = torch.cat([R, d[..., :3, None]], -1)
            bot = torch.zeros(d.shape[:-1] + (1, 4), dtype=d.dtype, device=d.device)
            bot[..., 0, 3] = 1
            return torch.cat([top, bot], -2)
        raise TypeError('matrix() needs SO3/SE3')

    def Act(self, p):
        d, p = _promote(_plain(self), _plain(p))
        if self.ltype is SE3_type:
            return _SE3Act.apply(d, p)
        if self.ltype is SO3_type:
            return _SO3Act.apply(d, p)
        raise TypeError('Act() needs SO3/SE3')

    def _mul(self, other):
        if isinstance(other, LieTensor):
            a, b = _promote(_plain(self), _plain(other))
            if self.ltype is SE3_type and other.ltype is SE3_type:
                return LieTensor(_SE3Mul.apply(a, b), SE3_type)
            if self.ltype is SO3_type and other.ltype is SO3_type:
                return LieTensor(_SO3Mul.apply(a, b), SO3_type)
            raise TypeError('cannot multiply %s with %s' % (self.ltype.name, other.ltype.name))
        if isinstance(other, torch.Tensor) and other.dim() >= 1 and other.shape[-1] == 3 and self.ltype.group:
            return self.Act(other)
        return _plain(self) * other

    __matmul__ = _mul
    __mul__ = _mul

    def __repr__(self):
        return '%s:\n%s' % (self.ltype, _plain(self).__repr__())


def _promote(a, b):
    """Binary group ops follow torch's type promotion (float32 x float64 -> float64), like plain tensor arithmetic."""
    if a.dtype != b.dtype:
        dt = torch.promote_types(a.dtype, b.dtype)
        a, b = a.to(dt), b.to(dt)
    return a, b


def SE3(data):
    return LieTensor(data, SE3_type)


def SO3(data):
    return LieTensor(data, SO3_type)


def so3(data):
    return LieTensor(data, so3_type)


def se3(data):
    return LieTensor(data, se3_type)


def identity_SO3(*size, **kw):
    q = torch.zeros(tuple(size) + (4,), **kw)
    q[..., 3] = 1
    return LieTensor(q, SO3_type)


def identity_SE3(*size, **kw):
    x = torch.zeros(tuple(size) + (7,), **kw)
    x[..., 6] = 1
    return LieTensor(x, SE3_type)


def from_matrix(m, ltype):
    """4x4 / 3x3 (nested lists or tensors) -> SE3 / SO3 (Shepperd's method, w >= 0)."""
    m = torch.as_tensor(m, dtype=torch.get_default_dtype()) if not isinstance(m, torch.Tensor) else m
    R = m[..., :3, :3]
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    w = torch.sqrt(torch.clamp(1 + tr, min=0)) / 2
    x = torch.sqrt(torch.clamp(1 + R[..., 0, 0] - R[..., 1, 1] - R[..., 2, 2], min=0)) / 2
    y = torch.sqrt(torch.clamp(1 - R[..., 0, 0] + R[..., 1, 1] - R[..., 2, 2], min=0)) / 2
    z = torch.sqrt(torch.clamp(1 - R[..., 0, 0] - R[..., 1, 1] + R[..., 2, 2], min=0)) / 2
    sgn = lambda v: torch.where(v < 0, -torch.ones_like(v), torch.ones_like(v))
    x = x * sgn(R[..., 2, 1] - R[..., 1, 2])
    y = y * sgn(R[..., 0, 2] - R[..., 2, 0])
    z = z * sgn(R[..., 1, 0] - R[..., 0, 1])
    q = torch.stack([x, y, z, w], -1)
    if ltype is SO3_type:
        return LieTensor(q, SO3_type)
    return LieTensor(torch.cat([m[..., :3, 3], q], -1), SE3_type)


def Parameter(data):
    return torch.nn.Parameter(data)
