"""General-topology PVGO (arbitrary ``links``: loop closures, skipped frames) on the GPU -- SURVEY.md section 8f rank 4.

The chain fast path (islam_amd/csrc/pvgo.hip) needs links[k] = [k, k+1].  For any other edge set this module runs the
same LM (pp.optim.LM + Cholesky + TrustRegion + StopOnPlateau as constructed at pvgo.py:169-172) with the same dense
normal matrix PyPose factorises, but never forms the Jacobian PyPose multiplies (rows x 10N):

  * residuals and Jacobian blocks per factor from the HIP kernels (islam_pvgo_linearize_edges for the VO factors on
    arbitrary edges, islam_pvgo_linearize for the IMU factors, which always couple consecutive nodes);
  * A = J^T W J (9N x 9N) and b = -J^T W r assembled block by block on the device (islam_pvgo_assemble_dense):
    O(E) work instead of the 2*rows*cols^2 dense product;
  * fp64 POTRF / POTRS from rocSOLVER (the only O(N^3) piece; its trailing updates are fp64 MFMA GEMMs);
  * retraction, trial residuals and the trust-region term (J D)^T (2R + J D) from the per-factor blocks.
Sized by the dense matrix: (9N)^2 doubles (N = 5001: 16 GB of the 288 GB).

Long trajectories with a few loop closures do not need the dense matrix at all (run_lm_band_pcg): the IMU factors always
couple consecutive nodes and a VO edge (i, j) adds S = w J^T J to the diagonal blocks of i and j and -S to the block (i, j), so
  A = B + R,  B = block-tridiagonal (all diagonal blocks, the couplings |i-j| = 1),  R = the off-band blocks of the k long edges,
B is positive definite on its own, rank(R) <= 12 k, and conjugate gradients preconditioned with B^-1 -- the block-tridiagonal
solver of the chain path, islam_pvgo_solve_chain, ~60 us per application at N = 5001 -- reaches the solution of the SAME
normal equations in at most 12 k + 1 iterations.  Same LM control, same trial / trust-region evaluation; memory O(N + k)."""
import numpy as np
import torch

from . import ops
from ._lib import c_double, check, lib, ptr, stream_ptr
from .lm_control import LMControl

_NOCLAMP = 1e300


def _linearize(nodes, vels, edges, poses, drots, dtrans, dvels, dts, dummy_poses):
    E = edges.shape[0]
    vo = torch.empty((24, E), dtype=torch.float64, device=nodes.device)
    check(lib().islam_pvgo_linearize_edges(ptr(nodes), ptr(edges), ptr(poses), E, ptr(vo), stream_ptr(nodes.device)))
    lin, _ = ops.pvgo_linearize(nodes, vels, dummy_poses, drots, dtrans, dvels, dts)     # IMU rows of `lin`; VO rows unused
    return vo, lin


def _loss(vo, lin):
    """Unweighted sum of squares over the model outputs (pvgo.py:64): pgerr | adjvelerr | imuroterr | transvelerr."""
    return (vo[0:6] ** 2).sum() + (lin[36:39] ** 2).sum() + (lin[24:27] ** 2).sum() + (lin[39:42] ** 2).sum()


def _node_adjacency(edges_host, N):
    """CSR of edge ends per node, entry = 2*edge + end, ascending: fixes the summation order of the diagonal blocks."""
    E = edges_host.shape[0]
    node = edges_host.reshape(-1)                          # entry index = 2*e + end
    order = np.argsort(node, kind='stable')
    ptr_ = np.zeros(N + 1, dtype=np.int64)
    np.add.at(ptr_, node + 1, 1)
    return np.cumsum(ptr_), order.astype(np.int64)


def _quality_term(vo, lin, edges, dts, D):
    """(J D)^T (2 R + J D) with the unweighted J, R of the linearisation point, factor by factor."""
    D6, Dv = D[:, :6], D[:, 6:]
    E = edges.shape[0]
    dp = D6[edges[:, 1]] - D6[edges[:, 0]]
    G = vo[6:15].t().reshape(E, 3, 3)
    C = vo[15:24].t().reshape(E, 3, 3)
    mv = lambda M, v: (M @ v[:, :, None])[:, :, 0]
    j0 = mv(G, dp[:, :3]) + mv(C, dp[:, 3:])
    j1 = mv(G, dp[:, 3:])
    dc = D6[1:] - D6[:-1]
    M = dc.shape[0]
    B = lin[27:36].t().reshape(M, 3, 3)
    j2 = Dv[:-1] - Dv[1:]
    j3 = mv(B, dc[:, 3:])
    j4 = dc[:, :3] - dts[:, None] * Dv[:-1]
    R0, R1 = vo[0:3].t(), vo[3:6].t()
    R2, R3, R4 = lin[36:39].t(), lin[24:27].t(), lin[39:42].t()
    q = 0.0
    for j, r in ((j0, R0), (j1, R1), (j2, R2), (j3, R3), (j4, R4)):
        q = q + (j * (2 * r + j)).sum()
    return q


class _ReprojTerms:
    """The sparse reprojection factor (pvgo.py:53-61; 5th residual) of one linearisation point, in NODE coordinates.  The factor
    always couples consecutive nodes k, k+1 (motion = nodes[:-1].Inv() @ nodes[1:], whatever `links` holds), so its blocks land
    in the chain arrays (Hd, Ho, rhs) of either general-topology solver.  islam_pvgo_reproj_reduce gives, per link, J^T J / J^T r
    / r^T r in the coordinates eta of a left perturbation of T_k = C^-1 (X_k^-1 X_{k+1}) C; eta = M (delta_{k+1} - delta_k),
    M = Ad(C^-1 X_k^-1) = [[R, [t]x R], [0, R]] (reproj_link in csrc/pvgo.hip)."""

    def __init__(self, nodes, rp, dx=None):
        red = ops.pvgo_reproj_reduce(nodes, rp, dx)
        self.rr = red[:, 27].sum()                                        # unweighted r^T r over all links
        if dx is not None:
            return                                                        # trial point: only the loss is needed
        Mn, dev = red.shape[0], nodes.device
        iu = torch.triu_indices(6, 6, device=dev)
        S = torch.zeros((Mn, 6, 6), dtype=torch.float64, device=dev)
        S[:, iu[0], iu[1]] = red[:, :21]
        S = S + S.transpose(1, 2) - torch.diag_embed(S.diagonal(dim1=1, dim2=2))
        b = red[:, 21:27]
        from . import lietensor as pp
        C = pp.SE3(torch.tensor([float(v) for v in rp.rgb2imu], dtype=torch.float64, device=dev))
        Y = C.Inv() @ pp.SE3(nodes[:-1]).Inv()
        R = Y.rotation().matrix()
        T = pp._skew(Y.translation()) @ R
        Mm = torch.zeros((Mn, 6, 6), dtype=torch.float64, device=dev)
        Mm[:, :3, :3], Mm[:, :3, 3:], Mm[:, 3:, 3:] = R, T, R
        Mt = Mm.transpose(1, 2)
        self.S = Mt @ S @ Mm                                               # (M,6,6) unweighted J^T J w.r.t. (delta_{k+1} - delta_k)
        self.g = (Mt @ b[:, :, None])[:, :, 0]                             # (M,6)   unweighted J^T r
        self.w = float(rp.weight)

    def add_to_chain(self, Hd, Ho, rhs):
        """+w S on the pose blocks of nodes k and k+1, -w S on their coupling, b = -J^T W r: +w g at k, -w g at k+1."""
        wS, wg = self.w * self.S, self.w * self.g
        Hd[:-1, :6, :6] += wS
        Hd[1:, :6, :6] += wS
        Ho[:-1, :6, :6] -= wS
        rhs[:-1, :6] += wg
        rhs[1:, :6] -= wg

    def quality(self, D):
        """(J D)^T (2 R + J D) of the reprojection rows (unweighted, like the other factors)."""
        d = D[1:, :6] - D[:-1, :6]
        return (d * (2 * self.g + (self.S @ d[:, :, None])[:, :, 0])).sum()


def run_lm_dense(nodes, vels, edges, poses, drots, dtrans, dvels, dts, loss_weight, radius=1e4, max_steps=10, patience=3,
                 decreasing=1e-3, vmin=1e-4, vmax=1e32, reproj=None):
    """In: float64 contiguous device tensors; reproj: ops.pvgo_reproj_struct or None.  Returns (nodes, vels, info dict)."""
    N, E, M = nodes.shape[0], edges.shape[0], nodes.shape[0] - 1
    if E != M:
        raise ValueError('PoseVelGraph needs as many VO edges as IMU intervals (dts broadcasts over both, pvgo.py:51): E=%d, N-1=%d' % (E, M))
    dev = nodes.device
    w = [float(x) ** 2 for x in loss_weight[:4]]
    dummy = torch.zeros((M, 7), dtype=torch.float64, device=dev)
    dummy[:, 6] = 1.0
    nptr, nadj = _node_adjacency(edges.cpu().numpy(), N)
    nptr, nadj = torch.from_numpy(nptr).to(dev), torch.from_numpy(nadj).to(dev)
    A = torch.empty((9 * N, 9 * N), dtype=torch.float64, device=dev)
    b = torch.empty((9 * N,), dtype=torch.float64, device=dev)
    ctl = LMControl(radius=radius, max_steps=max_steps, patience=patience, decreasing=decreasing)
    trials = 0
    while ctl.continual:
        vo, lin = _linearize(nodes, vels, edges, poses, drots, dtrans, dvels, dts, dummy)
        rpt = _ReprojTerms(nodes, reproj) if reproj is not None else None
        if not ctl.has_loss:
            ctl.set_initial_loss(float(_loss(vo, lin) + (rpt.rr if rpt is not None else 0.0)))
        ctl.begin_step()
        Hd, Ho, rhs = ops.pvgo_build_normal(lin, dts, N, (0.0, w[1], w[2], w[3]), -_NOCLAMP, _NOCLAMP)     # IMU factors
        if rpt is not None:
            rpt.add_to_chain(Hd, Ho, rhs)
        check(lib().islam_pvgo_assemble_dense(ptr(Hd), ptr(Ho), ptr(rhs), ptr(vo), ptr(edges), ptr(nptr), ptr(nadj),
                                              c_double(w[0]), N, E, ptr(A), ptr(b), stream_ptr(dev)))
        d = A.diagonal().clamp(vmin, vmax).clone()                    # A.diagonal().clamp_(min, max)
        while True:
            d = d + d * ctl.damping                                   # cumulative, like A.diagonal().add_(...)
            A.diagonal().copy_(d)
            L, info = torch.linalg.cholesky_ex(A)
            trials += 1
            if int(info) != 0 or not bool(torch.isfinite(L.diagonal()).all()):
                print('Linear solver failed. Breaking optimization step...')
                ctl.solver_failed()
                break
            D = torch.cholesky_solve(b[:, None], L)[:, 0].view(N, 9).contiguous()
            del L
            nt, vt = ops.pvgo_retract(nodes, vels, D, 1.0)
            vo_t, lin_t = _linearize(nt, vt, edges, poses, drots, dtrans, dvels, dts, dummy)
            st, qt = _loss(vo_t, lin_t), _quality_term(vo, lin, edges, dts, D)
            if rpt is not None:
                st, qt = st + _ReprojTerms(nodes, reproj, D).rr, qt + rpt.quality(D)
            s, q = torch.stack([st, qt]).tolist()
            if ctl.after_trial(s, q):
                nodes, vels = nt, vt
                break
        ctl.end_step()
    return nodes, vels, dict(steps=ctl.steps, trials=trials, loss=ctl.loss, trace=ctl.trace)


# ------------------------------------------------------------------------------------------ band + low-rank (PCG)
def _edge_blocks(vo, w0):
    """Per VO edge: S = w0 J^T J (E,6,6) and g = w0 J^T r (E,6), J = [[G, C], [0, G]] (edge_normal in csrc/pvgo.hip)."""
    E = vo.shape[1]
    G = vo[6:15].t().reshape(E, 3, 3)
    C = vo[15:24].t().reshape(E, 3, 3)
    J = torch.zeros((E, 6, 6), dtype=vo.dtype, device=vo.device)
    J[:, :3, :3], J[:, :3, 3:], J[:, 3:, 3:] = G, C, G
    r = vo[0:6].t()
    Jt = J.transpose(1, 2)
    return w0 * (Jt @ J), w0 * (Jt @ r[:, :, None])[:, :, 0]


def off_band_edges(edges_host):
    """Indices of the edges that couple nodes more than one apart (the part of A outside the block-tridiagonal band)."""
    d = np.abs(edges_host[:, 1] - edges_host[:, 0])
    return np.nonzero(d > 1)[0]


class _BandSystem:
    """A = B + R of one linearisation: B as (Hd, Ho) of the chain solver, R as the list of off-band 6x6 blocks."""

    def __init__(self, vo, lin, edges, dts, N, w, vmin, vmax, off_idx, rpt=None):
        dev = vo.device
        Hd, Ho, rhs = ops.pvgo_build_normal(lin, dts, N, (0.0, w[1], w[2], w[3]), -_NOCLAMP, _NOCLAMP)     # IMU factors
        if rpt is not None:
            rpt.add_to_chain(Hd, Ho, rhs)
        S, g = _edge_blocks(vo, w[0])
        E = S.shape[0]
        i, j = edges[:, 0], edges[:, 1]
        Sp = torch.zeros((E, 9, 9), dtype=S.dtype, device=dev)
        Sp[:, :6, :6] = S
        gp = torch.zeros((E, 9), dtype=S.dtype, device=dev)
        gp[:, :6] = g
        Hd.index_add_(0, i, Sp)
        Hd.index_add_(0, j, Sp)
        rhs.index_add_(0, i, gp)                     # b = -J^T W r: +g at the edge's first node, -g at its second
        rhs.index_add_(0, j, -gp)
        adj = (j - i).abs() == 1
        Ho.index_add_(0, torch.minimum(i, j)[adj], -Sp[adj])
        self.Hd, self.Ho, self.b = Hd, Ho, rhs
        self.io, self.jo, self.So = i[off_idx], j[off_idx], S[off_idx]
        self.diag0 = Hd.diagonal(dim1=1, dim2=2).clamp(vmin, vmax).clone()                  # A.diagonal().clamp_(min, max)

    def set_diagonal(self, d):
        self.Hd.diagonal(dim1=1, dim2=2).copy_(d)

    def matvec(self, p):
        y = (self.Hd @ p[:, :, None])[:, :, 0]
        y[:-1] += (self.Ho[:-1] @ p[1:, :, None])[:, :, 0]
        y[1:] += (self.Ho[:-1].transpose(1, 2) @ p[:-1, :, None])[:, :, 0]
        if self.io.numel():
            y6 = torch.zeros((p.shape[0], 6), dtype=p.dtype, device=p.device)
            y6.index_add_(0, self.io, (self.So @ p[self.jo, :6, None])[:, :, 0])
            y6.index_add_(0, self.jo, (self.So @ p[self.io, :6, None])[:, :, 0])
            y[:, :6] -= y6
        return y


PCG_FAIL_RTOL = 1e-8      # true relative residual above which a PCG solve counts as failed


def _pcg(sysm, ws, rtol=1e-13, check_every=6):
    """Solve (B + R) x = b with conjugate gradients preconditioned by B^-1 (the block-tridiagonal solver, enqueued without
    host round trips; the residual is looked at every `check_every` iterations).  Raises IslamHipError (ISLAM_ENOTPD) if B
    is not positive definite."""
    N, dev = sysm.b.shape[0], sysm.b.device
    Minv = lambda r: ops.pvgo_solve_chain_enqueue(sysm.Hd, sysm.Ho, r.contiguous(), ws)
    b = sysm.b
    x = Minv(b)
    ops.pvgo_solve_status(N, ws, dev)                      # B = L D L^T went through: positive definite
    k = int(sysm.io.numel())
    if k == 0:
        return x, 0, 0.0
    zero = torch.zeros((), dtype=b.dtype, device=dev)
    safe_div = lambda a, c: torch.where(c != 0, a / torch.where(c != 0, c, torch.ones_like(c)), zero)
    r = b - sysm.matvec(x)
    tol = rtol * float(b.norm())
    z = Minv(r)
    p = z.clone()
    rz = (r * z).sum()
    its = 0
    for its in range(1, 12 * k + 12):
        Hp = sysm.matvec(p)
        alpha = safe_div(rz, (p * Hp).sum())               # (0/0 once the residual is exactly zero)
        x = x + alpha * p
        r = r - alpha * Hp
        if its % check_every == 0 and float(r.norm()) <= tol:
            break
        z = Minv(r)
        rz_new = (r * z).sum()
        p = z + safe_div(rz_new, rz) * p
        rz = rz_new
    ops.pvgo_solve_status(N, ws, dev)
    # the TRUE residual of what is returned (the recurrence above can drift, and the loop may have hit its iteration cap)
    rel = float((b - sysm.matvec(x)).norm()) / max(float(b.norm()), 1e-300)
    return x, its, rel


def run_lm_band_pcg(nodes, vels, edges, poses, drots, dtrans, dvels, dts, loss_weight, radius=1e4, max_steps=10, patience=3,
                    decreasing=1e-3, vmin=1e-4, vmax=1e32, reproj=None):
    """The LM of run_lm_dense on the band + low-rank form of the same normal equations.  In: float64 contiguous device
    tensors.  Returns (nodes, vels, info dict)."""
    from ._lib import IslamHipError
    N, E, M = nodes.shape[0], edges.shape[0], nodes.shape[0] - 1
    if E != M:
        raise ValueError('PoseVelGraph needs as many VO edges as IMU intervals (dts broadcasts over both, pvgo.py:51): E=%d, N-1=%d' % (E, M))
    dev = nodes.device
    w = [float(x) ** 2 for x in loss_weight[:4]]
    dummy = torch.zeros((M, 7), dtype=torch.float64, device=dev)
    dummy[:, 6] = 1.0
    off_idx = torch.from_numpy(off_band_edges(edges.cpu().numpy())).to(dev)
    ws = ops.pvgo_workspace(N, dev)
    try:
        ops.pvgo_solve_status(N, ws, dev)                  # initialises the fresh workspace's status / hand-off words
    except IslamHipError:
        pass
    ctl = LMControl(radius=radius, max_steps=max_steps, patience=patience, decreasing=decreasing)
    trials = pcg_its = 0
    pcg_worst = 0.0
    while ctl.continual:
        vo, lin = _linearize(nodes, vels, edges, poses, drots, dtrans, dvels, dts, dummy)
        rpt = _ReprojTerms(nodes, reproj) if reproj is not None else None
        if not ctl.has_loss:
            ctl.set_initial_loss(float(_loss(vo, lin) + (rpt.rr if rpt is not None else 0.0)))
        ctl.begin_step()
        sysm = _BandSystem(vo, lin, edges, dts, N, w, vmin, vmax, off_idx, rpt)
        d = sysm.diag0
        while True:
            d = d + d * ctl.damping                                   # cumulative, like A.diagonal().add_(...)
            sysm.set_diagonal(d)
            trials += 1
            try:
                D, its, rel = _pcg(sysm, ws)
            except IslamHipError:
                print('Linear solver failed. Breaking optimization step...')
                ctl.solver_failed()
                break
            pcg_its += its
            pcg_worst = max(pcg_worst, rel)
            # an unconverged step must not reach LM silently: treat it like a failed factorisation (PyPose's Cholesky would
            # have returned the exact solution or raised)
            if not bool(torch.isfinite(D).all()) or rel > PCG_FAIL_RTOL:
                print('Linear solver failed. Breaking optimization step...')
                ctl.solver_failed()
                break
            nt, vt = ops.pvgo_retract(nodes, vels, D.contiguous(), 1.0)
            vo_t, lin_t = _linearize(nt, vt, edges, poses, drots, dtrans, dvels, dts, dummy)
            st, qt = _loss(vo_t, lin_t), _quality_term(vo, lin, edges, dts, D)
            if rpt is not None:
                st, qt = st + _ReprojTerms(nodes, reproj, D.contiguous()).rr, qt + rpt.quality(D)
            s, q = torch.stack([st, qt]).tolist()
            if ctl.after_trial(s, q):
                nodes, vels = nt, vt
                break
        ctl.end_step()
    return nodes, vels, dict(steps=ctl.steps, trials=trials, loss=ctl.loss, trace=ctl.trace, pcg_iterations=pcg_its,
                             pcg_worst_relative_residual=pcg_worst, off_band_edges=int(off_idx.numel()))
