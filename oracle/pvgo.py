"""Oracle (test infrastructure): PVGO = pose-velocity graph optimisation.

CPU restatement (numpy) of
  * reference pvgo.py:26-64   PoseVelGraph.forward   (4 residual blocks)
  * reference pvgo.py:67-78   vo_loss, :95-111 imu_loss, :114-119 align_to
  * reference pvgo.py:122-205 run_pvgo               (weights, LM loop, outputs)
  * PyPose (NOT in /root/reference; unpinned, v0.6.x era -- "parity unpinned"):
      pp.optim.LM.step, ppos.Cholesky, ppost.TrustRegion, StopOnPlateau,
      LieTensor.add_ (left-multiplicative retraction) and the autograd
      conventions that define the Jacobian (SURVEY.md section 8a box, F11).

Two linear-algebra modes share every other line:
  mode='dense'  : J (rows x 10N), dense A = J^T W J, dense Cholesky -- what
                  PyPose does (the faithful CPU baseline).
  mode='banded' : same normal equations in node-major order (9 per node) solved
                  with a banded Cholesky -- the fair CPU baseline for the chain.
"""
import numpy as np
import scipy.linalg as sla

from . import lie
from . import reproj as oreproj


# ------------------------------------------------------------------ residuals
def residuals(nodes, vels, edges, poses, drots, dtrans, dvels, dts):
    """pvgo.py:26-64.  Returns (pgerr (E,6), adjvelerr (M,3), imuroterr (M,3), transvelerr (M,3))."""
    n1 = nodes[edges[:, 0]]
    n2 = nodes[edges[:, 1]]
    err = lie.se3_mul(lie.se3_mul(lie.se3_inv(poses), lie.se3_inv(n1)), n2)      # pvgo.py:38
    pgerr = lie.se3_log(err)
    adjvelerr = dvels - np.diff(vels, axis=0)                                     # pvgo.py:42
    r1 = nodes[:-1, 3:]
    r2 = nodes[1:, 3:]
    rerr = lie.quat_mul(lie.quat_mul(lie.quat_inv(drots), lie.quat_inv(r1)), r2)  # pvgo.py:47
    imuroterr = lie.so3_log(rerr)
    transvelerr = np.diff(nodes[:, :3], axis=0) - (vels[:-1] * dts.reshape(-1, 1) + dtrans)  # pvgo.py:51
    return pgerr, adjvelerr, imuroterr, transvelerr


def loss_unweighted(res):
    """RobustModel.loss with the trivial kernel: sum of squared residuals, no weights."""
    return float(sum(np.sum(np.square(r.astype(np.float64))) for r in res)) if res[0].dtype == np.float64 \
        else float(sum(np.sum(np.square(r)) for r in res))


# ------------------------------------------------------------------ jacobian blocks
def jac_blocks(nodes, edges, poses, drots, pgerr, imuroterr):
    """Left-perturbation Jacobian blocks as PyPose's autograd produces them.

    A_e (E,6,6): d pgerr / d delta_j = Jl^-1(e) Ad(P^-1 X_i^-1);  d/d delta_i = -A_e
    B_k (M,3,3): d imuroterr / d dphi_{k+1} = Jl^-1(e_R) R(dR^-1 R_k^-1); d/d dphi_k = -B_k
    """
    n1 = nodes[edges[:, 0]]
    pre = lie.se3_mul(lie.se3_inv(poses), lie.se3_inv(n1))
    A = lie.se3_Jl_inv(pgerr) @ lie.se3_adj(pre)
    rpre = lie.quat_mul(lie.quat_inv(drots), lie.quat_inv(nodes[:-1, 3:]))
    B = lie.so3_Jl_inv(imuroterr) @ lie.quat_matrix(rpre)
    return A, B


def jacobian_dense(N, edges, A, B, dts, true_translation_jacobian=False, nodes=None):
    """Dense J, PyPose layout: rows [pgerr 6E | adjvel 3M | imurot 3M | transvel 3M],
    columns [nodes.flatten (7N) | vels.flatten (3N)]; 7th pose column stays 0.

    compat (default): transvelerr uses d t / d delta = [I 0] (raw .translation() slice, F11).
    """
    E, M = edges.shape[0], N - 1
    rows = 6 * E + 9 * M
    J = np.zeros((rows, 10 * N), dtype=A.dtype)
    for e in range(E):
        i, j = int(edges[e, 0]), int(edges[e, 1])
        J[6 * e:6 * e + 6, 7 * j:7 * j + 6] += A[e]
        J[6 * e:6 * e + 6, 7 * i:7 * i + 6] -= A[e]
    r0 = 6 * E
    I3 = np.eye(3, dtype=A.dtype)
    vc = 7 * N
    for k in range(M):
        # adjvelerr = dv - (v_{k+1} - v_k)
        J[r0 + 3 * k:r0 + 3 * k + 3, vc + 3 * (k + 1):vc + 3 * (k + 1) + 3] = -I3
        J[r0 + 3 * k:r0 + 3 * k + 3, vc + 3 * k:vc + 3 * k + 3] = I3
    r1 = r0 + 3 * M
    for k in range(M):
        J[r1 + 3 * k:r1 + 3 * k + 3, 7 * (k + 1) + 3:7 * (k + 1) + 6] = B[k]
        J[r1 + 3 * k:r1 + 3 * k + 3, 7 * k + 3:7 * k + 6] = -B[k]
    r2 = r1 + 3 * M
    for k in range(M):
        J[r2 + 3 * k:r2 + 3 * k + 3, 7 * (k + 1):7 * (k + 1) + 3] = I3
        J[r2 + 3 * k:r2 + 3 * k + 3, 7 * k:7 * k + 3] = -I3
        J[r2 + 3 * k:r2 + 3 * k + 3, vc + 3 * k:vc + 3 * k + 3] = -dts[k] * I3
        if true_translation_jacobian:
            J[r2 + 3 * k:r2 + 3 * k + 3, 7 * (k + 1) + 3:7 * (k + 1) + 6] = -lie.skew(nodes[k + 1, :3])
            J[r2 + 3 * k:r2 + 3 * k + 3, 7 * k + 3:7 * k + 6] = lie.skew(nodes[k, :3])
    return J


def weight_vector(E, M, loss_weight, dtype, reproj_n=0):
    """pvgo.py:125-162: all information matrices are scalar*I, so W is diagonal.
    Order follows the model outputs: [VO 6E | imu_vel 3M | imu_rot 3M | transvel 3M | reproj 2*n*M (pvgo.py:130-131)]."""
    lw = loss_weight
    w = [np.full(6 * E, lw[0] ** 2), np.full(3 * M, lw[1] ** 2), np.full(3 * M, lw[2] ** 2), np.full(3 * M, lw[3] ** 2)]
    if reproj_n:
        w.append(np.full(2 * reproj_n * M, (lw[4] / reproj_n) ** 2))
    return np.concatenate(w).astype(dtype)


def retract(nodes, vels, D_nodes6, D_vels):
    """LieTensor.add_: X <- Exp(d[:6]) * X ; vels += d."""
    return lie.se3_mul(lie.se3_exp(D_nodes6), nodes).astype(nodes.dtype), (vels + D_vels).astype(vels.dtype)


# ------------------------------------------------------------------ LM
class TrustRegion:
    """ppost.TrustRegion (radius given by pvgo.py:170; other values PyPose defaults)."""

    def __init__(self, radius=1e6, high=0.5, low=1e-3, up=2.0, down=0.5, factor=0.5, min=1e-6, max=1e16):
        self.pg = dict(radius=radius, high=high, low=low, up=up, down=down, factor=factor,
                       damping=1.0 / radius)
        self.down0, self.min, self.max = down, min, max

    def update(self, last, loss, JD, R):
        pg = self.pg
        den = -float(JD @ (2 * R + JD))
        with np.errstate(divide='ignore', invalid='ignore'):
            quality = np.float64(last - loss) / np.float64(den)
        pg['radius'] = 1.0 / pg['damping']
        if quality > pg['high']:
            pg['radius'] = pg['up'] * pg['radius']
            pg['down'] = self.down0
        elif quality > pg['low']:
            pg['down'] = self.down0
        else:
            pg['radius'] = pg['radius'] * pg['down']
            pg['down'] = pg['down'] * pg['factor']
        pg['down'] = max(self.min, min(pg['down'], self.max))
        pg['radius'] = max(self.min, min(pg['radius'], self.max))
        pg['damping'] = 1.0 / pg['radius']
        return quality


class _DenseLin:
    """PyPose's linear algebra: dense J, dense A = J^T W J, dense Cholesky (ppos.Cholesky)."""

    def __init__(self, nodes, inp, res, A_e, B_k, w, ttj, J_rp=None):
        edges, dts = inp[0], inp[5]
        N = nodes.shape[0]
        self.N, self.dt = N, nodes.dtype
        self.J = jacobian_dense(N, edges, A_e, B_k, dts, ttj, nodes)
        if J_rp is not None:                                  # 5th output: link k couples nodes k, k+1 (pvgo.py:54-56)
            M, n2 = J_rp.shape[0], J_rp.shape[1]
            Jr = np.zeros((M * n2, 10 * N), dtype=self.J.dtype)
            for k in range(M):
                Jr[n2 * k:n2 * (k + 1), 7 * (k + 1):7 * (k + 1) + 6] = J_rp[k]
                Jr[n2 * k:n2 * (k + 1), 7 * k:7 * k + 6] = -J_rp[k]
            self.J = np.concatenate([self.J, Jr], 0)
        R = np.concatenate([r.reshape(-1) for r in res])
        JTW = self.J.T * w[None, :]
        self.A = JTW @ self.J
        self.b = -(JTW @ R)
        self.diag = np.diagonal(self.A).copy()

    def solve(self, d):
        np.fill_diagonal(self.A, d)
        L = np.linalg.cholesky(self.A)
        if np.any(np.isnan(L)):
            raise np.linalg.LinAlgError('NaN in Cholesky factor')
        D = sla.cho_solve((L, True), self.b).astype(self.dt)
        N = self.N
        return D[:7 * N].reshape(N, 7)[:, :6], D[7 * N:].reshape(N, 3)

    def JD(self, Dn, Dv):
        N = self.N
        D = np.concatenate([np.concatenate([Dn, np.zeros((N, 1), self.dt)], 1).reshape(-1), Dv.reshape(-1)])
        return self.J @ D


class _BandedLin:
    """Same normal equations, node-major order [rho phi v] (pad column dropped: its pivot is the
    clamp value 1e-4 and its right-hand side 0, so its step is exactly 0), canonical chain
    links [k, k+1] only; banded Cholesky (bandwidth 17)."""

    def __init__(self, nodes, inp, res, A_e, B_k, w4, ttj, J_rp=None, w_rp=0.0):
        edges, dts = inp[0], np.asarray(inp[5]).reshape(-1)
        N = nodes.shape[0]
        M = N - 1
        assert not ttj, 'banded mode implements the PyPose-compat Jacobian only'
        assert edges.shape[0] == M and np.all(edges[:, 0] == np.arange(M)) and np.all(edges[:, 1] == np.arange(1, N)), \
            'banded mode needs canonical chain links [k, k+1]'
        dt = nodes.dtype
        self.N, self.dt, self.A_e, self.B_k, self.dts = N, dt, A_e, B_k, dts
        w0, w1, w2, w3 = [dt.type(x) for x in w4]
        e, rv, er, rt = res[:4]
        self.J_rp = J_rp
        I3 = np.eye(3, dtype=dt)
        S = w0 * (np.swapaxes(A_e, 1, 2) @ A_e)
        S[:, :3, :3] += w3 * I3
        S[:, 3:, 3:] += w2 * (np.swapaxes(B_k, 1, 2) @ B_k)
        if J_rp is not None:
            S += dt.type(w_rp) * (np.swapaxes(J_rp, 1, 2) @ J_rp)
        Hd = np.zeros((N, 9, 9), dt)
        Ho = np.zeros((M, 9, 9), dt)
        Hd[:-1, :6, :6] += S
        Hd[1:, :6, :6] += S
        Ho[:, :6, :6] = -S
        Hd[:-1, 6:, 6:] += (w1 + w3 * dts * dts)[:, None, None] * I3
        Hd[1:, 6:, 6:] += w1 * I3
        Ho[:, 6:, 6:] = -w1 * I3
        c = (w3 * dts)[:, None, None] * I3
        Hd[:-1, 0:3, 6:9] += c
        Hd[:-1, 6:9, 0:3] += c
        Ho[:, 6:9, 0:3] += -c
        gp = w0 * (np.swapaxes(A_e, 1, 2) @ e[:, :, None])[:, :, 0]
        gp[:, 3:] += w2 * (np.swapaxes(B_k, 1, 2) @ er[:, :, None])[:, :, 0]
        gp[:, :3] += w3 * rt
        if J_rp is not None:
            gp += dt.type(w_rp) * (np.swapaxes(J_rp, 1, 2) @ res[4][:, :, None])[:, :, 0]
        g = np.zeros((N, 9), dt)
        g[1:, :6] += gp
        g[:-1, :6] -= gp
        g[:-1, 6:] += w1 * rv - (w3 * dts)[:, None] * rt
        g[1:, 6:] -= w1 * rv
        self.b = (-g).reshape(-1)
        ab = np.zeros((18, 9 * N), dt)
        for r in range(9):
            for cc in range(9):
                if r >= cc:
                    ab[r - cc, cc::9] = Hd[:, r, cc]
                # coupling rows = node k+1 unknown cc', cols = node k unknown r': A[9(k+1)+cc, 9k+r] = Ho[k][r][cc]
                ab[9 + cc - r, r:9 * M:9] = Ho[:, r, cc]
        self.ab = ab
        self.diag = ab[0].copy()

    def solve(self, d):
        self.ab[0, :] = d
        D = sla.solveh_banded(self.ab, self.b, lower=True).astype(self.dt).reshape(self.N, 9)
        return D[:, :6], D[:, 6:]

    def JD(self, Dn, Dv):
        dp = Dn[1:] - Dn[:-1]
        jd_vo = (self.A_e @ dp[:, :, None])[:, :, 0]
        jd_av = Dv[:-1] - Dv[1:]
        jd_rot = (self.B_k @ dp[:, 3:, None])[:, :, 0]
        jd_tv = dp[:, :3] - self.dts[:, None] * Dv[:-1]
        out = [jd_vo.reshape(-1), jd_av.reshape(-1), jd_rot.reshape(-1), jd_tv.reshape(-1)]
        if self.J_rp is not None:
            out.append((self.J_rp @ dp[:, :, None]).reshape(-1))
        return np.concatenate(out)


class LM:
    """pp.optim.LM with Cholesky solver + TrustRegion strategy, as constructed at pvgo.py:169-171."""

    def __init__(self, nodes, vels, radius=1e4, vmin=1e-4, vmax=1e32, reject=16, mode='dense',
                 true_translation_jacobian=False, reproj=None, compat_first_motion=True, info_scalars=None):
        self.nodes, self.vels = nodes.copy(), vels.copy()
        self.reproj, self.compat_first = reproj, compat_first_motion
        self.info_scalars = info_scalars      # tests only: the four information scalars directly (may be indefinite)
        self.strategy = TrustRegion(radius=radius)
        self.min, self.max, self.reject = vmin, vmax, reject
        self.reject_count = 0
        self.mode = mode
        self.ttj = true_translation_jacobian
        self.loss = None
        self.last = None
        self.trace = []   # (loss, damping, accepted) per inner iteration
        self.step_losses = []   # what step() returned, one per optimizer step (what the scheduler sees)

    def _res(self, inp):
        res = residuals(self.nodes, self.vels, *inp)
        if self.reproj is not None:                          # pvgo.py:53-61
            res = res + (oreproj.residual(self.reproj, self.nodes, self.compat_first),)
        return res

    def step(self, inp, loss_weight):
        edges, poses, drots, dtrans, dvels, dts = inp
        E, M = edges.shape[0], self.nodes.shape[0] - 1
        dt = self.nodes.dtype
        res = self._res(inp)
        R = np.concatenate([r.reshape(-1) for r in res])
        A_e, B_k = jac_blocks(self.nodes, edges, poses, drots, res[0], res[2])
        J_rp, n_rp = None, 0
        if self.reproj is not None:
            J_rp, n_rp = oreproj.jac_link(self.reproj, self.nodes, self.compat_first), self.reproj.N
        if self.mode == 'dense':
            lin = _DenseLin(self.nodes, inp, res, A_e, B_k, weight_vector(E, M, loss_weight, dt, n_rp), self.ttj, J_rp)
        else:
            w4 = [x ** 2 for x in loss_weight[:4]] if self.info_scalars is None else list(self.info_scalars)
            lin = _BandedLin(self.nodes, inp, res, A_e, B_k, w4, self.ttj, J_rp,
                             (loss_weight[4] / n_rp) ** 2 if n_rp else 0.0)
        if self.loss is None:
            self.loss = loss_unweighted(res)
        self.last = self.loss
        d = np.clip(lin.diag, self.min, self.max)            # A.diagonal().clamp_(min, max)
        self.reject_count = 0
        pg = self.strategy.pg
        while self.last <= self.loss:
            d = d + d * pg['damping']                        # cumulative across retries
            try:
                Dn, Dv = lin.solve(d)
            except (np.linalg.LinAlgError, sla.LinAlgError) as e:   # PyPose: print, break
                print(e, '\nLinear solver failed. Breaking optimization step...')
                break
            self.nodes, self.vels = retract(self.nodes, self.vels, Dn, Dv)
            self.loss = loss_unweighted(self._res(inp))
            self.strategy.update(self.last, self.loss, lin.JD(Dn, Dv), R)
            if self.last < self.loss and self.reject_count < self.reject:
                self.nodes, self.vels = retract(self.nodes, self.vels, -Dn, -Dv)
                self.trace.append((self.loss, pg['damping'], False))
                self.loss, self.reject_count = self.last, self.reject_count + 1
            else:
                self.trace.append((self.loss, pg['damping'], True))
                break
        self.step_losses.append(self.loss)
        return self.loss


class StopOnPlateau:
    """pypose.optim.scheduler.StopOnPlateau as used at pvgo.py:172,177-180."""

    def __init__(self, optimizer, steps, patience=5, decreasing=1e-3):
        self.opt, self.max_steps, self.patience, self.decreasing = optimizer, steps, patience, decreasing
        self.steps, self.patience_count, self._continual = 0, 0, True

    def continual(self):
        return self._continual

    def step(self, loss):
        self.steps += 1
        if self.steps >= self.max_steps:
            self._continual = False
        if (self.opt.last - loss) < self.decreasing:
            self.patience_count += 1
        else:
            self.patience_count = 0
        if self.patience_count >= self.patience:
            self._continual = False
        if self.opt.reject_count >= self.opt.reject:
            self._continual = False


# ------------------------------------------------------------------ losses / alignment
def vo_loss(nodes, edges, poses):
    """pvgo.py:67-78 -> (trans_loss (E,), rot_loss (E,), e (E,6))."""
    err = lie.se3_mul(lie.se3_mul(lie.se3_inv(poses), lie.se3_inv(nodes[edges[:, 0]])), nodes[edges[:, 1]])
    e = lie.se3_log(err)
    return np.sum(e[:, :3] ** 2, 1), np.sum(e[:, 3:] ** 2, 1), e


def vo_loss_grad(nodes, edges, poses, g_trans, g_rot):
    """Gradient of sum(g_trans*trans_loss + g_rot*rot_loss) w.r.t. ``poses`` in PyPose's
    convention: left-tangent 6-vector padded to 7."""
    _, _, e = vo_loss(nodes, edges, poses)
    ge = np.concatenate([2 * e[:, :3] * g_trans[:, None], 2 * e[:, 3:] * g_rot[:, None]], 1)
    gE = (ge[:, None, :] @ lie.se3_Jl_inv(e))[:, 0, :]
    gP = -(gE[:, None, :] @ lie.se3_adj(lie.se3_inv(poses)))[:, 0, :]
    return np.concatenate([gP, np.zeros_like(gP[:, :1])], 1)


def imu_loss(nodes, vels, drots, dvels):
    """pvgo.py:95-111."""
    adj = dvels - np.diff(vels, axis=0)
    rerr = lie.quat_mul(lie.quat_mul(lie.quat_inv(drots), lie.quat_inv(nodes[:-1, 3:])), nodes[1:, 3:])
    r = lie.so3_log(rerr)
    return np.sum(adj ** 2, 1), np.sum(r ** 2, 1)


def align_to(nodes, vels, target, idx=0):
    """pvgo.py:114-119."""
    src = nodes[idx]
    rel = lie.se3_mul(target, lie.se3_inv(src))
    rq = lie.quat_mul(target[3:], lie.quat_inv(src[3:]))
    return lie.se3_mul(rel[None, :], nodes), lie.quat_act(rq[None, :], vels)


def run_pvgo(init_nodes, init_vels, vo_motions, links, dts, imu_drots, imu_dtrans, imu_dvels,
             radius=1e4, loss_weight=(1, 1, 1, 1), target='vo', mode='dense', dtype=np.float64,
             true_translation_jacobian=False, max_steps=10, return_optimizer=False, reproj=None,
             compat_first_motion=True, info_scalars=None):
    """pvgo.py:122-205 on numpy arrays.  Returns (trans_loss, rot_loss, nodes, vels, covs[, optimizer])."""
    c = lambda a: np.ascontiguousarray(np.asarray(a), dtype=dtype)
    init_nodes, init_vels, vo_motions = c(init_nodes), c(init_vels), c(vo_motions)
    dts, imu_drots, imu_dtrans, imu_dvels = c(dts), c(imu_drots), c(imu_dtrans), c(imu_dvels)
    links = np.asarray(links, dtype=np.int64)
    opt = LM(init_nodes, init_vels, radius=radius, vmin=1e-4, mode=mode,
             true_translation_jacobian=true_translation_jacobian, reproj=reproj, compat_first_motion=compat_first_motion,
             info_scalars=info_scalars)
    sched = StopOnPlateau(opt, steps=max_steps, patience=3, decreasing=1e-3)
    inp = (links, vo_motions, imu_drots, imu_dtrans, imu_dvels, dts)
    while sched.continual():
        loss = opt.step(inp, loss_weight)
        sched.step(loss)
    if target == 'vo':
        tl, rl, _ = vo_loss(opt.nodes, links, vo_motions)
    else:
        tl, rl = imu_loss(opt.nodes, opt.vels, imu_drots, imu_dvels)
    nodes, vels = align_to(opt.nodes, opt.vels, init_nodes[0])
    n = len(init_nodes) - 1
    covs = {'vo_rot': np.ones(len(links)) * loss_weight[0] ** 2, 'imu_rot': np.ones(n) * loss_weight[2] ** 2,
            'vo_trans': np.ones(len(links)) * loss_weight[0] ** 2, 'imu_vel': np.ones(n) * loss_weight[1] ** 2,
            'transvel': np.ones(n) * loss_weight[3] ** 2}
    if reproj is not None:
        covs['reproj'] = np.ones(n) * (loss_weight[4] / reproj.N) ** 2                   # pvgo.py:131,202-203
    out = (tl, rl, nodes, vels, covs)
    return out + (opt,) if return_optimizer else out
