"""General-topology PVGO (arbitrary ``links``: loop closures, skipped frames) on the GPU -- SURVEY.md section 8f rank 4.

The chain fast path (islam_amd/csrc/pvgo.hip) needs links[k] = [k, k+1].  For any other edge set this module runs the
same LM (pp.optim.LM + TrustRegion + StopOnPlateau as constructed at pvgo.py:169-172) the way PyPose does it -- dense
Jacobian, dense J^T W J, dense Cholesky -- but with every O(rows) piece on the device: residuals and Jacobian blocks from
the HIP kernels (islam_pvgo_linearize_edges for the VO factors, islam_pvgo_linearize for the IMU factors, which always
couple consecutive nodes), the fp64 GEMM / POTRF / POTRS from rocBLAS / rocSOLVER (fp64 MFMA), the retraction from
islam_pvgo_retract.  Sized for N up to a few thousand nodes (the dense matrix is (9N)^2 doubles)."""
import torch

from . import ops
from ._lib import check, lib, ptr, stream_ptr
from .lm_control import LMControl


def _linearize(nodes, vels, edges, poses, drots, dtrans, dvels, dts, dummy_poses):
    E = edges.shape[0]
    vo = torch.empty((24, E), dtype=torch.float64, device=nodes.device)
    check(lib().islam_pvgo_linearize_edges(ptr(nodes), ptr(edges), ptr(poses), E, ptr(vo), stream_ptr(nodes.device)))
    lin, _ = ops.pvgo_linearize(nodes, vels, dummy_poses, drots, dtrans, dvels, dts)     # IMU rows of `lin`; VO rows unused
    return vo, lin


def _residual_vector(vo, lin):
    # model output order (pvgo.py:64): pgerr (6E) | adjvelerr (3M) | imuroterr (3M) | transvelerr (3M)
    return torch.cat([vo[0:6].t().reshape(-1), lin[36:39].t().reshape(-1), lin[24:27].t().reshape(-1), lin[39:42].t().reshape(-1)])


def _dense_jacobian(N, edges, vo, lin, dts):
    """Rows as _residual_vector, columns node-major [rho phi v] (the always-zero pad column of PyPose's 7-slot pose is dropped)."""
    dev = vo.device
    E, M = edges.shape[0], N - 1
    rows = 6 * E + 9 * M
    J = torch.zeros((rows, 9 * N), dtype=torch.float64, device=dev)
    G = vo[6:15].t().reshape(E, 3, 3)
    C = vo[15:24].t().reshape(E, 3, 3)
    A = torch.zeros((E, 6, 6), dtype=torch.float64, device=dev)
    A[:, :3, :3], A[:, :3, 3:], A[:, 3:, 3:] = G, C, G
    ar6 = torch.arange(6, device=dev)
    r = (6 * torch.arange(E, device=dev)[:, None, None] + ar6[None, :, None]).expand(E, 6, 6)
    for col, sign in ((edges[:, 1], 1.0), (edges[:, 0], -1.0)):
        c = (9 * col[:, None, None] + ar6[None, None, :]).expand(E, 6, 6)
        J.index_put_((r, c), sign * A, accumulate=True)
    ar3 = torch.arange(3, device=dev)
    k = torch.arange(M, device=dev)
    I3 = torch.eye(3, dtype=torch.float64, device=dev).expand(M, 3, 3)
    B = lin[27:36].t().reshape(M, 3, 3)

    def put(row0, node, off, block):
        rr = (row0 + 3 * k[:, None, None] + ar3[None, :, None]).expand(M, 3, 3)
        cc = (9 * node[:, None, None] + off + ar3[None, None, :]).expand(M, 3, 3)
        J.index_put_((rr, cc), block, accumulate=True)
    r0 = 6 * E
    put(r0, k + 1, 6, -I3)                       # adjvelerr = dv - (v_{k+1} - v_k)
    put(r0, k, 6, I3)
    r1 = r0 + 3 * M
    put(r1, k + 1, 3, B)                         # imuroterr
    put(r1, k, 3, -B)
    r2 = r1 + 3 * M
    put(r2, k + 1, 0, I3)                        # transvelerr (raw-slice translation Jacobian)
    put(r2, k, 0, -I3)
    put(r2, k, 6, -dts[:, None, None] * I3)
    return J


def run_lm_dense(nodes, vels, edges, poses, drots, dtrans, dvels, dts, loss_weight, radius=1e4, max_steps=10, patience=3,
                 decreasing=1e-3, vmin=1e-4, vmax=1e32):
    """In: float64 contiguous device tensors.  Returns (nodes, vels, info dict)."""
    N, E, M = nodes.shape[0], edges.shape[0], nodes.shape[0] - 1
    if E != M:
        raise ValueError('PoseVelGraph needs as many VO edges as IMU intervals (dts broadcasts over both, pvgo.py:51): E=%d, N-1=%d' % (E, M))
    dev = nodes.device
    w = torch.cat([torch.full((6 * E,), float(loss_weight[0]) ** 2), torch.full((3 * M,), float(loss_weight[1]) ** 2),
                   torch.full((3 * M,), float(loss_weight[2]) ** 2), torch.full((3 * M,), float(loss_weight[3]) ** 2)]).to(dev, torch.float64)
    dummy = torch.zeros((M, 7), dtype=torch.float64, device=dev)
    dummy[:, 6] = 1.0
    ctl = LMControl(radius=radius, max_steps=max_steps, patience=patience, decreasing=decreasing)
    trials = 0
    while ctl.continual:
        vo, lin = _linearize(nodes, vels, edges, poses, drots, dtrans, dvels, dts, dummy)
        R = _residual_vector(vo, lin)
        J = _dense_jacobian(N, edges, vo, lin, dts)
        if not ctl.has_loss:
            ctl.set_initial_loss(float((R * R).sum()))
        ctl.begin_step()
        JTW = J.t() * w[None, :]
        A = JTW @ J
        b = -(JTW @ R)
        d = A.diagonal().clamp(vmin, vmax).clone()
        while True:
            d = d + d * ctl.damping                                   # cumulative, like A.diagonal().add_(...)
            A.diagonal().copy_(d)
            L, info = torch.linalg.cholesky_ex(A)
            trials += 1
            if int(info) != 0 or not bool(torch.isfinite(L).all()):
                print('Linear solver failed. Breaking optimization step...')
                ctl.solver_failed()
                break
            D = torch.cholesky_solve(b[:, None], L)[:, 0]
            nt, vt = ops.pvgo_retract(nodes, vels, D.view(N, 9).contiguous(), 1.0)
            vo_t, lin_t = _linearize(nt, vt, edges, poses, drots, dtrans, dvels, dts, dummy)
            Rt = _residual_vector(vo_t, lin_t)
            JD = J @ D
            s, q = torch.stack([(Rt * Rt).sum(), (JD * (2 * R + JD)).sum()]).tolist()
            if ctl.after_trial(s, q):
                nodes, vels = nt, vt
                break
        ctl.end_step()
    return nodes, vels, dict(steps=ctl.steps, trials=trials, loss=ctl.loss, trace=ctl.trace)
