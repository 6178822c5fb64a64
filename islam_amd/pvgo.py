"""run_pvgo with the reference's call surface (reference pvgo.py:122-205) on the HIP back-end.

What runs where:
  * the LM loop (pp.optim.LM + Cholesky + TrustRegion + StopOnPlateau, pvgo.py:168-180) is ONE call into
    libislam_hip.so (islam_pvgo_run_chain): fp64, block-tridiagonal, partitioned block Cholesky;
  * vo_loss (pvgo.py:67-78) and align_to (:114-119) are HIP kernels; vo_loss is differentiable w.r.t. the VO
    motions with PyPose's gradient convention, so ``loss_bp.backward`` in train.py:280-283 works unchanged;
  * imu_loss (:95-111) is O(N) glue on the LieTensor shim (the IMU epoch carries no gradient in the
    reference release).
The information matrices of pvgo.py:125-143 are scalar multiples of the identity; they enter as four scalars (five with
the optional sparse reprojection factor ``reproj``, islam_amd/dense_ba.py).
"""
import numpy as np
import torch

from . import lietensor as pp
from . import ops


class UnsupportedGraphError(NotImplementedError):
    pass


def _is_canonical_chain(links, n_nodes):
    l = links.detach().cpu().numpy() if isinstance(links, torch.Tensor) else np.asarray(links)
    return l.shape == (n_nodes - 1, 2) and np.array_equal(l[:, 0], np.arange(n_nodes - 1)) and \
        np.array_equal(l[:, 1], np.arange(1, n_nodes))


def run_pvgo(init_nodes, init_vels, vo_motions, links, dts, imu_drots, imu_dtrans, imu_dvels,
             device='cuda:0', radius=1e4, loss_weight=(1, 1, 1, 1), reproj=None, target='vo', seg_len=(0, 0),
             return_info=False, general_solver='auto'):
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError("islam_amd.run_pvgo runs on the MI355X only (device=%r); there is no CPU fallback" % (device,))
    N = len(init_nodes)
    chain = _is_canonical_chain(links, N)
    rp = None
    if reproj is not None:               # 5th residual (pvgo.py:53-61,130-143): SparseReprojectionLoss-like object
        w5 = (loss_weight[4] / reproj.N) ** 2                                                  # pvgo.py:131
        K = reproj.K.detach().cpu().double()
        rp = ops.pvgo_reproj_struct(reproj.point3d.detach().to(dev, torch.float64).contiguous(),
                                    reproj.target.detach().to(dev, torch.float64).contiguous(),
                                    (K[0, 0], K[1, 1], K[0, 2], K[1, 2]),
                                    pp._plain(reproj.rgb2imu_pose).detach().cpu().double().reshape(7).tolist(), w5,
                                    getattr(reproj, 'compat_first_motion', True))
    out_dtype = pp._plain(init_nodes).dtype if isinstance(init_nodes, torch.Tensor) else torch.get_default_dtype()
    t64 = lambda x: pp._plain(torch.as_tensor(x)).detach().to(dev, torch.float64).contiguous()
    nodes, vels = t64(init_nodes).clone(), t64(init_vels).clone()
    poses, drots, dtrans, dvels = t64(vo_motions), t64(imu_drots), t64(imu_dtrans), t64(imu_dvels)
    dts64 = t64(dts).reshape(-1)
    edges = torch.as_tensor(links).to(dev, torch.int64).contiguous()
    target0 = nodes[0].clone()

    if chain:            # the topology train.py produces: block-tridiagonal fast path, whole LM loop in one library call
        prm = ops.pvgo_default_params(loss_weight, radius=radius, seg_len=seg_len)
        res, _ = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts64, prm, reproj=rp)
    else:                # loop closures / arbitrary links: dense formulation on the device (islam_amd/pvgo_dense.py)
        from .pvgo_dense import off_band_edges, run_lm_band_pcg, run_lm_dense
        k_off = len(off_band_edges(np.asarray(edges.cpu())))
        how = general_solver
        if how == 'auto':        # a long chain with a few loop closures: block-tridiagonal solver + low-rank correction (PCG)
            how = 'band_pcg' if (N > 512 and k_off <= 64) else 'dense'
        if how == 'band_pcg':
            nodes, vels, res = run_lm_band_pcg(nodes, vels, edges, poses, drots, dtrans, dvels, dts64, loss_weight, radius=radius, reproj=rp)
        elif how == 'dense':
            if N > 12000:
                raise UnsupportedGraphError('dense general-topology path is sized for N <= 12000 nodes, (9N)^2 doubles (got %d '
                                            'nodes, %d off-band edges; general_solver="band_pcg" has no such limit)' % (N, k_off))
            nodes, vels, res = run_lm_dense(nodes, vels, edges, poses, drots, dtrans, dvels, dts64, loss_weight, radius=radius, reproj=rp)
        else:
            raise ValueError("general_solver must be 'auto', 'dense' or 'band_pcg'")

    if target == 'vo':
        vo = vo_motions if isinstance(vo_motions, torch.Tensor) else torch.as_tensor(vo_motions)
        vo = pp._plain(vo).to(dev)
        trans_loss, rot_loss = ops.pvgo_vo_loss(nodes, edges, vo)
    elif target == 'imu':
        n = pp.SE3(nodes.to(out_dtype))
        v = vels.to(out_dtype)
        dr = imu_drots.to(dev) if isinstance(imu_drots, pp.LieTensor) else pp.SO3(torch.as_tensor(imu_drots).to(dev))
        dv = pp._plain(torch.as_tensor(imu_dvels)).to(dev)
        adj = dv - torch.diff(v, dim=0)
        err = (dr.Inv() @ n.rotation()[:-1].Inv() @ n.rotation()[1:]).Log().tensor()
        trans_loss, rot_loss = torch.sum(adj ** 2, dim=1), torch.sum(err ** 2, dim=1)
    else:
        raise ValueError("target must be 'vo' or 'imu'")

    an, av = ops.pvgo_align(nodes, vels, target0)
    # ONE device -> host copy for everything the caller reads on the host: aligned poses, velocities and the two loss vectors (their
    # host values ride along as ``.host`` on the returned loss tensors, so that a caller that only wants to report the loss does not
    # pay another synchronising read: BilevelLoop.step)
    E = trans_loss.shape[0]
    host = torch.cat([an.reshape(-1), av.reshape(-1), trans_loss.detach().to(torch.float64).reshape(-1),
                      rot_loss.detach().to(torch.float64).reshape(-1)]).cpu()
    nodes_out = pp.SE3(host[:7 * N].view(N, 7).to(out_dtype))
    vels_out = host[7 * N:10 * N].view(N, 3).to(out_dtype)
    trans_loss.host, rot_loss.host = host[10 * N:10 * N + E], host[10 * N + E:10 * N + 2 * E]
    n1 = N - 1
    covs = {'vo_rot': np.ones(len(links)) * loss_weight[0] ** 2, 'imu_rot': np.ones(n1) * loss_weight[2] ** 2,
            'vo_trans': np.ones(len(links)) * loss_weight[0] ** 2, 'imu_vel': np.ones(n1) * loss_weight[1] ** 2,
            'transvel': np.ones(n1) * loss_weight[3] ** 2}
    if reproj is not None:
        covs['reproj'] = np.ones(n1) * (loss_weight[4] / reproj.N) ** 2                        # pvgo.py:202-203
    if return_info:
        return trans_loss, rot_loss, nodes_out, vels_out, covs, res
    return trans_loss, rot_loss, nodes_out, vels_out, covs
