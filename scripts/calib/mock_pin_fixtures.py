"""PLUMBING CHECK ONLY -- NOT A PIN.  Writes files in the format of tests/golden/make_pvgo_golden.py into a scratch directory,
filled from the repo's OWN oracle and LieTensor shim, so that the fixture consumers (tests/test_pypose_pin_{cpu,gpu}.py) can
be exercised end to end before real PyPose fixtures exist:

  python scripts/calib/mock_pin_fixtures.py /tmp/mockpin && ISLAM_PIN_DIR=/tmp/mockpin python -m pytest tests/test_pypose_pin_cpu.py

Agreement with these files says nothing about PyPose; never copy them into tests/golden/."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from islam_amd import lietensor as pp, synthetic, transformation as tf  # noqa: E402
from oracle import imu as oimu, pvgo as opvgo  # noqa: E402
from tests.golden.make_pvgo_golden import LW, noisy_problem  # noqa: E402
from tests.helpers import chain_problem  # noqa: E402

out_dir = sys.argv[1]
os.makedirs(out_dir, exist_ok=True)


def pv(prob, lw, target='vo'):
    p32 = {k: (np.asarray(v, np.float32).astype(np.float64) if k != 'links' else v) for k, v in prob.items()}
    tl, rl, nodes, vels, _, opt = opvgo.run_pvgo(**p32, loss_weight=lw, mode='dense', target=target, return_optimizer=True)
    o = dict(trans_loss=tl, rot_loss=rl, nodes=nodes.astype(np.float32), vels=vels.astype(np.float32), step_losses=np.array(opt.step_losses),
             loss_weight=np.array(lw, dtype=np.float64), target=np.array(target))
    if target == 'vo':
        o['vo_grad'] = opvgo.vo_loss_grad(opt.nodes, prob['links'], p32['vo_motions'], np.full(len(tl), 0.1), np.full(len(tl), 1.0))
    o.update({'in_' + k: np.asarray(v) for k, v in prob.items()})
    return o


cases = {'chain9': chain_problem(9)[0], 'chain65': chain_problem(65)[0], 'noisy33': noisy_problem(33, 1, 1.5),
         'noisy65a': noisy_problem(65, 8, 1.5), 'noisy65b': noisy_problem(65, 2, 1.0)}
for name, prob in cases.items():
    np.savez(os.path.join(out_dir, 'pvgo_%s.npz' % name), **pv(prob, LW))
np.savez(os.path.join(out_dir, 'pvgo_chain9_imu.npz'), **pv(cases['chain9'], LW, 'imu'))
np.savez(os.path.join(out_dir, 'pvgo_chain9_euroc.npz'), **pv(cases['chain9'], (4, 0.1, 2, 0.1)))

tr = synthetic.car_trajectory(13, seed=5)
sync = np.array([0, 0, 1, 11, 81, 91, 101, 101, 111], dtype=np.int64)
init = dict(pos=tr['gt_pos'][0], rot=tr['gt_quat'][0], vel=tr['gt_vel'][0])
for dt, tag in ((np.float32, 'f32'), (np.float64, 'f64')):
    o = dict(accels=tr['accels'], gyros=tr['gyros'], dts=tr['imu_dts'], sync=sync, gravity=np.array(tr['gravity']),
             init_pos=init['pos'], init_rot=init['rot'], init_vel=init['vel'])
    for motion, m in ((False, 'world'), (True, 'motion')):
        pos, rot, vel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], sync, 0, len(sync) - 1, init, tr['gravity'], motion, dtype=dt)
        o.update({m + '_pos': pos, m + '_rot': rot, m + '_vel': vel})
    np.savez(os.path.join(out_dir, 'imu_%s.npz' % tag), **o)

g = torch.Generator().manual_seed(3)
xi, yi = torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.5, torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.5
pts = torch.randn(6, 3, generator=g, dtype=torch.float64)
w6, w3 = torch.linspace(0.3, 1.1, 6, dtype=torch.float64), torch.tensor([0.7, -0.4, 1.3], dtype=torch.float64)
X0, Y0 = pp.se3(xi).Exp().tensor(), pp.se3(yi).Exp().tensor()
o = dict(xi=xi.numpy(), yi=yi.numpy(), pts=pts.numpy(), X=X0.numpy(), Y=Y0.numpy(), w6=w6.numpy(), w3=w3.numpy())


def grad_of(fn, *leaves):
    ls = [l.clone().requires_grad_(True) for l in leaves]
    fn(*ls).backward()
    return [l.grad.numpy() for l in ls]


o['val_mul'] = (pp.SE3(X0) @ pp.SE3(Y0)).tensor().numpy()
o['val_inv'] = pp.SE3(X0).Inv().tensor().numpy()
o['val_log'] = pp.SE3(X0).Log().tensor().numpy()
o['val_act'] = (pp.SE3(X0) @ pts).numpy()
o['val_rot_log'] = pp.SE3(X0).rotation().Log().tensor().numpy()
o['g_log'] = grad_of(lambda X: (pp.SE3(X).Log().tensor() * w6).sum(), X0)[0]
o['g_mul_left'], o['g_mul_right'] = grad_of(lambda X, Y: ((pp.SE3(X) @ pp.SE3(Y)).Log().tensor() * w6).sum(), X0, Y0)
o['g_inv'] = grad_of(lambda X: (pp.SE3(X).Inv().Log().tensor() * w6).sum(), X0)[0]
o['g_act'] = grad_of(lambda X: ((pp.SE3(X) @ pts) * w3).sum(), X0)[0]
o['g_exp'] = grad_of(lambda v: ((pp.se3(v).Exp() @ pp.SE3(Y0)).Log().tensor() * w6).sum(), xi)[0]
q0 = pp.so3(xi[:, 3:]).Exp().tensor()
o['q'] = q0.numpy()
o['g_so3_log'] = grad_of(lambda q: (pp.SO3(q).Log().tensor() * w3).sum(), q0)[0]
o['g_so3_exp'] = grad_of(lambda v: (pp.so3(v).Exp().Log().tensor() * w3).sum(), xi[:, 3:].contiguous())[0]
d7 = torch.cat([yi * 0.1, torch.ones(6, 1, dtype=torch.float64)], 1)
o['add_delta'], o['val_add'] = d7.numpy(), (pp.se3(d7[:, :6]).Exp() @ pp.SE3(X0)).tensor().numpy()
q32 = pp.so3((xi[:, 3:] * 0.3).float().repeat(3, 1)).Exp().tensor().numpy()
q, s = q32.copy(), 1
from oracle import lie  # noqa: E402
while s < len(q):
    prev = q.copy()
    q[s:] = lie.quat_mul(prev[:-s], prev[s:]).astype(np.float32)
    s *= 2
o['cumprod_in'], o['cumprod_out'] = q32, q
m6 = torch.randn(5, 6, generator=g, dtype=torch.float64) * 0.2
o['tf_in'] = m6.numpy()
o['tf_cvt'] = tf.cvtSE3_pypose(m6).tensor().numpy()
K = tf.tartan2kitti_pypose(m6)
o['tf_kitti'] = K.tensor().numpy()
P = tf.motion2pose_pypose(K, pp.SE3(X0[0]))
o['tf_poses'] = P.tensor().numpy()
o['tf_motions'] = tf.pose2motion_pypose(P).tensor().numpy()
np.savez(os.path.join(out_dir, 'lieops.npz'), **o)
print('mock fixtures (plumbing check only) in', out_dir)
