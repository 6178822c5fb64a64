"""Shared builders of seeded test problems (tests only; may import the oracle)."""
import numpy as np

from islam_amd import synthetic
from oracle import imu as oimu


def chain_problem(n_frames, seed=11, **kw):
    """Synthetic car trajectory -> run_pvgo inputs, IMU quantities from the oracle integrator."""
    tr = synthetic.car_trajectory(n_frames, seed=seed, **kw)
    F = n_frames
    pos, rot, vel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'],
                                   tr['gravity'], False)
    dpos, drot, dvel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'],
                                      tr['gravity'], True)
    return synthetic.pvgo_problem_from_deltas(tr, drot, dpos, dvel, pos, rot, vel), tr


def se3_log_err(X, Xref):
    from oracle import lie
    d = lie.se3_log(lie.se3_mul(lie.se3_inv(Xref), X))
    return np.linalg.norm(d, axis=-1)
