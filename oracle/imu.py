"""Oracle (test infrastructure): IMUModule.integrate driver on top of the C restatement.

Mirrors reference imu_integrator.py:69-164: slices the IMU stream for frames [st, end) with
rgb2imu_sync, applies the static bias when no denoiser is used (:101-105), then runs the frame
loop (in C, oracle/imu_preint.c)."""
import numpy as np

from . import cwrap


def integrate(accels, gyros, dts, rgb2imu_sync, st, end, init, gravity, motion_mode,
              accel_bias=None, gyro_bias=None, dtype=np.float64):
    sync = np.asarray(rgb2imu_sync, dtype=np.int64)
    b0, b1 = int(sync[st]), int(sync[end]) + 1
    dtype = np.dtype(dtype)
    d = np.asarray(dts, dtype=dtype)[b0:b1].copy()
    g = np.asarray(gyros, dtype=dtype)[b0:b1].copy()
    a = np.asarray(accels, dtype=dtype)[b0:b1].copy()
    if accel_bias is not None:
        a -= np.asarray(accel_bias, dtype=dtype).reshape(1, 3)
    if gyro_bias is not None:
        g -= np.asarray(gyro_bias, dtype=dtype).reshape(1, 3)
    seg = sync[st:end + 1] - b0
    return cwrap.imu_integrate(d, g, a, seg, init['pos'], init['rot'], init['vel'], gravity, motion_mode, dtype)
