"""Oracle (test infrastructure): ctypes bindings of oracle/liboracle.so (built by oracle/Makefile)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def sincos(x):
    s, c = ctypes.c_double(), ctypes.c_double()
    lib().islam_oracle_sincos(ctypes.c_double(x), ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value


def imu_integrate(dt, gyro, acc, seg, init_pos, init_rot, init_vel, gravity, motion_mode, dtype=np.float64):
    """IMUModule.integrate restated (imu_integrator.py:69-164).  seg: (nframes+1,) sample offsets.
    Returns (pos, rot, vel): nframes+1 rows in world mode (row 0 = init), nframes rows in motion mode."""
    dtype = np.dtype(dtype)
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=dtype))
    dt, gyro, acc = c(dt).reshape(-1), c(gyro).reshape(-1, 3), c(acc).reshape(-1, 3)
    seg = np.ascontiguousarray(np.asarray(seg, dtype=np.int64))
    nframes = len(seg) - 1
    rows = nframes if motion_mode else nframes + 1
    pos, rot, vel = np.zeros((rows, 3), dtype), np.zeros((rows, 4), dtype), np.zeros((rows, 3), dtype)
    ip, ir, iv = c(init_pos), c(init_rot), c(init_vel)
    if dtype == np.float64:
        fn, g = lib().islam_oracle_imu_integrate_f64, ctypes.c_double(gravity)
    else:
        fn, g = lib().islam_oracle_imu_integrate_f32, ctypes.c_float(gravity)
    fn(_p(dt), _p(gyro), _p(acc), _p(seg), ctypes.c_int(nframes), _p(ip), _p(ir), _p(iv), g,
       ctypes.c_int(1 if motion_mode else 0), _p(pos), _p(rot), _p(vel))
    return pos, rot, vel


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def corr81_fwd(f1, f2):
    f1, f2 = _f32(f1), _f32(f2)
    B, C, H, W = f1.shape
    out = np.zeros((B, 81, H, W), np.float32)
    lib().islam_oracle_corr81_fwd(_p(f1), _p(f2), _p(out), B, C, H, W)
    return out


def corr81_bwd(f1, f2, gout):
    f1, f2, gout = _f32(f1), _f32(f2), _f32(gout)
    B, C, H, W = f1.shape
    g1, g2 = np.zeros_like(f1), np.zeros_like(f2)
    lib().islam_oracle_corr81_bwd_first(_p(f2), _p(gout), _p(g1), B, C, H, W)
    lib().islam_oracle_corr81_bwd_second(_p(f1), _p(gout), _p(g2), B, C, H, W)
    return g1, g2


def warp(x, flo):
    x, flo = _f32(x), _f32(flo)
    B, C, H, W = x.shape
    out = np.zeros_like(x)
    lib().islam_oracle_warp(_p(x), _p(flo), _p(out), B, C, H, W)
    return out
