"""IMUModule with the reference's call surface (reference imu_integrator.py:11-164) on the HIP integrator.

One ``integrate`` call = slice the stream (:94-99), static-bias / denoiser correction (:101-113, PyTorch-ROCm),
then ONE call into libislam_hip.so for the whole frame loop (:116-158): the reference runs ~60 tiny kernels
and 3 device->host copies per frame.  Outputs come back on the CPU like the reference's (poses, rots, covs, vels).
"""
import numpy as np
import torch

from . import lietensor as pp
from . import ops
from .nets import IMUCorrector_CNN_GRU_WO_COV


def prase_init(init=None, motion_mode=False, device='cuda:0', dtype=None):
    """imu_integrator.py:11-28 (name kept, typo included)."""
    dtype = dtype or torch.get_default_dtype()
    z3 = torch.zeros(3, dtype=dtype, device=device)
    if init is not None:
        rot = torch.as_tensor(np.asarray(init['rot']), dtype=dtype).to(device)
        if motion_mode:
            return z3, pp.SO3(rot), z3.clone()
        return (torch.as_tensor(np.asarray(init['pos']), dtype=dtype).to(device), pp.SO3(rot),
                torch.as_tensor(np.asarray(init['vel']), dtype=dtype).to(device))
    return z3, pp.identity_SO3(dtype=dtype, device=device), z3.clone()


class IMUModule:
    def __init__(self, accels, gyros, dts, accel_bias=torch.zeros(3), gyro_bias=torch.zeros(3), init=None, gravity=9.81007,
                 rgb2imu_sync=None, device='cuda:0', denoise_model_name=None, denoise_accel=True, denoise_gyro=True,
                 use_est_cov=False, dtype=None):
        if torch.device(device).type != 'cuda':
            raise RuntimeError('islam_amd.IMUModule runs on the MI355X only (device=%r); there is no CPU fallback' % (device,))
        self.device = device
        self.last_frame_dt = 0.1
        self.dtype = dtype or torch.get_default_dtype()       # the reference integrates in the default dtype (:44)
        self.gravity = float(gravity)
        self.rgb2imu_sync = np.arange(len(accels), dtype=np.int64) if rgb2imu_sync is None else \
            np.asarray(rgb2imu_sync, dtype=np.int64)
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=self.dtype).to(device)
        self.accels, self.gyros = t(accels), t(gyros)
        self.dts = t(dts).unsqueeze(-1)
        self.denoise_accel, self.denoise_gyro = denoise_accel, denoise_gyro
        self.use_denoise_model = denoise_model_name is not None and denoise_model_name != '' and (denoise_accel or denoise_gyro)
        self.optm_bias = not self.use_denoise_model and (denoise_accel or denoise_gyro)
        self.accel_bias, self.gyro_bias = t(accel_bias), t(gyro_bias)
        if self.use_denoise_model:
            self.denoiser = IMUCorrector_CNN_GRU_WO_COV()
            self.denoiser.load_state_dict(torch.load(denoise_model_name))
            self.denoiser = self.denoiser.to(device)
            self.use_est_cov = use_est_cov
        # The reference calls the denoiser with eval=True (imu_integrator.py:109): no gradient ever reaches it, so its
        # "IMU epochs" (train.py:177-179) step the optimizer on empty gradients.  train_denoiser = True is the
        # fix SURVEY section 8f rank 4 asks for: the denoiser runs with grad enabled and the pre-integration is differentiable
        # (islam_imu_preint_bwd), so run_pvgo(target='imu') back-propagates into its parameters.  Default: reference behaviour.
        self.train_denoiser = False

    def _corrected(self, st, end):
        """The stream slice of frames [st, end] after the static-bias / denoiser correction (imu_integrator.py:94-113)."""
        b0 = int(self.rgb2imu_sync[st])
        b1 = int(self.rgb2imu_sync[end]) + 1
        dts = self.dts[b0:b1, 0]                    # contiguous views of the stream; only written to through new tensors
        gyros = self.gyros[b0:b1]
        accels = self.accels[b0:b1]
        if self.optm_bias:
            if self.denoise_accel:
                accels = accels - self.accel_bias.view(1, 3)
            if self.denoise_gyro:
                gyros = gyros - self.gyro_bias.view(1, 3)
        if self.use_denoise_model and b1 - b0 >= 10:
            ddt = next(self.denoiser.parameters()).dtype          # the reference's denoiser lives in the default dtype too
            d_acc, d_gyro, _, _ = self.denoiser({'acc': accels.to(ddt), 'gyro': gyros.to(ddt)}, eval=not self.train_denoiser)
            if self.denoise_accel:
                accels = d_acc.to(self.dtype)
            if self.denoise_gyro:
                gyros = d_gyro.to(self.dtype)
        return b0, dts, gyros, accels

    def integrate_both(self, st, end, init=None):
        """``integrate(st, end, init, motion_mode=False)`` and ``integrate(st, end, init, motion_mode=True)`` -- the pair the
        reference's loop asks for on every batch (train.py:200-215) -- from ONE pass over the samples (islam_imu_preint_both: the
        scan, the rotation chain and the frame sums are common to the two modes) and one device->host copy.  Returns the two
        result tuples, bit-identical to the two calls.  Forward values only: with ``train_denoiser`` the two differentiable calls
        are made instead."""
        if self.train_denoiser and self.use_denoise_model:
            return self.integrate(st, end, init, motion_mode=False), self.integrate(st, end, init, motion_mode=True)
        b0, dts, gyros, accels = self._corrected(st, end)
        np_dt = {torch.float32: np.float32, torch.float64: np.float64}[self.dtype]
        i10 = np.zeros(10, dtype=np_dt)
        i10[6] = 1.0
        if init is not None:                        # prase_init: the motion rows ignore pos / vel (the kernel starts them from zero)
            i10[3:7] = np.asarray(init['rot'], dtype=np_dt)
            i10[0:3] = np.asarray(init['pos'], dtype=np_dt)
            i10[7:10] = np.asarray(init['vel'], dtype=np_dt)
        i10 = torch.from_numpy(i10).to(self.device)
        seg_host = np.ascontiguousarray(self.rgb2imu_sync[st:end + 1] - b0, dtype=np.int64)
        seg = torch.from_numpy(seg_host).to(self.device)
        with torch.no_grad():
            world, motion, packed = ops.imu_preint_both(dts.contiguous(), gyros.detach().contiguous(), accels.detach().contiguous(), seg,
                                                        seg_host, i10[0:3], i10[3:7], i10[7:10], self.gravity)
        host = packed.cpu()
        n = len(seg_host) - 1
        res, o = [], 0
        for rows in (n + 1, n):
            pos = host[o:o + rows * 3].view(rows, 3); o += rows * 3
            rot = host[o:o + rows * 4].view(rows, 4); o += rows * 4
            vel = host[o:o + rows * 3].view(rows, 3); o += rows * 3
            res.append((pos.contiguous(), pp.SO3(rot.contiguous()), [], vel.contiguous()))
        return res[0], res[1]

    def integrate(self, st, end, init=None, motion_mode=False):
        """imu_integrator.py:69-164.  world mode: (end-st+1) rows incl. the initial state; motion mode: (end-st) rows.
        Host traffic per call: one H2D of the 10 initial-state values, one of the frame offsets, one D2H of the packed
        result (the reference: 3 D2H copies per frame)."""
        b0, dts, gyros, accels = self._corrected(st, end)
        # prase_init (imu_integrator.py:11-28) packed into one transfer: [pos(3) | rot(4) | vel(3)]
        np_dt = {torch.float32: np.float32, torch.float64: np.float64}[self.dtype]
        i10 = np.zeros(10, dtype=np_dt)
        i10[6] = 1.0
        if init is not None:
            i10[3:7] = np.asarray(init['rot'], dtype=np_dt)
            if not motion_mode:
                i10[0:3] = np.asarray(init['pos'], dtype=np_dt)
                i10[7:10] = np.asarray(init['vel'], dtype=np_dt)
        i10 = torch.from_numpy(i10).to(self.device)
        seg_host = np.ascontiguousarray(self.rgb2imu_sync[st:end + 1] - b0, dtype=np.int64)
        seg = torch.from_numpy(seg_host).to(self.device)
        pos, rot, vel = ops.imu_preint(dts.contiguous(), gyros.contiguous(), accels.contiguous(), seg, seg_host,
                                       i10[0:3], i10[3:7], i10[7:10], self.gravity, motion_mode)
        out = torch.cat((pos, rot, vel), 1).cpu()
        return out[:, 0:3].contiguous(), pp.SO3(out[:, 3:7].contiguous()), [], out[:, 7:10].contiguous()
