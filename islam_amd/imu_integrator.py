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

    def integrate(self, st, end, init=None, motion_mode=False):
        """imu_integrator.py:69-164.  world mode: (end-st+1) rows incl. the initial state; motion mode: (end-st) rows."""
        init_pos, init_rot, init_vel = prase_init(init, motion_mode, self.device, self.dtype)
        b0 = int(self.rgb2imu_sync[st])
        b1 = int(self.rgb2imu_sync[end]) + 1
        dts = self.dts[b0:b1, 0].clone()
        gyros = self.gyros[b0:b1].clone()
        accels = self.accels[b0:b1].clone()
        if self.optm_bias:
            if self.denoise_accel:
                accels -= self.accel_bias.view(1, 3)
            if self.denoise_gyro:
                gyros -= self.gyro_bias.view(1, 3)
        if self.use_denoise_model and b1 - b0 >= 10:
            d_acc, d_gyro, _, _ = self.denoiser({'acc': accels.float(), 'gyro': gyros.float()}, eval=True)
            if self.denoise_accel:
                accels = d_acc.to(self.dtype)
            if self.denoise_gyro:
                gyros = d_gyro.to(self.dtype)
        seg_host = np.ascontiguousarray(self.rgb2imu_sync[st:end + 1] - b0, dtype=np.int64)
        seg = torch.from_numpy(seg_host).to(self.device)
        pos, rot, vel = ops.imu_preint(dts.contiguous(), gyros.contiguous(), accels.contiguous(), seg, seg_host,
                                       init_pos.contiguous(), init_rot.tensor().contiguous(), init_vel.contiguous(),
                                       self.gravity, motion_mode)
        return pos.cpu(), pp.SO3(rot.cpu()), [], vel.cpu()
