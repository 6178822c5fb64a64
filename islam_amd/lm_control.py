"""Scalar control logic of the LM loop on the host, for the sharded (multi-GPU) path where every trial already
ends in a collective: pp.optim.LM's accept/reject rule, ppost.TrustRegion.update and StopOnPlateau as the reference
constructs them (pvgo.py:169-180; SURVEY.md section 8a box).  The single-GPU path runs the same logic on the device
(control_*_kernel in islam_amd/csrc/pvgo.hip)."""
import numpy as np


class LMControl:
    def __init__(self, radius=1e4, high=0.5, low=1e-3, up=2.0, down=0.5, factor=0.5, rmin=1e-6, rmax=1e16, reject=16,
                 max_steps=10, patience=3, decreasing=1e-3):
        self.radius, self.damping = radius, 1.0 / radius
        self.high, self.low, self.up, self.down0, self.factor, self.rmin, self.rmax = high, low, up, down, factor, rmin, rmax
        self.down = down
        self.reject, self.max_steps, self.patience, self.decreasing = reject, max_steps, patience, decreasing
        self.has_loss, self.loss, self.last = False, None, None
        self.reject_count, self.steps, self.patience_count = 0, 0, 0
        self.continual = True
        self.trace = []

    def set_initial_loss(self, loss):
        self.loss, self.has_loss = loss, True

    def begin_step(self):
        self.last = self.loss                     # self.last = self.loss
        self.reject_count = 0

    def after_trial(self, loss_trial, qsum):
        """qsum = sum JD.(2R + JD) (unweighted).  Returns True when the step is kept (loop breaks), False on reject."""
        # plain IEEE division like PyPose, the device code (lm_control in csrc/pvgo.hip) and the C loop (csrc/pvgo_dist.hip):
        # x/0 = +-inf, 0/0 = NaN -- and NaN fails both comparisons below, i.e. takes the shrink-the-radius branch
        with np.errstate(divide='ignore', invalid='ignore'):
            quality = float(np.float64(self.last - loss_trial) / np.float64(-qsum))
        radius = 1.0 / self.damping
        if quality > self.high:
            radius, self.down = self.up * radius, self.down0
        elif quality > self.low:
            self.down = self.down0
        else:
            radius, self.down = radius * self.down, self.down * self.factor
        self.down = max(self.rmin, min(self.down, self.rmax))
        radius = max(self.rmin, min(radius, self.rmax))
        self.radius, self.damping = radius, 1.0 / radius
        if self.last < loss_trial and self.reject_count < self.reject:
            self.trace.append((loss_trial, self.damping, False))
            self.loss = self.last
            self.reject_count += 1
            return False
        self.trace.append((loss_trial, self.damping, True))
        self.loss = loss_trial
        return True

    def solver_failed(self):
        """PyPose: 'Linear solver failed. Breaking optimization step...' -- parameters and loss unchanged."""
        self.trace.append((float('nan'), self.damping, False))

    def end_step(self):
        """scheduler.step(loss): StopOnPlateau."""
        self.steps += 1
        if self.steps >= self.max_steps:
            self.continual = False
        if (self.last - self.loss) < self.decreasing:
            self.patience_count += 1
        else:
            self.patience_count = 0
        if self.patience_count >= self.patience:
            self.continual = False
        if self.reject_count >= self.reject:
            self.continual = False
