"""General-topology PVGO (chain + loop closures): band + PCG path vs the dense path, wall time of one run_pvgo."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from islam_amd import lietensor as pp
from islam_amd.pvgo import run_pvgo
from oracle import lie
from tests.helpers import chain_problem
dev = 'cuda'
for F, nclose, dense in ((3000, 6, True), (5001, 6, True), (5001, 24, False), (50001, 12, False)):
    prob, tr = chain_problem(F)
    links, vo = prob['links'].copy(), prob['vo_motions'].copy()
    gt = np.concatenate([tr['gt_pos'], tr['gt_quat']], 1)
    rng = np.random.default_rng(1)
    for e in rng.choice(F - 1, nclose, replace=False):
        i, j = rng.choice(F, 2, replace=False)
        if abs(int(i) - int(j)) < 2:
            j = (i + F // 2) % F
        links[e] = (i, j)
        vo[e] = lie.se3_mul(lie.se3_mul(lie.se3_inv(gt[i]), gt[j]), lie.se3_exp(rng.normal(0, 0.01, 6)))
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    args = (pp.SE3(t(prob['init_nodes'])), t(prob['init_vels']), pp.SE3(t(vo).to(dev)), torch.tensor(links), t(prob['dts']),
            pp.SO3(t(prob['imu_drots'])), t(prob['imu_dtrans']), t(prob['imu_dvels']))
    for how in (('band_pcg', 'dense') if dense else ('band_pcg',)):
        run_pvgo(*args, device=dev, loss_weight=(1, 0.1, 10, 0.1), general_solver=how, return_info=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = run_pvgo(*args, device=dev, loss_weight=(1, 0.1, 10, 0.1), general_solver=how, return_info=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        info = out[5]
        print('N=%d, %d loop closures, %s: %.3f s per run_pvgo, %d LM trials, %s PCG iterations, final loss %.6g' % (
            F, nclose, how, dt, info['trials'], info.get('pcg_iterations', '-'), info['loss']), flush=True)
