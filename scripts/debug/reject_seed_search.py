"""CPU search (oracle banded LM; no GPU) for a perturbation of the bench's N = 5001 start state that makes LM REJECT trials:
bench.reject_heavy_rate perturbs with torch.Generator(seed) exactly as below.  Prints the accept/reject pattern per (seed, sig)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from islam_amd import synthetic, lietensor as pp
from oracle import imu as oimu, pvgo as opvgo

N = int(os.environ.get('N', 5001))
tr = synthetic.car_trajectory(N)
F = N
pos, rot, vel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'], tr['gravity'], False)
dpos, drot, dvel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'], tr['gravity'], True)
prob = synthetic.pvgo_problem_from_deltas(tr, drot, dpos, dvel, pos, rot, vel)
RS = float(os.environ.get('ROT', 0.2))          # rotation noise = RS * sig rad per axis (bench: 0.2)
cands = [(int(a), float(b)) for a, b in (c.split(':') for c in sys.argv[1:])] or [(s, sig) for sig in (1.5, 3.0, 5.0) for s in range(1, 9)]
for seed, sig in cands:
    g = torch.Generator().manual_seed(seed)
    dt_ = torch.randn(N, 3, generator=g, dtype=torch.float64) * sig
    dr_ = torch.randn(N, 3, generator=g, dtype=torch.float64) * (RS * sig)
    n0 = torch.tensor(prob['init_nodes'])
    n0 = torch.cat([n0[:, :3] + dt_, n0[:, 3:]], 1)
    pert = pp.SE3(torch.cat([torch.zeros(N, 3, dtype=torch.float64), pp.so3(dr_).Exp().tensor()], 1))
    start = (pert @ pp.SE3(n0)).tensor().numpy()
    t0 = time.time()
    out = opvgo.run_pvgo(**dict(prob, init_nodes=start), loss_weight=(1, 0.1, 10, 0.1), mode='banded', return_optimizer=True)
    opt = out[5]
    pat = ''.join(str(int(not t[2])) for t in opt.trace)
    print('seed %d sig %.2f: trials %d rejects %d pattern %s  (%.1f s)' % (seed, sig, len(pat), pat.count('1'), pat, time.time() - t0), flush=True)
