"""Worker of tests/test_dist_rccl_gpu.py: one process per GPU under torch.distributed.run, REAL RCCL between them.  Every rank
runs the fused single-GPU LM loop (islam_pvgo_run_chain) and the sharded loop (islam_pvgo_run_chain_sharded on the library's own
RCCL communicator, islam_amd/csrc/pvgo_dist.hip) on the same problems -- a plain chain, two reject-heavy ones (cancelled
run-ahead chains whose collectives still run), a failed solve -- and compares.  Exit code 0 = every rank agreed."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LW = (1, 0.1, 10, 0.1)


def main():
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', device_id=dev)
    from islam_amd import dist_pvgo, ops
    from tests.helpers import chain_problem
    from tests.test_dist_c_gpu import _noisy, _problem
    comm = dist_pvgo.RcclComm(device=dev)
    assert comm.world == world and comm.rank == rank
    cases = [('chain300', _problem(300, dev), None), ('chain5001', _problem(5001, dev), None),
             ('noisy65a', _noisy(65, 8, 1.5, dev), None), ('noisy65b', _noisy(65, 2, 1.0, dev), None)]

    def bad_params():
        p = ops.pvgo_default_params(LW, radius=1e4)
        for i, w in enumerate((1.0, -0.5, 100.0, 0.01)):
            p.w[i] = w
        return p
    cases.append(('failed_solve', _problem(33, dev), bad_params))
    for name, args, mk in cases:
        nodes, vels = args[0].clone(), args[1].clone()
        ref, _ = ops.pvgo_run_chain(nodes, vels, *args[2:], mk() if mk else ops.pvgo_default_params(LW, radius=1e4), trace_cap=256)
        for rep in range(2):                               # twice: no stale state between calls
            n, v, rr, xb = dist_pvgo.run_chain_sharded(comm, *args, LW, params=mk() if mk else None)
            assert (rr.trials, rr.steps, rr.status) == (ref.trials, ref.steps, ref.status), (name, rank, rr.trials, ref.trials)
            assert abs(rr.loss - ref.loss) <= 1e-9 * abs(ref.loss), (name, rank)
            tol = 1e-8 if name.startswith('noisy') else 1e-9
            assert float((n - nodes).abs().max()) <= tol and float((v - vels).abs().max()) <= tol, (name, rank)
            assert xb > 0 or world == 1
        if rank == 0:
            print('rccl world %d: %s ok (trials %d, steps %d, status %d, %d bytes through the collectives)' %
                  (world, name, ref.trials, ref.steps, ref.status, xb), flush=True)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
