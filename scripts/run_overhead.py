"""Where the per-run_pvgo time outside the kernels goes: the Python wrapper against a bare ctypes call with prebuilt arguments."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from islam_amd import ops, _lib
from islam_amd._lib import c_size_t, c_void_p, lib, ptr, stream_ptr
dev = torch.device('cuda:0')
prob, tr = bench.build_problem(dev, 5001)
prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
ws = ops.pvgo_workspace(5001, dev)
R = 220
states = [(prob['init_nodes'].clone(), prob['init_vels'].clone()) for _ in range(2 * R)]


def timed(fn, states):
    for s in states[:20]:
        fn(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for s in states[20:]:
        n += fn(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def wrapper(s):
    res, _ = ops.pvgo_run_chain(s[0], s[1], prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
    return res.trials


raw = lib()._cdll.islam_pvgo_run_chain_reproj
res = _lib.PvgoResult()
fixed = [ptr(prob[k]) for k in ('vo', 'drots', 'dtrans', 'dvels', 'dts')]
sp = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
wsp, wsb = ptr(ws[0]), c_size_t(ws[1])
pr, rr = ctypes.byref(prm), ctypes.byref(res)


def bare(s):
    rc = raw(ptr(s[0]), ptr(s[1]), *fixed, 5001, pr, None, wsp, wsb, rr, c_void_p(0), 0, sp)
    assert rc == 0
    return res.trials


print('wrapper : %.2f us per LM iteration' % timed(wrapper, states[:R]))
print('bare    : %.2f us per LM iteration' % timed(bare, states[R:]))
print('wrapper : %.2f us per LM iteration' % timed(wrapper, [(a.copy_(prob['init_nodes']), b.copy_(prob['init_vels'])) and (a, b) for a, b in states[:R]]))
