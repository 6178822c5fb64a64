"""Per-stage GPU time of ONE eager forward of the frozen stereo / flow execution copies (B=8, 448x640, bf16): torch.profiler kernel
events attributed to the module (or PWC level) whose forward launched them, plus the time-ordered kernel list of one forward.
    python scripts/frozen_timeline.py [stereo|flow|both] [outdir]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
outdir = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out'
os.makedirs(outdir, exist_ok=True)
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev).train()
vonet.set_frozen_dtype(torch.bfloat16, torch.bfloat16)
x = torch.randn(8, 6, 448, 640, device=dev)
REPS = 3


MARKS = []          # ('enter' | 'exit', name) in host order; each one also launches a marker kernel (torch.cuda._sleep -> spin_kernel),
                    # so the k-th marker kernel in device order IS the k-th entry: attribution relies on stream order only


def mark(kind, name):
    MARKS.append((kind, name))
    torch.cuda._sleep(1)


class rng:
    def __init__(self, name):
        self.name = name
    def __enter__(self):
        mark('enter', self.name)
    def __exit__(self, *a):
        mark('exit', self.name)


def hook_ranges(mods):
    for name, m in mods:
        m.register_forward_pre_hook(lambda mod, inp, name=name: mark('enter', name))
        m.register_forward_hook(lambda mod, inp, out, name=name: mark('exit', name))


def analyse(prof, tag, marks):
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    evs.sort(key=lambda e: e.time_range.start)
    per = collections.OrderedDict()
    seq, stack, mi = [], [], 0
    for e in evs:
        if 'spin_kernel' in e.name:
            kind, name = marks[mi]
            mi += 1
            if kind == 'enter':
                stack.append(name)
            else:
                assert stack and stack[-1] == name, (stack, name)
                stack.pop()
            continue
        name = stack[-1] if stack else '(outside)'
        d = per.setdefault(name, [0.0, 0])
        d[0] += e.device_time / 1e3 / REPS
        d[1] += 1
        seq.append((name, e.name, e.device_time))
    assert mi == len(marks), (mi, len(marks))
    tot = sum(v[0] for v in per.values())
    lines = ['%s: %.3f ms of kernels per forward, %d launches' % (tag, tot, sum(v[1] for v in per.values()) // REPS)]
    for k, v in per.items():
        lines.append('  %7.3f ms  n=%-4d %s' % (v[0], v[1] // REPS, k))
    print('\n'.join(lines))
    with open(os.path.join(outdir, 'timeline_%s.txt' % tag), 'w') as f:
        f.write('\n'.join(lines) + '\n\n# time-ordered kernels of the LAST forward: stage, us, kernel\n')
        n = len(seq) // REPS
        for name, kn, d in seq[-n:]:
            f.write('%-28s %8.1f  %s\n' % (name, d, kn[:110]))


def run_profile(fn, tag):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        del MARKS[:]
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(REPS):
                fn()
            torch.cuda.synchronize()
        marks = list(MARKS)
    analyse(prof, tag, marks)


if which in ('stereo', 'both'):
    vonet._run_frozen('stereo', vonet.stereoNet, torch.bfloat16, x, quarter=True)      # builds the execution copy
    ex = vonet._exec['stereo'].module()
    fe = ex.feature_extraction
    mods = [('fe.' + n, m) for n, m in fe.named_children()] + [(n, m) for n, m in ex.named_children() if n != 'feature_extraction']
    fine = os.environ.get('TIMELINE_FINE', '1') == '1'
    if fine:                                                  # every hourglass residual / PSM block gets its own line
        mods += [(n.replace('feature_extraction.', 'fe.'), m) for n, m in ex.named_modules()
                 if isinstance(m, (nets._HGResidual, nets._PSMBlock)) or (isinstance(m, nets.Hourglass) and '.' in n)]
    hook_ranges(mods)
    hook_ranges([('feature_extraction', fe)])
    run_profile(lambda: vonet._run_frozen('stereo', vonet.stereoNet, torch.bfloat16, x, quarter=True), 'stereo')

if which in ('flow', 'both'):
    run = lambda: vonet._run_frozen('flow', vonet.flowNet, torch.bfloat16, x)
    run()
    fx = vonet.flowNet
    # PWC levels are not modules: wrap the methods that make up a level
    def wrap(obj, meth, label):
        orig = getattr(obj, meth)
        def w(*a, **k):
            l = a[0] if a and isinstance(a[0], (int, str)) else ''
            with rng('%s%s' % (label, l)):
                return orig(*a, **k)
        setattr(obj, meth, w)
    wrap(fx, '_pyramid_level', 'pyr')
    wrap(fx, '_dense_run', 'dense')
    wrap(fx, '_head_up_mirror', 'head')
    wrap(fx, '_up2', 'up2_')
    orig_c = fx._c
    def c(name, *a, **k):
        with rng('c_' + name):
            return orig_c(name, *a, **k)
    fx._c = c
    from islam_amd import ops
    for fn in ('corr81_act', 'warp'):
        o = getattr(ops, fn)
        def w(*a, _o=o, _n=fn, **k):
            with rng(_n):
                return _o(*a, **k)
        setattr(ops, fn, w)
    nets.warp_fn = lambda xx, fl, sc: ops.warp(xx, fl, sc)
    run_profile(run, 'flow')
