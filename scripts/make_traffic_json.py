"""profiles/traffic_r<NN>.json from a scripts/gpu_profile.sh run: HBM bytes per launch of the LM loop's kernels from the two rocprofv3
PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-B requests at 64 B on
gfx950 -> doubled; WRITE_SIZE as reported; KB = 1024 B), live dispatches only (gated no-op dispatches dropped), stamped with the
hash of the kernel source they were taken on (bench.py refuses the figures for any other source).
    python scripts/make_traffic_json.py gpurun_out/prof_<tag> [out.json]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'profiles', 'traffic_r05.json')
S = json.load(open(os.path.join(src, 'summary.json')))
sys.path.insert(0, ROOT)
import bench
sha = bench.pvgo_source_sha16()          # pvgo.hip + the pvgo_*.inl parts it includes


def pick(table, needle, grid=None):
    best = None
    for k, v in table.items():
        if needle in k and (grid is None or k.endswith('grid=%s' % grid)):
            if best is None or v['live_avg_kb'] > best[1]['live_avg_kb']:
                best = (k, v)
    return best


def dur(needle):
    best = None
    for k, v in S.get('per_grid', {}).items():
        if needle in k and (best is None or v['live_avg_ns'] > best[1]['live_avg_ns']):
            best = (k, v)
    return best


kernels = {}
for name, needle in (('trial_elim_kernel', 'trial_elim_kernel'), ('bt_eliminate_tw_kernel_L0', 'bt_eliminate_tw_kernel<1>'),
                     ('bt_downsweep_kernel', 'bt_downsweep_kernel')):
    f, w = pick(S.get('FETCH_SIZE', {}), needle), pick(S.get('WRITE_SIZE', {}), needle)
    d = dur(needle)
    if not f or not w:
        continue
    kernels[name] = {'FETCH_SIZE_kb_per_dispatch': f[1]['live_avg_kb'], 'WRITE_SIZE_kb_per_dispatch': w[1]['live_avg_kb'],
                     'traffic_bytes_per_launch': (2.0 * f[1]['live_avg_kb'] + w[1]['live_avg_kb']) * 1024.0,
                     'rocprof_live_avg_duration_ns': d[1]['live_avg_ns'] if d else None, 'dispatch': f[0]}
json.dump({'pvgo_hip_sha16': sha, 'kernels': kernels,
           'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-frontend '
                      '(separate passes, scripts/gpu_profile.sh)',
           'correction': 'FETCH_SIZE doubled (128-B requests counted at 64 B on gfx950), WRITE_SIZE as reported, KB = 1024 B; live dispatches only',
           'source': src}, open(out, 'w'), indent=1)
print(json.dumps(kernels, indent=1))
