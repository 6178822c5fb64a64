"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol
include/islam_hip.h declares (no compute calls: there is no GPU here), host-only entry points behave, and the
repository obeys the oracle / product separation rule."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from islam_amd import _lib
    return _lib.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'islam_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(islam_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from islam_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), 'libislam_hip.so does not export %s' % s
        assert s in _lib.SIGNATURES, 'islam_amd/_lib.py has no ctypes signature for %s' % s
    assert sorted(_lib.SIGNATURES) == syms


def test_shared_object_contains_gfx950_code():
    from islam_amd import _lib
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--list', '--type=o', '--input=' + _lib.LIB_PATH],
                         capture_output=True, text=True)
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob, 'no gfx950 code object embedded'
    assert b'gfx942' not in blob and b'sm_' not in blob       # single target, no compatibility paths


def test_host_only_entry_points(lib):
    from islam_amd import _lib
    assert lib.islam_abi_version() == 1
    p = _lib.PvgoParams()
    lib.islam_pvgo_default_params(ctypes.byref(p))
    assert (p.radius, p.vmin, p.vmax, p.reject, p.max_steps, p.patience, p.decreasing) == (1e4, 1e-4, 1e32, 16, 10, 3, 1e-3)
    assert (p.high, p.low, p.up, p.down, p.factor, p.rmin, p.rmax) == (0.5, 1e-3, 2.0, 0.5, 0.5, 1e-6, 1e16)
    assert lib.islam_pvgo_workspace_bytes(0) == 0
    b1, b2 = lib.islam_pvgo_workspace_bytes(9), lib.islam_pvgo_workspace_bytes(5001)
    assert 0 < b1 < b2 < 64 << 20
    assert lib.islam_imu_scratch_bytes(50001, 5000, 1) >= 8 * (4 * 55001 + 4 * 5001 + 7 * 5000)
    # argument validation happens before any device work
    rc = lib.islam_corr81_fwd(None, None, None, 0, 1, 1, 1, None, None)
    assert rc == -1 and b'bad shape' in lib.islam_last_error()
    rc = lib.islam_pvgo_linearize(None, None, None, None, None, None, None, 1, None, None, None)
    assert rc == -1
    # round-2 entry points: shapes and channel slices are checked on the host too
    assert lib.islam_conv_nhwc_flow(None, 568, 4, 120, None, None, None, 565, 0, None, 0, 0, 1, 8, 8, 128, 1, 0.1, None) == -1   # no output / xoff % 8
    assert b'islam_conv_nhwc_flow' in lib.islam_last_error()
    assert lib.islam_nchw_f32_to_nhwc_bf16(None, 10, 8, None, 16, 0, 1, 4, 2, 2, None) == -1                                     # 8 + 4 > 10 channels
    assert b'source slice' in lib.islam_last_error()
    assert lib.islam_maxpool2_nhwc_bf16(None, None, 1, 12, 4, 4, 0, None) == -1 and lib.islam_avgpool_nhwc_bf16(None, None, 1, 8, 4, 4, 8, None) == -1
    assert lib.islam_pvgo_run_chain_sharded(None, 2, 0, None, None, None, None, None, None, None, 100, None, None, None, 0, None, 0, None, None, None) == -1


def test_level_plan_matches_python_mirror(lib):
    from tests.np_shard_backend import plan_levels
    for N in (1, 2, 9, 40, 41, 64, 100, 257, 1000, 5001, 20000):
        for seg in ((0, 0), (4, 4), (19, 15), (7, 0), (2, 3)):
            sl = (ctypes.c_int * 2)(*seg)
            out = (ctypes.c_int * 19)()
            nl = lib.islam_pvgo_plan(N, sl, out)
            got = [(out[3 * l], out[3 * l + 1], out[3 * l + 2]) for l in range(nl)]
            ref, top = plan_levels(N, seg, with_top=True, twisted=True)
            assert got == ref and out[18] == top, (N, seg)
            n = N
            for (nn, m, P) in got[:-1]:
                assert nn == n and m >= 4 and P == -(-n // (m + 1))
                n = n // (m + 1)
            assert got[-1] == (n, n, 1) and len(got) <= 6


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'islam_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M) or 'liboracle' in text or '#include "../../oracle' in text:
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    uses = [m.start() for m in re.finditer(r'from oracle', bench)]
    a, b = bench.index('def cpu_baseline'), bench.index('def main')
    assert uses and all(a < u < b for u in uses)
    entry = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    assert entry.index('def smoke') < entry.index('from oracle')
    assert 'parity unpinned' in open(os.path.join(ROOT, 'oracle', '__init__.py')).read()


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from islam_amd import ops
    from islam_amd.imu_integrator import IMUModule
    from islam_amd.pvgo import run_pvgo
    with pytest.raises(RuntimeError):
        ops.corr81_forward(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4))
    with pytest.raises(RuntimeError):
        run_pvgo(torch.zeros(3, 7), torch.zeros(3, 3), torch.zeros(2, 7), torch.tensor([[0, 1], [1, 2]]), torch.zeros(2),
                 torch.zeros(2, 4), torch.zeros(2, 3), torch.zeros(2, 3), device='cpu')
    with pytest.raises(RuntimeError):
        IMUModule(torch.zeros(5, 3), torch.zeros(5, 3), torch.zeros(5), device='cpu')


@pytest.mark.parametrize('name', ['bench_r01_final.json', 'bench_r02_final.json', 'bench_r03_final.json', 'bench_r04_final.json', 'bench_r05_final.json'])
def test_committed_bench_line_follows_the_contract(name):
    """profiles/bench_rNN_final.json is the last `python bench.py` line of a round measured on the MI355X: one JSON object with
    the driver's keys, BASELINE.json's metric, a roofline object for the dominant kernel and a CPU baseline; from round 2 on
    also the whole-iteration and per-launch roofline entries."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = open(os.path.join(root, 'profiles', name)).read().strip().split('\n')[-1]
    d = json.loads(line)
    if 'r02' in name:
        it = d['roofline']['iteration']
        assert abs(it['frac'] - it['bytes'] / (it['us'] * 1e-6) / 1e9 / d['roofline']['peak']) < 1e-9
        assert abs(it['us'] - d['us_per_lm_iter']) < 1e-9 and it['bytes'] == 5000 * 5001
        pl = d['roofline']['per_launch']
        assert {'bt_eliminate_tw_kernel_L0', 'bt_downsweep_kernel', 'trial_lin_kernel'} <= set(pl)
        assert sum(v['us'] for v in pl.values()) < it['us']               # the three biggest launches fit inside the iteration
        assert 0 < d['stereo_vio']['mfma_frac'] < 1
    if 'r03' in name:                 # round 3: the dominant launch is the fused trial + elimination kernel; traffic carries its provenance
        it = d['roofline']['iteration']
        assert abs(it['us'] - d['us_per_lm_iter']) < 1e-9 and it['bytes'] == 5000 * 5001
        pl = d['roofline']['per_launch']
        assert {'bt_eliminate_tw_kernel_L0', 'bt_downsweep_kernel', 'trial_elim_kernel'} <= set(pl)
        assert 'trial_elim_kernel' in d['roofline']['kernel'] and d['roofline']['algorithmic_bytes_per_launch'] == pl['trial_elim_kernel']['bytes']
        assert d['roofline']['traffic'] is None or 'pvgo.hip sha256' in d['roofline']['traffic_source']
        dense = d['cpu_baseline']['dense_pypose_style']
        assert {'f32', 'f64'} <= set(dense) and dense['cores'] >= 1 and dense['f32']['fit_exponent'] > 1
    if 'r04' in name:                 # round 4: front-end roofline with executed matrix-core work, correlation / warp per launch, reject-heavy and large graph
        it = d['roofline']['iteration']
        assert abs(it['us'] - d['us_per_lm_iter']) < 1e-9 and it['bytes'] == 5000 * 5001
        assert {'bt_eliminate_tw_kernel_L0', 'bt_downsweep_kernel', 'trial_elim_kernel'} <= set(d['roofline']['per_launch'])
        assert d['roofline']['traffic'] is None or 'pvgo.hip sha256' in d['roofline']['traffic_source']
        fr = d['stereo_vio']['roofline']
        assert fr['bound'] == 'mfma' and abs(fr['frac'] - fr['achieved'] / fr['peak']) < 1e-9 and 0 < fr['frac'] < fr['frac_at_clock'] < 1
        assert abs(fr['achieved'] - fr['executed_gflop_per_frame'] * 1e-3 * d['stereo_vio']['value']) < 1e-6 * fr['achieved']
        for k in ('corr81_fwd_level2', 'warp_mask_level2'):
            e = d['front_end_per_launch'][k]
            assert abs(e['frac'] - e['bytes'] / (e['us'] * 1e-6) / 1e9 / 8000.0) < 1e-9 and 0 < e['frac'] < 1
        assert d['reject_heavy']['value'] > 0 and d['reject_heavy']['damping_changes_per_run'] + d['reject_heavy']['rejected_trials_per_run'] > 0
        assert d['large_graph']['N'] == 300007 and d['large_graph']['scaling'] == 'strong' and d['large_graph']['value'] > 0
        assert 'MIOpen stride-2 flow' not in d['stereo_vio']['nets'] and 'islam_hg_residual_nhwc_bf16' in d['stereo_vio']['nets']
    if 'r05' in name:                 # round 5: median-of-passes headline, counter traffic valid for the split source, self-explaining stereo_vio
        tp = d['timed_passes']
        assert tp['passes'] >= 5 and tp['passes'] * d['steps'] * d['ms_per_step'] >= 100.0 and tp['pass_ms']['min'] <= tp['pass_ms']['median'] <= tp['pass_ms']['max']
        assert abs(tp['pass_ms']['median'] - d['steps'] * d['ms_per_step']) < 1e-6
        assert d['roofline']['traffic'] is not None and 'pvgo.hip sha256' in d['roofline']['traffic_source']
        assert 0.9 < d['roofline']['traffic'] / d['roofline']['algorithmic_bytes_per_launch'] < 1.1
        assert 'extrapolated_iters_per_s_N5001' not in json.dumps(d['cpu_baseline'])
        dg = d['stereo_vio']['diagnostics']
        assert len(dg['pipelined_runs_frames_per_s']['runs']) == 3 and dg['pipelined_runs_frames_per_s']['median'] == d['stereo_vio']['value']
        assert dg['gpu_side_ms_per_step']['pipelined']['frozen_replay_gpu_ms'] > 0 and dg['shader_clock_mhz']['during_pipelined_median'] > 1000
        assert dg['miopen_pinned_db']['matches_device_and_version'] is True and set(dg['pipelined_host_stage_ms_per_batch']) == {'vo', 'imu', 'pgo', 'opt'}
        assert d['stereo_vio']['schedule_fallback'] is None and 'conv_nhwc_kernel' in d['stereo_vio']['nets']
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'pvgo_lm_iters_per_sec' and d['unit'] == 'LM iters/s' and d['higher_is_better'] is True
    assert d['vs_baseline'] is None and d['dtype'] == 'f64' and d['data'] == 'synthetic' and 'workload' in d['config']
    assert not any(k in d['config'] for k in ('model', 'seq_len', 'global_batch'))
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    assert r['traffic'] is None or r['traffic'] > 0
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str)
    assert abs(d['value'] * d['ms_per_step'] / 1e3 - d['lm_iters_per_step']) < 1e-6 * d['lm_iters_per_step'] + 1e-9
