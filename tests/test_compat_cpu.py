"""The drop-in claim "train.py runs unchanged" (SURVEY.md section 8b): after islam_amd.compat.install() every PyPose name the
reference's kept files touch resolves on the shim, the replaced modules import under the reference's module names with the
reference's call signatures, and -- in the build container, where /root/reference exists -- the reference's own
Datasets/transformation.py executes on top of the shim and agrees with islam_amd.transformation.
tests/golden/pypose_names.json is produced by tests/golden/make_pypose_names.py (ast over the reference)."""
import importlib
import inspect
import json
import os
import sys

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), 'golden')
IMPLEMENTED_INSIDE = ('pypose.optim', 'pypose.module', 'pypose.function')      # LM loop, pre-integrator, reprojection: HIP side


@pytest.fixture()
def installed():
    from islam_amd import compat
    saved = {k: sys.modules.get(k) for k in ('pypose', 'TartanVO', 'pvgo', 'imu_integrator', 'dense_ba', 'Network.PWC.correlation')}
    compat.install(force=True)
    yield sys.modules['pypose']
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _resolve(root, dotted):
    obj = root
    for part in dotted.split('.')[1:]:
        obj = getattr(obj, part)
    return obj


def test_every_pypose_name_of_the_kept_files_resolves(installed):
    names = json.load(open(os.path.join(G, 'pypose_names.json')))
    pp = installed
    X = pp.SE3(torch.tensor([[0.1, 0.2, 0.3, 0, 0, 0, 1.0]]))
    for grp in ('kept', 'replaced'):
        for path, rec in names[grp].items():
            for d in rec['pp'] + rec['from_imports']:
                if grp == 'replaced' and d.startswith(IMPLEMENTED_INSIDE):
                    continue
                assert _resolve(pp, d) is not None, (path, d)
            if grp == 'kept':
                for m in rec['lie_methods']:
                    assert hasattr(X, m), (path, m)
    # the LieTensor surface SURVEY section 8b lists, on an actual object
    for m in ('Inv', 'Log', 'rotation', 'translation', 'tensor', 'matrix', 'to', 'cpu', 'detach', 'clone', 'numpy'):
        assert callable(getattr(X, m)), m
    assert isinstance(X @ X, pp.LieTensor) and isinstance(X[0], pp.LieTensor) and len(X) == 1
    assert isinstance(pp.identity_SO3(), pp.LieTensor) and pp.SE3_type is not pp.se3_type
    assert isinstance(pp.Parameter(X), torch.nn.Parameter)


def test_replaced_modules_import_under_the_reference_names(installed):
    from imu_integrator import IMUModule            # noqa: F401  (train.py:7-8)
    from pvgo import run_pvgo
    from TartanVO import TartanVO
    import dense_ba
    sig = inspect.signature(run_pvgo)
    ref_args = ['init_nodes', 'init_vels', 'vo_motions', 'links', 'dts', 'imu_drots', 'imu_dtrans', 'imu_dvels', 'device', 'radius',
                'loss_weight', 'reproj', 'target']                                      # pvgo.py:122-123
    assert list(sig.parameters)[:len(ref_args)] == ref_args
    assert sig.parameters['device'].default == 'cuda:0' and sig.parameters['radius'].default == 1e4 and sig.parameters['target'].default == 'vo'
    ref_args = ['self', 'vo_model_name', 'pose_model_name', 'flow_model_name', 'stereo_model_name', 'device_id', 'correct_scale',
                'fix_parts', 'use_kitti_coord']                                         # TartanVO.py:17-18
    assert list(inspect.signature(TartanVO.__init__).parameters)[:len(ref_args)] == ref_args
    fwd = inspect.signature(TartanVO.forward).parameters                                # TartanVO.py:90; extras only behind, with defaults
    assert list(fwd)[:4] == ['self', 'sample', 'is_train', 'given_scale']
    assert all(p.default is not inspect.Parameter.empty for p in list(fwd.values())[4:])
    ref_args = ['self', 'accels', 'gyros', 'dts', 'accel_bias', 'gyro_bias', 'init', 'gravity', 'rgb2imu_sync', 'device',
                'denoise_model_name', 'denoise_accel', 'denoise_gyro', 'use_est_cov']  # imu_integrator.py:31-33
    assert list(inspect.signature(IMUModule.__init__).parameters)[:len(ref_args)] == ref_args
    assert list(inspect.signature(IMUModule.integrate).parameters)[:5] == ['self', 'st', 'end', 'init', 'motion_mode']
    assert hasattr(dense_ba, 'scale_from_disp_flow') and hasattr(dense_ba, 'SparseReprojectionLoss')
    from Network.PWC.correlation import FunctionCorrelation                            # noqa: F401  (PWCNet.py import)


@pytest.mark.skipif(not os.path.exists('/root/reference/Datasets/transformation.py'), reason='build container only: needs /root/reference')
def test_reference_transformation_module_runs_on_the_shim(installed):
    """The reference's OWN Datasets/transformation.py:72-124, executed on top of the shim, against islam_amd.transformation."""
    from islam_amd import lietensor as pp, transformation as tf
    spec = importlib.util.spec_from_file_location('_ref_transformation', '/root/reference/Datasets/transformation.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(0)
    m6 = torch.randn(7, 6, generator=g, dtype=torch.float64) * 0.3
    for fn in ('cvtSE3_pypose', 'tartan2kitti_pypose'):
        a, b = getattr(ref, fn)(m6), getattr(tf, fn)(m6)
        assert isinstance(a, pp.LieTensor)
        np.testing.assert_array_equal(a.tensor().numpy(), b.tensor().numpy())
    K = ref.tartan2kitti_pypose(m6)
    T0 = pp.se3(torch.randn(6, generator=g, dtype=torch.float64) * 0.2).Exp()
    Pa, Pb = ref.motion2pose_pypose(K, T0), tf.motion2pose_pypose(K, T0)
    np.testing.assert_array_equal(Pa.tensor().numpy(), Pb.tensor().numpy())
    np.testing.assert_array_equal(ref.pose2motion_pypose(Pa).tensor().numpy(), tf.pose2motion_pypose(Pb).tensor().numpy())
    # se3 / SE3 inputs and the gradient path train.py relies on (motions carry grad into motion2pose)
    m = m6.clone().requires_grad_(True)
    ref.motion2pose_pypose(ref.tartan2kitti_pypose(m), T0).tensor().sum().backward()
    ga = m.grad.clone()
    m.grad = None
    tf.motion2pose_pypose(tf.tartan2kitti_pypose(m), T0).tensor().sum().backward()
    np.testing.assert_allclose(ga.numpy(), m.grad.numpy(), atol=1e-14)
