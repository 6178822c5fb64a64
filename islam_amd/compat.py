"""Drop-in installer: lets the reference's own ``train.py`` import this package under the module names it
expects.  ``islam_amd.compat.install()`` registers

    pypose                         -> islam_amd.lietensor   (only the names train.py / the kept modules touch)
    TartanVO, pvgo, imu_integrator, dense_ba -> islam_amd.TartanVO / .pvgo / .imu_integrator / .dense_ba
    Datasets.transformation        -> islam_amd.transformation (PyPose-side helpers only)
    Network.PWC.correlation        -> FunctionCorrelation on the HIP kernel

in ``sys.modules``; see INTEGRATION.md."""
import sys
import types


def install(force=False):
    from . import TartanVO, dense_ba, imu_integrator, lietensor, ops, pvgo, transformation
    table = {'pypose': lietensor, 'TartanVO': TartanVO, 'pvgo': pvgo, 'imu_integrator': imu_integrator, 'dense_ba': dense_ba}
    for name, mod in table.items():
        if force or name not in sys.modules:
            sys.modules[name] = mod
    corr = types.ModuleType('Network.PWC.correlation')
    corr.FunctionCorrelation = ops.FunctionCorrelation
    sys.modules.setdefault('Network.PWC.correlation', corr)
    tr = sys.modules.get('Datasets.transformation')
    if tr is not None:
        for n in ('cvtSE3_pypose', 'tartan2kitti_pypose', 'motion2pose_pypose', 'pose2motion_pypose'):
            setattr(tr, n, getattr(transformation, n))
    return table
