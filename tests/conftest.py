import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if os.environ.get('ISLAM_ABORT_TRACE') == '1':        # debug aid: native call stack on a silent abort() (scripts/debug)
        import ctypes
        import subprocess
        src = os.path.join(ROOT, 'scripts', 'debug', 'abort_trace.c')
        so = os.path.join('/tmp', 'libabort_trace.so')
        subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-o', so, src])
        lib = ctypes.CDLL(so)
        lib.install_abort_trace()
        config._abort_trace_lib = lib


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')
