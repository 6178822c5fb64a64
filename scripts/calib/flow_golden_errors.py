"""Error statistics of PWCDCNet.forward_mfma against the reference-generated flow vectors (tests/golden/nets_pwc.npz) for the settings
of the round-3 flow kernels: max / rms / bias per output, as tests/test_golden_gpu.py::test_flow_net_matrix_core_path_matches_reference."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) == 1:
    for env in ({}, {'ISLAM_FLOW_HEAD_MIRROR': '0'}, {'ISLAM_FLOW_PYR': '0'}, {'ISLAM_FLOW_HEAD_MIRROR': '0', 'ISLAM_FLOW_PYR': '0'},
                {'ISLAM_FLOW_HEAD_MIRROR': '0', 'ISLAM_FLOW_PYR': '0', 'ISLAM_FLOW_UP2': '0'}):
        print('settings', env or '(default)', flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, **env))
    sys.exit(0)
import numpy as np, torch
from tests.golden.netfill import fill_state_dict, make_input
from tests.test_golden_gpu import _g, _stats
from islam_amd import nets
ref = _g('pwc')
net = fill_state_dict(nets.PWCDCNet()).to('cuda').eval()
x = make_input('pwc').to('cuda')
with torch.no_grad():
    for rep in range(2):
        flows, _ = net.forward_mfma(x)
        print('  call %d:' % rep, '  '.join('flow%d max %.3e rms %.3e bias %+.1e' % ((i,) + _stats(f, ref['flow%d' % i])) for i, f in enumerate(flows)))
print('  input', tuple(x.shape))
