"""A few eager forwards of the frozen stereo and flow nets (B=8, 448x640, bf16 execution copies) -- the workload of scripts/frozen_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev).train()
vonet.set_frozen_dtype(torch.bfloat16, torch.bfloat16)
x = torch.randn(8, 6, 448, 640, device=dev)
with torch.no_grad():
    for _ in range(int(os.environ.get('FWD_REPS', '4'))):
        vonet._run_frozen('stereo', vonet.stereoNet, torch.bfloat16, x, quarter=True)
        vonet._run_frozen('flow', vonet.flowNet, torch.bfloat16, x)
torch.cuda.synchronize()
