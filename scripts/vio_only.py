"""bench.py's stereo_vio measurement alone (full bilevel step at B=8, pipelined and sequential): A/B of front-end settings."""
import json, os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # as bench.py: before the HIP runtime initialises
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
r = bench.vio_frames_per_sec(torch.device("cuda:0"), steps=int(os.environ.get("VIO_STEPS", "64")))
print(json.dumps({k: r[k] for k in ('value', 'ms_per_batch', 'sequential_frames_per_s', 'sequential_ms_per_batch', 'forward_only_frames_per_s', 'diagnostics')}))
