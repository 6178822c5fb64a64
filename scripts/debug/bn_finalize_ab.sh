#!/bin/bash
# A/B of the convbn statistics path on the full bilevel step (scripts/vio_only.py): ISLAM_BN_FOLD_FINALIZE=0 (three calls per convbn)
# against the default (finalize reads the persistent kernels' rows directly), alternating runs in one job
for rep in 1 2; do
  for v in 0 1; do
    ISLAM_BN_FOLD_FINALIZE=$v python3 scripts/vio_only.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['diagnostics']['gpu_side_ms_per_step']
print('BN_FOLD_FINALIZE=$v  pipelined %.1f f/s (%.3f ms)  sequential %.1f  forward-only %.1f  replay %.3f / %.3f ms (pipelined / sequential)' % (d['value'], d['ms_per_batch'], d['sequential_frames_per_s'], d['forward_only_frames_per_s'], g['pipelined']['frozen_replay_gpu_ms'], g['sequential']['frozen_replay_gpu_ms']))"
  done
done
