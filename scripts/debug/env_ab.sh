#!/bin/bash
# A/B of one environment switch on the full bilevel step (scripts/vio_only.py), alternating fresh processes in one job:
#   bash scripts/debug/env_ab.sh ISLAM_CONV1X1_BN 0 1
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v python3 scripts/vio_only.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['diagnostics']['gpu_side_ms_per_step']
print('$VAR=$v  pipelined %.1f f/s (%.3f ms)  sequential %.1f  forward-only %.1f  replay %.3f / %.3f ms (pipelined / sequential)' % (d['value'], d['ms_per_batch'], d['sequential_frames_per_s'], d['forward_only_frames_per_s'], g['pipelined']['frozen_replay_gpu_ms'], g['sequential']['frozen_replay_gpu_ms']))"
  done
done
