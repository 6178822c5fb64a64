#!/bin/bash
# LM iteration time against the polling interval of the down-sweep's ready words (s_sleep argument in wait_ready).
cd $GRAFT_REPO_ROOT/islam_amd/csrc
for n in 8 4 2 16; do
  touch pvgo.hip
  make FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter -DISLAM_POLL_SLEEP=$n" > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  for i in 1 2; do
    python3 bench.py --no-frontend --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('POLL_SLEEP=$n', round(d['value'],1), 'it/s', round(d['us_per_lm_iter'],2), 'us/iter')"
  done
  cd islam_amd/csrc
done
