#!/bin/bash
# LM iteration time against the block size of linbuild / trial_lin (nodes per workgroup).  Rebuilds the library in place on the GPU box.
cd $GRAFT_REPO_ROOT/islam_amd/csrc
for n in 63 31 21 15; do
  touch pvgo.hip
  make FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter -DISLAM_LB_NODES=$n" > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  for i in 1 2; do
    python3 bench.py --no-frontend --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LB_NODES=$n', round(d['value'],1), 'it/s', round(d['us_per_lm_iter'],2), 'us/iter')"
  done
  cd islam_amd/csrc
done
