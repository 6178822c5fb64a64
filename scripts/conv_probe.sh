#!/bin/bash
# Experiment builds of conv_nhwc.hip (see ISLAM_CONV_PROBE there) and their timings on the stereo net's shapes.
# Run on the GPU box:  bash scripts/conv_probe.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cd $ROOT/islam_amd/csrc
mkdir -p /tmp/cprobe
for v in 0 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_CONV_PROBE=$v -c conv_nhwc.hip -o /tmp/cprobe/conv_$v.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c abi.hip -o /tmp/cprobe/abi.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/cprobe/libconv_$v.so /tmp/cprobe/conv_$v.o /tmp/cprobe/abi.o
done
cd $ROOT
for v in 0 1 2; do echo "== ISLAM_CONV_PROBE=$v"; python3 scripts/conv_probe.py /tmp/cprobe/libconv_$v.so; done
