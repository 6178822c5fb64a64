#!/bin/bash
# Experiment builds of conv_nhwc.hip / conv_mfma.hip (see ISLAM_CONV_PROBE there) and their timings on the nets' shapes:
# 0 = the product kernel, 1 = fetch + staging only, 2 = multiply phase only.   Run on the GPU box:  bash scripts/conv_probe.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cd $ROOT/islam_amd/csrc
mkdir -p /tmp/cprobe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c abi.hip -o /tmp/cprobe/abi.o
for v in 0 1 2; do
  for f in conv_nhwc conv_mfma; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_CONV_PROBE=$v -c $f.hip -o /tmp/cprobe/${f}_$v.o
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/cprobe/libconv_$v.so /tmp/cprobe/conv_nhwc_$v.o /tmp/cprobe/conv_mfma_$v.o /tmp/cprobe/abi.o
done
cd $ROOT
for v in 0 1 2; do echo "== ISLAM_CONV_PROBE=$v"; python3 scripts/conv_probe.py /tmp/cprobe/libconv_$v.so; done
