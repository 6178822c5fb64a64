#!/bin/bash
# per-phase timeline of one workgroup of conv_nhwc_kernel.  Run on the GPU box:  bash scripts/conv_phase_probe.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cd $ROOT/islam_amd/csrc
mkdir -p /tmp/cprobe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c abi.hip -o /tmp/cprobe/abi.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_CONV_PROBE=3 -c conv_nhwc.hip -o /tmp/cprobe/conv_nhwc_3.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/cprobe/libconv_3.so /tmp/cprobe/conv_nhwc_3.o /tmp/cprobe/abi.o
cd $ROOT
python3 scripts/conv_phase_probe.py /tmp/cprobe/libconv_3.so
