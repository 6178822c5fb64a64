#!/bin/bash
# per-phase timeline of one workgroup of hg_residual_kernel (wall-clock stamps, 100 MHz).  Run on the GPU box:  bash scripts/hg_probe.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
cd $ROOT/islam_amd/csrc
mkdir -p /tmp/hprobe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c abi.hip -o /tmp/hprobe/abi.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_HG_PROBE=1 -c hourglass.hip -o /tmp/hprobe/hg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/hprobe/libhg.so /tmp/hprobe/hg.o /tmp/hprobe/abi.o
cd $ROOT
python3 scripts/hg_probe.py /tmp/hprobe/libhg.so
