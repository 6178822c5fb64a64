#!/bin/bash
# builds islam_amd/lib/libislam_probe${SUFFIX}.so = the product library with -DISLAM_PROBE (phase timestamps) [+ EXTRA flags]
# usage: scripts/build_probe.sh [SUFFIX [EXTRA_FLAGS...]]
set -e
SUFFIX=$1; shift || true
cd "$(dirname "$0")/../islam_amd/csrc"
O=/tmp/probe_obj$SUFFIX
mkdir -p $O
for f in abi pvgo corr_warp imu_preint scale_ls conv_mfma conv_nhwc edge_mask pvgo_dist pose_ops pyramid; do
  if [ $f = pvgo ] || [ ! -f $O/$f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_PROBE "$@" -c $f.hip -o $O/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libislam_probe$SUFFIX.so $O/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
