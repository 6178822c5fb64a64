#!/bin/bash
# builds islam_amd/lib/libislam_probe.so = the product library with -DISLAM_PROBE (phase timestamps)
set -e
cd "$(dirname "$0")/../islam_amd/csrc"
mkdir -p /tmp/probe_obj
for f in abi pvgo corr_warp imu_preint scale_ls conv_mfma conv_nhwc edge_mask pvgo_dist pose_ops; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_PROBE -c $f.hip -o /tmp/probe_obj/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libislam_probe.so /tmp/probe_obj/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
