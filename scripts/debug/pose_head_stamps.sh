#!/bin/bash
# builds islam_amd/lib/libislam_probe_pose.so = the product library with pose_head.hip compiled -DISLAM_POSE_STAMPS (run on the build box)
set -e
cd "$(dirname "$0")/../../islam_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_POSE_STAMPS -c pose_head.hip -o /tmp/pose_head_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libislam_probe_pose.so $(ls build/*.o | grep -v 'pose_head\.o\|_stamps\.o') /tmp/pose_head_stamps.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
