#!/bin/bash
# debug builds of the pose head: no MFMA / no global loads in the chunk loop (differential timing; results are wrong by design)
set -e
cd "$(dirname "$0")/../../islam_amd/csrc"
make -s -j8
for v in NO_MFMA NO_LOAD; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_POSE_DBG_$v -c pose_head.hip -o /tmp/pose_head_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libislam_probe_pose_$v.so $(ls build/*.o | grep -v 'pose_head\.o\|_stamps\.o') /tmp/pose_head_$v.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
done
