#!/bin/bash
# builds islam_amd/lib/libislam_probe_corr.so = the product library with corr_warp.hip compiled -DISLAM_CORR_STAMPS (run on the build box)
set -e
cd "$(dirname "$0")/../../islam_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_CORR_STAMPS -c corr_warp.hip -o /tmp/corr_warp_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libislam_probe_corr.so $(ls build/*.o | grep -v 'corr_warp\.o') /tmp/corr_warp_stamps.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
