#!/bin/bash
# build_old.sh name "flags": conv_h_bench with the round-5 pair kernels (tools/tmp_ab/*_r5.hip) and extra flags
set -e
cd /root/repo
C=image-matching_amd/csrc; F="-O3 -std=c++17 --offload-arch=gfx950 -Iinclude -I$C -mllvm -amdgpu-mfma-vgpr-form=1"
O=/tmp/ab_$1; mkdir -p $O /tmp/ab_common
/opt/rocm/bin/hipcc $F $2 -x hip -c tools/tmp_ab/conv3x3_wino24p_r5.hip -o $O/p3.o &
/opt/rocm/bin/hipcc $F $2 -x hip -c tools/tmp_ab/conv1ab_wino24p_r5.hip -o $O/p1.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude $2 -x hip -c tools/ubench/conv_h_bench.cpp -o $O/bench.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/ab_common/*.o $O/*.o -o tools/tmp_ab/conv_h_bench_$1
