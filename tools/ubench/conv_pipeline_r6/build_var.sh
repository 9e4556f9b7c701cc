#!/bin/bash
# build_var.sh name "flags": a conv_h_bench variant from the tree's pair kernels with extra -D flags
set -e
cd /root/repo
C=image-matching_amd/csrc; F="-O3 -std=c++17 --offload-arch=gfx950 -Iinclude -I$C -mllvm -amdgpu-mfma-vgpr-form=1"
O=/tmp/ab_$1; mkdir -p $O /tmp/ab_common
for f in conv3x3_wino24 conv3x3_wino24h conv1ab_wino24 conv1ab_wino24h; do
  [ -f /tmp/ab_common/$f.o ] || /opt/rocm/bin/hipcc $F -c $C/$f.hip -o /tmp/ab_common/$f.o
done
/opt/rocm/bin/hipcc $F $2 -c $C/conv3x3_wino24p.hip -o $O/p3.o &
/opt/rocm/bin/hipcc $F $2 -c $C/conv1ab_wino24p.hip -o $O/p1.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude $2 -x hip -c tools/ubench/conv_h_bench.cpp -o $O/bench.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/ab_common/*.o $O/*.o -o tools/tmp_ab/conv_h_bench_$1
