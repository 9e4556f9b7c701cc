#!/bin/bash
# builds tools/ubench/conv_h_bench; SUFFIX=_trace tools/ubench/build_conv_bench.sh -DH_TRACE -DP_TRACE adds the phase clocks
# (-DP_TRACE_WAVES: the eight waves of workgroup 0, printed with TRACE_WAVES=1)
set -e
cd "$(dirname "$0")/../.."
C=image-matching_amd/csrc; O=/tmp/conv_h_bench_obj; mkdir -p $O
F="-O3 -std=c++17 --offload-arch=gfx950 -Iinclude"
EXTRA="$@"
for f in conv3x3_wino24 conv3x3_wino24h conv1ab_wino24 conv1ab_wino24h conv1ab_wino24p; do
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA -c $C/$f.hip -o $O/$f.o &
done
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA -c $C/conv3x3_wino24p.hip -o $O/conv3x3_wino24p.o &
/opt/rocm/bin/hipcc $F $EXTRA -x hip -c tools/ubench/conv_h_bench.cpp -o $O/bench.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 $O/*.o -o tools/ubench/conv_h_bench${SUFFIX}
