// split_dot2.hip -- is the v_dot2c_f32_bf16 form of the three-term bf16 split (csrc/split3.h) the same bits as the shift / subtract
// form?  Every fp32 value whose low 8 mantissa bits are swept over 2^24 patterns x a set of exponents (incl. zeros, subnormals, the
// largest finite values, both signs); the three planes must agree bit for bit, and h + m + l == x exactly where x is normal and
// the residuals are not subnormal.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/split_dot2.hip -o tools/ubench/split_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define IMX_SPLIT_DOT2 1
#include "../../image-matching_amd/csrc/split3.h"
using namespace imx;
__device__ __forceinline__ void split_ref(float x0, float x1, split_bf16x2& h, split_bf16x2& m, split_bf16x2& l) {
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  l[0] = (__bf16)(r0 - (float)m[0]); l[1] = (__bf16)(r1 - (float)m[1]);
}
__global__ void check(unsigned expo, unsigned long long* bad, unsigned long long* inexact, unsigned* first) {
  const unsigned mant = blockIdx.x * blockDim.x + threadIdx.x;          // 2^23 mantissas
  for (unsigned sign = 0; sign < 2; ++sign) {
    const unsigned b0 = (sign << 31) | (expo << 23) | mant, b1 = (sign << 31) | (expo << 23) | ((mant * 2654435761u) & 0x7FFFFF);
    const float x0 = __builtin_bit_cast(float, b0), x1 = __builtin_bit_cast(float, b1);
    split_bf16x2 h, m, l, hr, mr, lr;
    split3_pair(x0, x1, h, m, l);
    split_ref(x0, x1, hr, mr, lr);
    const unsigned a = __builtin_bit_cast(unsigned, h) ^ __builtin_bit_cast(unsigned, hr), b = __builtin_bit_cast(unsigned, m) ^ __builtin_bit_cast(unsigned, mr),
                   c = __builtin_bit_cast(unsigned, l) ^ __builtin_bit_cast(unsigned, lr);
    if (a | b | c) { atomicAdd(bad, 1ull); atomicCAS(first, 0u, b0); }
    const float s0 = ((float)l[0] + (float)m[0]) + (float)h[0];
    if (s0 != x0 && expo != 0 && expo != 255) atomicAdd(inexact, 1ull);
  }
}
int main() {
  unsigned long long *bad, *inexact; unsigned* first;
  hipMalloc(&bad, 8); hipMalloc(&inexact, 8); hipMalloc(&first, 4);
  unsigned long long tot_bad = 0;
  for (unsigned e = 0; e < 255; ++e) {
    hipMemset(bad, 0, 8); hipMemset(inexact, 0, 8); hipMemset(first, 0, 4);
    hipLaunchKernelGGL(check, dim3(1 << 15), dim3(256), 0, 0, e, bad, inexact, first);
    unsigned long long hb, hi; unsigned hf;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hi, inexact, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    tot_bad += hb;
    if (hb || e < 3 || e == 127 || e == 254) printf("exponent %3u: %llu of 2^24 pairs differ from the shift/subtract split (first %08x); h+m+l != x on %llu\n", e, hb, hf, hi);
  }
  printf("all 255 exponents x 2^23 mantissas x 2 signs: %llu pairs differ\n", tot_bad);
  return 0;
}
