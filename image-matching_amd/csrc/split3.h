// split3.h -- x = h + m + l, an fp32 value as three bf16 terms (8 significant bits each, round-to-nearest residuals): the operand
// split of every "six bf16 term products per fp32 product" kernel (gemm_x3.hip, gnn_tail_x3.hip, attention_x3.hip).
//
// Round 4: the residuals r = x - float(h) and r - float(m) are ONE instruction per value, v_dot2c_f32_bf16 (D += a.lo * b.lo +
// a.hi * b.hi with the packed pair (h0, h1) as `a` and the constant (-1, 0) / (0, -1) as `b`): the products of a bf16 value with
// -1 / 0 are exact and the sum x - h is exactly representable (it is the rounding residual of x at 8 bits), so the result is the
// same bits as the unpack (shift / mask) + v_sub_f32 pair it replaces -- 3.5 VALU instructions per value instead of 5.5
// (tools/ubench/split_dot2.hip checks all three planes bit for bit over 2^24 values incl. zeros, subnormal residuals and the
// largest finite values).  IMX_SPLIT_DOT2=0 at compile time keeps the shift/subtract form (A/B).
#pragma once
#include <hip/hip_runtime.h>

#ifndef IMX_SPLIT_DOT2
#define IMX_SPLIT_DOT2 1
#endif

namespace imx {

typedef __bf16 split_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3_pair(float x0, float x1, split_bf16x2& h, split_bf16x2& m, split_bf16x2& l) {
#if IMX_SPLIT_DOT2
  // The constants go through SGPRs behind an (un-foldable, side-effect-free) asm: hipcc 7.2 encodes the packed constant {-1, 0} as
  // the INLINE constant -1.0, which this instruction on gfx950 does not read as bf16 (-1, 0) -- the result is x + 0.0034 instead of
  // x - h (tools/ubench/split_dot2.hip's first version; a literal or a register operand is correct).
  unsigned lo_u, hi_u;
  asm("s_mov_b32 %0, 0x0000bf80" : "=s"(lo_u));
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(hi_u));
  const split_bf16x2 lo = __builtin_bit_cast(split_bf16x2, lo_u), hi = __builtin_bit_cast(split_bf16x2, hi_u);
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;                                    // v_cvt_pk_bf16_f32
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(h, lo, x0, false);      // x0 - h0
  const float r1 = __builtin_amdgcn_fdot2_f32_bf16(h, hi, x1, false);      // x1 - h1
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  l[0] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(m, lo, r0, false);
  l[1] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(m, hi, r1, false);
#else
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  l[0] = (__bf16)(r0 - (float)m[0]); l[1] = (__bf16)(r1 - (float)m[1]);
#endif
}

}  // namespace imx
