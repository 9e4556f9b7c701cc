// wino24_pk.h — the F(4,3) input transform of conv1ab_wino24.hip / conv3x3_wino24.hip as 18 pinned packed instructions.
// hipcc scalarises a third of these v_pk_* operations (8 become v_fma_f32 / v_add_f32 pairs: 28 VALU instructions instead
// of 18 per chunk and wave), and on this SIMD every VALU instruction costs ~5 cycles of matrix-pipe time
// (tools/ubench/mfma_mix.hip), so they are written as inline assembly.  All operate on a channel pair (f32x2).
#pragma once

namespace imx {

typedef float w24_f32x2 __attribute__((ext_vector_type(2)));

#define IMX_PK3(name_, text_)                                                                      \
  __device__ __forceinline__ w24_f32x2 name_(w24_f32x2 a, w24_f32x2 b, w24_f32x2 c) {              \
    w24_f32x2 r;                                                                                   \
    asm volatile(text_ : "=v"(r) : "v"(a), "v"(b), "v"(c));                                        \
    return r;                                                                                      \
  }
#define IMX_PK2(name_, text_)                                                                      \
  __device__ __forceinline__ w24_f32x2 name_(w24_f32x2 a, w24_f32x2 b) {                           \
    w24_f32x2 r;                                                                                   \
    asm volatile(text_ : "=v"(r) : "v"(a), "v"(b));                                                \
    return r;                                                                                      \
  }
IMX_PK3(pk_fma, "v_pk_fma_f32 %0, %1, %2, %3")                                                              // a*b + c
IMX_PK2(pk_add, "v_pk_add_f32 %0, %1, %2")                                                                  // a + b
IMX_PK2(pk_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                                        // a - b
IMX_PK2(pk_fma_p4, "v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]")                                        //  4a + b
IMX_PK2(pk_fma_m4, "v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")          // -4a + b
IMX_PK2(pk_fma_p2, "v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]")                                        //  2a + b
IMX_PK2(pk_fma_m2, "v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")          // -2a + b
#undef IMX_PK3
#undef IMX_PK2

// B4^T of F(4,3) on six values o[0..5] (one transformed row of the patch), in two batches of six instructions:
//   t0 = 4 o0 - 5 o2 + o4      t1 = (o4 - 4 o2) + (o3 - 4 o1)     t2 = (o4 - 4 o2) - (o3 - 4 o1)
//   t5 = 4 o1 - 5 o3 + o5      t3 = (o4 - o2) + 2 (o3 - o1)       t4 = (o4 - o2) - 2 (o3 - o1)
struct W24Half { w24_f32x2 e42, e31, f42, f31, t0, t5; };
__device__ __forceinline__ W24Half w24_batch_a(const w24_f32x2 (&o)[6], w24_f32x2 m5) {
  W24Half h;
  h.e42 = pk_fma_m4(o[2], o[4]);
  h.e31 = pk_fma_m4(o[1], o[3]);
  h.f42 = pk_sub(o[4], o[2]);
  h.f31 = pk_sub(o[3], o[1]);
  h.t0 = pk_fma(o[2], m5, o[4]);         // completed in batch b
  h.t5 = pk_fma(o[3], m5, o[5]);
  return h;
}
__device__ __forceinline__ void w24_batch_b(const w24_f32x2 (&o)[6], const W24Half& h, w24_f32x2 (&t)[6]) {
  t[0] = pk_fma_p4(o[0], h.t0);
  t[1] = pk_add(h.e42, h.e31);
  t[2] = pk_sub(h.e42, h.e31);
  t[3] = pk_fma_p2(h.f31, h.f42);
  t[4] = pk_fma_m2(h.f31, h.f42);
  t[5] = pk_fma_p4(o[1], h.t5);
}

}  // namespace imx
