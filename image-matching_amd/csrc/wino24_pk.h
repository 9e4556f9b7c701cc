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

// a 128-bit accumulator cleared with TWO v_mov_b64 (hipcc materialises a zero f32x4 as four v_mov_b32; 24 accumulators
// per work item: 96 -> 48 instructions, each ~4.5 cycles of matrix-pipe time on this SIMD)
typedef float w24_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w24_f32x4 w24_zero4() {
  w24_f32x2 a, b;
  asm volatile("v_mov_b64 %0, 0" : "=v"(a));
  asm volatile("v_mov_b64 %0, 0" : "=v"(b));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3);
}
__device__ __forceinline__ w24_f32x2 w24_lo(w24_f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ w24_f32x2 w24_hi(w24_f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// Output transform Y = A2^T M A4 of one lane's 24 accumulators (m[j*4 + i]: position (i, j), four consecutive channels
// each), on channel PAIRS with pinned packed instructions -- left to the compiler a third of it is scalarised (64 v_sub_f32):
//   down the rows     s0[j] = m0j + m1j + m2j        s1[j] = m1j - m2j - m3j
//   along the columns y[r][0] = s0 + (s1+s2) + (s3+s4)      y[r][1] = (s1-s2) + 2 (s3-s4)
//                     y[r][2] = (s1+s2) + 4 (s3+s4)         y[r][3] = (s1-s2) + 8 (s3-s4) + s5
// 2 x (24 + 20) = 88 packed instructions for the 8 x 4 outputs of the lane.
// (in two stages, so that the pair kernels can run the row stage where the accumulators are and exchange its results -- half the
// registers -- between the two waves that share a tile: the same instructions on the same values either way)
__device__ __forceinline__ void w24_out_rows(const w24_f32x4& m0, const w24_f32x4& m1, const w24_f32x4& m2, const w24_f32x4& m3, w24_f32x4& s0, w24_f32x4& s1) {
  const w24_f32x2 s0l = pk_add(pk_add(w24_lo(m0), w24_lo(m1)), w24_lo(m2)), s1l = pk_sub(pk_sub(w24_lo(m1), w24_lo(m2)), w24_lo(m3));
  const w24_f32x2 s0h = pk_add(pk_add(w24_hi(m0), w24_hi(m1)), w24_hi(m2)), s1h = pk_sub(pk_sub(w24_hi(m1), w24_hi(m2)), w24_hi(m3));
  s0 = __builtin_shufflevector(s0l, s0h, 0, 1, 2, 3);
  s1 = __builtin_shufflevector(s1l, s1h, 0, 1, 2, 3);
}
__device__ __forceinline__ void w24_out_cols(const w24_f32x4 (&s0)[6], const w24_f32x4 (&s1)[6], w24_f32x2 k8, w24_f32x4 (&y)[2][4]) {
  w24_f32x2 out[2][4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)                    // channel pair: registers 0-1 / 2-3
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      w24_f32x2 s[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) s[j] = h ? w24_hi(r ? s1[j] : s0[j]) : w24_lo(r ? s1[j] : s0[j]);
      const w24_f32x2 a12 = pk_add(s[1], s[2]), b12 = pk_sub(s[1], s[2]), c34 = pk_add(s[3], s[4]), d34 = pk_sub(s[3], s[4]);
      out[r][0][h] = pk_add(pk_add(s[0], a12), c34);
      out[r][1][h] = pk_fma_p2(d34, b12);
      out[r][2][h] = pk_fma_p4(c34, a12);
      out[r][3][h] = pk_add(pk_fma(d34, k8, b12), s[5]);
    }
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int x = 0; x < 4; ++x) y[r][x] = __builtin_shufflevector(out[r][x][0], out[r][x][1], 0, 1, 2, 3);
}
__device__ __forceinline__ void w24_output_transform(const w24_f32x4 (&m)[24], w24_f32x2 k8, w24_f32x4 (&y)[2][4]) {
  w24_f32x4 s0[6], s1[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) w24_out_rows(m[j * 4 + 0], m[j * 4 + 1], m[j * 4 + 2], m[j * 4 + 3], s0[j], s1[j]);
  w24_out_cols(s0, s1, k8, y);
}

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
