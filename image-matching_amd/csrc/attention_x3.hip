// attention_x3.hip — multi-head attention (superglue_test.py:85-89; heads at :92-109) with both fp32 products, S = Q.K^T and
// O = P.V, carried by the bf16 matrix pipe as six bf16 term products each (the scheme of gemm_x3.hip: x = h + m + l exactly,
// six of the nine term products, one round-to-nearest per 16-term MFMA; tools/ubench/mfma_bf16x3.hip).
//
// Same flash-style structure as attention.hip: a workgroup owns 128 queries of one (pair side, head), its four waves 32 queries
// each, and walks the keys in 32-key tiles; S^T = K.Q^T so that a lane owns ONE query's scores (row maximum / row sum in-lane
// plus one exchange with lane^32), the log2-domain online softmax, O^T = V^T.P^T with P fed from the S accumulators, two-level
// accumulation of O (64-key groups started from a zero accumulator).  What changes:
//   * Q (pre-scaled) is split once per workgroup; K and V tiles are split while they are staged (fp32 from global, three bf16
//     planes into LDS); P is split in registers after the exponentials (v_cvt_pk_bf16_f32 rounds and packs two values).
//   * S^T: A operand = K rows (lane (key, kb): eight consecutive dims, one ds_read_b128 per plane from 80-byte rows -- an odd
//     number of 16-byte slots), B operand = Q^T from registers.
//   * O^T: the A operand V^T[dim][key] wants eight KEYS per lane for one dim -- a transposed read of the row-major V tile:
//     ds_read_b64_tr_b16 (tools/ubench/ds_tr16.hip pins its semantics) gives a lane V[key0 + j][16 g + i], j = 0..3, when the
//     sixteen lanes of a group pass the addresses of the 4 x 16 block row by row.  The MFMA's k index is mapped to the keys a
//     lane already holds in its S accumulators: k = 8 kb + j' <-> key (j' & 3) + 8 (j' >> 2) + 4 kb + 16 t, two transposed reads
//     per operand.  V rows are 64 / 128 bytes, unpadded (a 32-lane half reads 4 rows x 64 contiguous bytes).
//   * 24 bf16 MFMAs of 32 cycles per 32 x 32 tile at HD = 32 instead of 32 fp32 MFMAs of 64 cycles; the VALU work (softmax,
//     the splits) issues beneath the other waves' MFMAs instead of stopping the pipe (the fp32 MFMA shares the VALU datapath).
#include "imx_kernels.h"
#include "split3.h"

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

// ---- operand formats.  A format says how an fp32 operand is cut into 16-bit planes and which plane products are kept.
// FmtX3: x = h + m + l in bf16 (8 significant bits each), six of the nine term products -- fp32 products to ~2^-24, any exponent.
struct FmtX3 {
  static constexpr int NP = 3, NT = 6;
  static constexpr bool SCALED = false;
  typedef __bf16 T;
  typedef __bf16 x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 x2 __attribute__((ext_vector_type(2)));
  // term products, smallest first: planes (A, B) = (m,m) (h,l) (l,h) (h,m) (m,h) (h,h)
  static __device__ __forceinline__ constexpr int pa(int i) { constexpr int t[6] = {1, 0, 2, 0, 1, 0}; return t[i]; }
  static __device__ __forceinline__ constexpr int pb(int i) { constexpr int t[6] = {1, 2, 0, 1, 0, 0}; return t[i]; }
  static __device__ __forceinline__ void split(float x0, float x1, x2 (&pl)[3]) { split3_pair(x0, x1, pl[0], pl[1], pl[2]); }
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
// FmtH2 (round 4): x s = h + m in fp16 (11 significant bits each: 22 bits, truncation 2^-22 |x|), THREE term products (h,m) (m,h)
// (h,h) -- half the MFMAs of FmtX3 and a split of 2 instead of 3.5 VALU instructions per value (v_cvt_pk_f16_f32 rounds and packs
// two values, the residual x - h is one v_dot2c_f32_f16 per value, exact).  fp16 has five exponent bits, so every operand is
// scaled by a power of two s that brings the largest |value| of its (side, pair) (AttnArgs::amax: q, k, v maxima over the valid rows,
// written by qkv_amax below or by the producing gnn_tail_x3) to [2^13, 2^14): every value within 2^-17 of the maximum keeps its 22 bits, smaller ones are exact to 2^-39 of
// the maximum; P (<= 1) is scaled by 2^15 inside its exponential.  The powers of two cancel exactly (one fma in the softmax, the
// final 1 / l).  Against a float64 evaluation the result is as close as the six-product bf16 form's on P.V (P's own rounding in
// fp32 dominates both) and 0.66 x the fp32-MFMA kernel's error on Q.K^T (tools/f16_split_emul.py; tools/ubench/attn_x3_bench.cpp
// measures all three kernels).
struct FmtH2 {
  static constexpr int NP = 2, NT = 3;
  static constexpr bool SCALED = true;
  typedef _Float16 T;
  typedef _Float16 x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 x4 __attribute__((ext_vector_type(4)));
  typedef _Float16 x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ constexpr int pa(int i) { constexpr int t[3] = {0, 1, 0}; return t[i]; }
  static __device__ __forceinline__ constexpr int pb(int i) { constexpr int t[3] = {1, 0, 0}; return t[i]; }
  static __device__ __forceinline__ void split(float x0, float x1, x2 (&pl)[2]) {
    // constants through SGPRs behind an asm, as in split3.h (hipcc 7.2 folds a packed {-1, 0} into the inline constant -1.0)
    unsigned lo_u, hi_u;
    asm("s_mov_b32 %0, 0x0000bc00" : "=s"(lo_u));
    asm("s_mov_b32 %0, 0xbc000000" : "=s"(hi_u));
    const x2 lo = __builtin_bit_cast(x2, lo_u), hi = __builtin_bit_cast(x2, hi_u);
    pl[0][0] = (_Float16)x0; pl[0][1] = (_Float16)x1;                                  // v_cvt_pk_f16_f32 (round to nearest even)
    const float r0 = __builtin_amdgcn_fdot2(pl[0], lo, x0, false);                     // x0 - h0, exact
    const float r1 = __builtin_amdgcn_fdot2(pl[0], hi, x1, false);
    pl[1][0] = (_Float16)r0; pl[1][1] = (_Float16)r1;
  }
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// the power of two that brings a tensor whose largest |value| is `amax` (bit pattern) to [2^13, 2^14); exponents clamped so that the
// scale, the product of two scales and their reciprocals stay normal fp32 numbers (amax in [2^-37, 2^73]: outside, fp32 attention
// itself is degenerate)
__device__ __forceinline__ float pow2_scale(unsigned amax_bits) {
  unsigned e = (amax_bits >> 23) & 0xffu;
  e = e < 90u ? 90u : e > 200u ? 200u : e;
  return __builtin_bit_cast(float, (267u - e) << 23);
}

// FmtH2's softmax reference point.  The online softmax is exact for ANY per-row reference r >= max - 15.9 used consistently (P' =
// 2^(s - r), the running sum and output rescaled by 2^(r_old - r_new); the common factor cancels in O / l): r = max - 15 puts the
// largest P' at 2^15, so that fp16's 2^-14 subnormal threshold sits 29 octaves below the row maximum.  max - 15 is rounded (any
// rounding is fine, the SAME r goes into every exponent); where one ulp of the maximum exceeds 2^-5 the shift is dropped (such
// rows are one-hot up to factors 2^-k anyway).
__device__ __forceinline__ float h2_reference(float mx) { return fabsf(mx) < 262144.f ? mx - 15.f : mx; }

// all NT term products of one 16-deep k-step
template <class F>
__device__ __forceinline__ f32x16 mfma_terms(const typename F::x8 (&a)[F::NP], const typename F::x8 (&b)[F::NP], f32x16 c) {
#pragma unroll
  for (int i = 0; i < F::NT; ++i) c = F::mfma(a[F::pa(i)], b[F::pb(i)], c);
  return c;
}

// same XCD-aware work mapping as attention.hip (the query blocks of one (pair side, head) meet in one L2)
struct AttnBlock { int x, y, z; };
__device__ __forceinline__ AttnBlock attn_block() {
  const int nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
  const int L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const int w = (total & 7) == 0 ? (L & 7) * (total >> 3) + (L >> 3) : L;
  AttnBlock r;
  r.x = w % nx;
  r.y = (w / nx) % ny;
  r.z = w / (nx * ny);
  return r;
}
__device__ __forceinline__ float xhalf_max(float x) { return fmaxf(x, __shfl_xor(x, 32)); }
__device__ __forceinline__ float xhalf_sum(float x) { return x + __shfl_xor(x, 32); }

template <bool V>
struct BoolC { static constexpr bool value = V; };

template <int HD, class F>
__global__ __launch_bounds__(256, 2) void attention_x3_kernel(AttnArgs p, float scale) {
  typedef typename F::T E;
  typedef typename F::x8 x8;
  typedef typename F::x4 x4;
  typedef typename F::x2 x2;
  constexpr int NP = F::NP;
  constexpr int TK = 32;
  constexpr int KSB = HD + 8;         // K plane row stride (bf16): 80 / 144 bytes = 5 / 9 sixteen-byte slots (odd)
  constexpr int OB = HD / 32;         // output blocks of 32 dims
  constexpr int NS = HD / 16;         // 16-dim steps of K.Q^T
  constexpr int V4 = HD / 4;          // float4 per K / V row
  constexpr int ITER = (TK * V4) / 256;
  // separate objects per buffer: the stores of tile t+1 must be seen not to alias the loads of tile t (one scheduling region)
  __shared__ __attribute__((aligned(16))) E Kt0[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Kt1[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Vt0[NP][TK * HD];
  __shared__ __attribute__((aligned(16))) E Vt1[NP][TK * HD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const AttnBlock blk = attn_block();
  const int head = blk.y;
  const int side = blk.z / p.B, b = blk.z % p.B;
  const int kside = p.cross ? 1 - side : side;
  const int Nqp = side ? p.N1p : p.N0p, Nkp = kside ? p.N1p : p.N0p;
  const int q0 = blk.x * 128;
  if (q0 >= Nqp) return;
  const int nq = side ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const int nk = kside ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const size_t qbase = (side ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nqp;
  const size_t kbase = (kside ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nkp;
  const int ld = 3 * p.d;
  const int qrow = q0 + 32 * wave + l31;
  const bool wave_active = (q0 + 32 * wave) < Nqp;   // waves past the padded row range compute on a clamped row and store nothing

  // FmtH2: the powers of two that bring q, k, v into fp16's range (pow2_scale), and what undoes them: the scores come out of the
  // MFMAs multiplied by sq sk (one fma in the softmax), P carries 2^15 (inside its exponential; so does the row sum), O carries sv
  float sq = 1.f, sk = 1.f, sv = 1.f, cinv = 1.f, svinv = 1.f;
  if constexpr (F::SCALED) {
    sq = pow2_scale(p.amax[(side * p.B + b) * 4]);
    sk = pow2_scale(p.amax[(kside * p.B + b) * 4 + 1]);
    sv = pow2_scale(p.amax[(kside * p.B + b) * 4 + 2]);
    cinv = 1.0f / (sq * sk);
    svinv = 1.0f / sv;
  }
  // Q^T fragments (B operand): lane (query, kb = hi) holds dims 16 s + 8 hi .. + 7, pre-scaled (1/sqrt(HD) and log2 e), split
  x8 qf[NS][NP];
  {
    const float* qp = p.qkv + (qbase + min(qrow, Nqp - 1)) * ld + head * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s), c = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
      // (q scale) sq: the product with the power of two is exact, so both formats round q scale once, identically
      float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, c[0] * scale, c[1] * scale, c[2] * scale, c[3] * scale};
      if constexpr (F::SCALED) {        // rows past the keypoint count are not covered by amax: anything there could overflow fp16
        if (qrow >= nq) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        x2 pl[NP];
        if constexpr (F::SCALED) F::split(v[j] * sq, v[j + 1] * sq, pl);
        else F::split(v[j], v[j + 1], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) { qf[s][q][j] = pl[q][0]; qf[s][q][j + 1] = pl[q][1]; }
      }
    }
  }

  f32x16 O[OB], T[OB];      // running output, and the current 64-key group's partial product (two-level accumulation)
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) { O[o][r] = 0.f; T[o][r] = 0.f; }
  float m = -INFINITY, l = 0.f;

  // K/V rows of this (pair, side) through a buffer descriptor (rows past the padded count read as zeros, never used)
  const unsigned long long kaddr = (unsigned long long)(p.qkv + kbase * ld);
  const unsigned long long kaddr_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(kaddr >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((unsigned)kaddr);
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kaddr_u, 0, __builtin_amdgcn_readfirstlane(Nkp * ld * 4), 0x00020000);
  int kvo[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int e = tid + it * 256, key = e / V4, v4 = e % V4;
    kvo[it] = (key * ld + head * HD + 4 * v4 + p.d) * 4;
  }
  const int nt = (nk + TK - 1) / TK;
  f32x4 kreg0[ITER], vreg0[ITER], kreg1[ITER], vreg1[ITER];
  auto gload = [&](f32x4 (&kr)[ITER], f32x4 (&vr)[ITER], int kt) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(kt * TK * ld * 4);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      kr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it], so, 0));
      vr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it] + p.d * 4, so, 0));
    }
  };
  // split a staged tile into the planes
  // (`t`: the tile the registers hold.  FmtH2 zeroes the keys past the valid count: amax does not cover them, a value that
  // overflows fp16 would meet P = 0 as 0 x inf)
  auto lstore = [&](E (&Kd)[NP][TK * KSB], E (&Vd)[NP][TK * HD], const f32x4 (&kr)[ITER], const f32x4 (&vr)[ITER], int t)
      __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int e = tid + it * 256, key = e / V4, v4 = e % V4;
      x4 kp[NP], vp[NP];
      f32x4 kq = kr[it], vq = vr[it];
      if constexpr (F::SCALED) {
        // (block-uniform branch: only the tile that holds the end of the pair's keys pays the eight selects -- they were 5 % of the
        // kernel's VALU instructions in every tile)
        if (t * TK + TK > nk) {
          asm volatile("" ::: "memory");            // (not to be if-converted back into selects)
          if (t * TK + key >= nk) { kq = (f32x4){0.f, 0.f, 0.f, 0.f}; vq = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
        kq *= sk; vq *= sv;
      }
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        x2 pl[NP];
        F::split(kq[j], kq[j + 1], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) { kp[q][j] = pl[q][0]; kp[q][j + 1] = pl[q][1]; }
        F::split(vq[j], vq[j + 1], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) { vp[q][j] = pl[q][0]; vp[q][j + 1] = pl[q][1]; }
      }
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        *reinterpret_cast<x4*>(&Kd[pl][key * KSB + 4 * v4]) = kp[pl];
        *reinterpret_cast<x4*>(&Vd[pl][key * HD + 4 * v4]) = vp[pl];
      }
    }
  };

  // transposed-read address pattern of this lane inside a [4 keys][16 dims] block of a V plane
  const int tr_off = ((lane & 15) >> 2) * HD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  // one 32-key tile: S^T = K.Q^T, online softmax, O^T += V^T.P^T
  // `first`: this tile opens a 64-key group (folds the finished group T into O, then starts T from a zero accumulator)
  // `full`:  every key of the tile is valid (no masking)
  auto tile = [&](int kt, const E (&Kr)[NP][TK * KSB], const E (&Vr)[NP][TK * HD], auto first, auto full)
      __attribute__((always_inline)) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- S^T = K . Q^T
    f32x16 S = zero16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      x8 kf[NP];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) kf[pl] = *reinterpret_cast<const x8*>(&Kr[pl][l31 * KSB + 16 * s + 8 * hi]);
      S = mfma_terms<F>(kf, qf[s], S);
    }
    // ---- V^T fragments (A operand of the second product), requested before the softmax
    x8 vf[OB][2][NP];
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          const E* base = &Vr[pl][(16 * t + 4 * hi) * HD + 32 * o + tr_off];
          const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base));
          const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + 8 * HD));
          const u32x2 aw = __builtin_bit_cast(u32x2, a), cw = __builtin_bit_cast(u32x2, c);
          const u32x4 w = {aw[0], aw[1], cw[0], cw[1]};
          vf[o][t][pl] = __builtin_bit_cast(x8, w);
        }
    // ---- online softmax over this tile's 32 keys (16 here, 16 in lane^32), log2 domain
    float mx = -INFINITY;
    if constexpr (decltype(full)::value) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float sv = key < nk ? S[r] : -INFINITY;
        S[r] = sv;
        mx = fmaxf(mx, sv);
      }
    }
    mx = xhalf_max(mx);
    if constexpr (F::SCALED) mx = h2_reference(mx * cinv);   // the scores carry sq sk (a power of two: exact)
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);      // m = -inf on the first tile -> 0
    float rs = 0.f;
    x8 pf[2][NP];                                            // P^T fragments (B operand): step t holds S[8 t .. 8 t + 7]
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float p0, p1;
      if constexpr (F::SCALED) {
        p0 = __builtin_amdgcn_exp2f(fmaf(S[r], cinv, -mn));
        p1 = __builtin_amdgcn_exp2f(fmaf(S[r + 1], cinv, -mn));
      } else {
        p0 = __builtin_amdgcn_exp2f(S[r] - mn);
        p1 = __builtin_amdgcn_exp2f(S[r + 1] - mn);
      }
      rs += p0 + p1;
      x2 pl[NP];
      F::split(p0, p1, pl);
      const int t = r >> 3, j = r & 7;
#pragma unroll
      for (int q = 0; q < NP; ++q) { pf[t][q][j] = pl[q][0]; pf[t][q][j + 1] = pl[q][1]; }
    }
    rs = xhalf_sum(rs);
    l = l * alpha + rs;
    m = mn;
    // ---- (O^T + T) * alpha + V^T . P^T, two-level; the rescale is skipped when no lane of the wave saw a larger maximum
    const bool rescale = __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0;
    if constexpr (decltype(first)::value) {
#pragma unroll
      for (int o = 0; o < OB; ++o) O[o] += T[o];          // the finished group (zeros before the first one)
      if (rescale) {
#pragma unroll
        for (int o = 0; o < OB; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[o][r] *= alpha;
      }
#pragma unroll
      for (int o = 0; o < OB; ++o) T[o] = mfma_terms<F>(vf[o][1], pf[1], mfma_terms<F>(vf[o][0], pf[0], zero16));
    } else {
      if (rescale) {
#pragma unroll
        for (int o = 0; o < OB; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) { O[o][r] *= alpha; T[o][r] *= alpha; }
      }
#pragma unroll
      for (int o = 0; o < OB; ++o) T[o] = mfma_terms<F>(vf[o][1], pf[1], mfma_terms<F>(vf[o][0], pf[0], T[o]));
    }
  };
  // HD = 64 folds every 32-key tile (as attention.hip does: C5's 2048 keys)
  constexpr bool EVERY = HD >= 64;

  // tile kt+2 is requested while tile kt is multiplied and tile kt+1 (requested one iteration earlier) is split and stored:
  // two static register sets, two static LDS buffers, loop unrolled by two
  int th0 = 0, th1 = nt > 1 ? 1 : 0;               // the tiles the two register sets hold
  gload(kreg0, vreg0, th0);
  gload(kreg1, vreg1, th1);
  lstore(Kt0, Vt0, kreg0, vreg0, th0);
  __syncthreads();
  for (int kt = 0; kt < nt; kt += 2) {
    th0 = kt + 2 < nt ? kt + 2 : kt;
    gload(kreg0, vreg0, th0);
    if (kt * 32 + 32 <= nk) tile(kt, Kt0, Vt0, BoolC<true>{}, BoolC<true>{});          // block-uniform
    else tile(kt, Kt0, Vt0, BoolC<true>{}, BoolC<false>{});
    lstore(Kt1, Vt1, kreg1, vreg1, th1);           // tile kt+1
    __syncthreads();
    if (kt + 1 < nt) {                             // block-uniform
      th1 = kt + 3 < nt ? kt + 3 : kt;
      gload(kreg1, vreg1, th1);
      if (kt * 32 + 64 <= nk) tile(kt + 1, Kt1, Vt1, BoolC<EVERY>{}, BoolC<true>{});
      else tile(kt + 1, Kt1, Vt1, BoolC<EVERY>{}, BoolC<false>{});
      lstore(Kt0, Vt0, kreg0, vreg0, th0);         // tile kt+2
      __syncthreads();
    }
  }

  if (wave_active) {
#pragma unroll
    for (int o = 0; o < OB; ++o) O[o] += T[o];                    // the last group
    const float inv = (l > 0.f && qrow < nq) ? (1.0f / l) * svinv : 0.f;   // rows past the valid count: zeros (FmtH2: l carries 2^15 like O; O carries sv)
    float* op = p.out + (qbase + qrow) * p.d + head * HD;
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int g = 0; g < 4; ++g) {      // accumulator registers 4g..4g+3 = dims 8g + 4hi ..
        float4 v = make_float4(O[o][4 * g] * inv, O[o][4 * g + 1] * inv, O[o][4 * g + 2] * inv, O[o][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + o * 32 + 8 * g + 4 * hi) = v;
      }
  }
}
// attention_h2q2_kernel (round 6): attention_x3_kernel<32, FmtH2> with TWO 32-query blocks per wave -- a workgroup owns 256 queries, so
// every staged K / V tile (its split into fp16 planes is a sixth of a wave-tile's VALU instructions) and every K / V^T fragment read
// from LDS serves two query blocks.  The second block's O / T / m / l cost 34 registers and its Q fragments 16 more, which the
// straight two-block form does not have (26 spilled registers, reloads inside the tile loop: measured in the compiler, not on the GPU);
// here the Q fragments of both blocks live in LDS (32 KB per workgroup, wave-private: no barrier) and are read per tile, 16 registers
// at a time.  Per query the instructions and their order are attention_x3_kernel's: bit-identical output (tools/ubench/attn_x3_bench
// QB=1 / QB=2 dumps, tests/test_gpu_superglue.py).  Same box: 300 -> 281 us per launch, 5.83 -> 5.49 ms per C3 step.
__global__ __launch_bounds__(256, 2) void attention_h2q2_kernel(AttnArgs p, float scale) {
  typedef FmtH2 F;
  typedef F::T E;
  typedef F::x8 x8;
  typedef F::x4 x4;
  typedef F::x2 x2;
  constexpr int HD = 32, NP = 2, TK = 32, KSB = HD + 8, NS = 2, V4 = HD / 4, QB = 2;
  __shared__ __attribute__((aligned(16))) E Kt0[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Kt1[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Vt0[NP][TK * HD];
  __shared__ __attribute__((aligned(16))) E Vt1[NP][TK * HD];
  __shared__ __attribute__((aligned(16))) E Qs[4][QB][NS][NP][64 * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const AttnBlock blk = attn_block();
  const int head = blk.y;
  const int side = blk.z / p.B, b = blk.z % p.B;
  const int kside = p.cross ? 1 - side : side;
  const int Nqp = side ? p.N1p : p.N0p, Nkp = kside ? p.N1p : p.N0p;
  const int q0 = blk.x * 128 * QB;
  if (q0 >= Nqp) return;
  const int nq = side ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const int nk = kside ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const size_t qbase = (side ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nqp;
  const size_t kbase = (kside ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nkp;
  const int ld = 3 * p.d;
  const int qrow0 = q0 + 32 * QB * wave + l31;            // query block jq of the wave: rows qrow0 + 32 jq

  const float sq = pow2_scale(p.amax[(side * p.B + b) * 4]);
  const float sk = pow2_scale(p.amax[(kside * p.B + b) * 4 + 1]);
  const float sv = pow2_scale(p.amax[(kside * p.B + b) * 4 + 2]);
  const float cinv = 1.0f / (sq * sk), svinv = 1.0f / sv;
  // Q^T fragments (B operand) of both blocks, split as in attention_x3_kernel, into this wave's LDS region in fragment order
#pragma unroll
  for (int jq = 0; jq < QB; ++jq) {
    const int qrow = qrow0 + 32 * jq;
    const float* qp = p.qkv + (qbase + min(qrow, Nqp - 1)) * ld + head * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s), c = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
      float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, c[0] * scale, c[1] * scale, c[2] * scale, c[3] * scale};
      if (qrow >= nq) {               // rows past the keypoint count are not covered by amax: anything there could overflow fp16
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      x8 qv[NP];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        x2 pl[NP];
        F::split(v[j] * sq, v[j + 1] * sq, pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) { qv[q][j] = pl[q][0]; qv[q][j + 1] = pl[q][1]; }
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<x8*>(&Qs[wave][jq][s][q][lane * 8]) = qv[q];
    }
  }

  f32x16 O[QB], T[QB];       // running output, and the current 64-key group's partial product (two-level accumulation)
  float m[QB], l[QB];
#pragma unroll
  for (int jq = 0; jq < QB; ++jq) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { O[jq][r] = 0.f; T[jq][r] = 0.f; }
    m[jq] = -INFINITY;
    l[jq] = 0.f;
  }

  // K/V rows of this (pair, side) through a buffer descriptor (attention_x3_kernel)
  const unsigned long long kaddr = (unsigned long long)(p.qkv + kbase * ld);
  const unsigned long long kaddr_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(kaddr >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((unsigned)kaddr);
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kaddr_u, 0, __builtin_amdgcn_readfirstlane(Nkp * ld * 4), 0x00020000);
  const int skey = tid / V4, sv4 = tid % V4;            // this thread's float4 of a staged tile: key, four dims
  const int kvo = (skey * ld + head * HD + 4 * sv4 + p.d) * 4;
  const int nt = (nk + TK - 1) / TK;
  f32x4 kreg0, vreg0, kreg1, vreg1;
  auto gload = [&](f32x4& kr, f32x4& vr, int kt) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(kt * TK * ld * 4);
    kr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo, so, 0));
    vr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo + p.d * 4, so, 0));
  };
  auto lstore = [&](E (&Kd)[NP][TK * KSB], E (&Vd)[NP][TK * HD], const f32x4& kr, const f32x4& vr, int t) __attribute__((always_inline)) {
    x4 kp[NP], vp[NP];
    f32x4 kq = kr, vq = vr;
    if (t * TK + TK > nk) {           // block-uniform: the tile that holds the end of the keys (amax does not cover what lies past them)
      asm volatile("" ::: "memory");
      if (t * TK + skey >= nk) { kq = (f32x4){0.f, 0.f, 0.f, 0.f}; vq = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    kq *= sk; vq *= sv;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      x2 pl[NP];
      F::split(kq[j], kq[j + 1], pl);
#pragma unroll
      for (int q = 0; q < NP; ++q) { kp[q][j] = pl[q][0]; kp[q][j + 1] = pl[q][1]; }
      F::split(vq[j], vq[j + 1], pl);
#pragma unroll
      for (int q = 0; q < NP; ++q) { vp[q][j] = pl[q][0]; vp[q][j + 1] = pl[q][1]; }
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      *reinterpret_cast<x4*>(&Kd[pl][skey * KSB + 4 * sv4]) = kp[pl];
      *reinterpret_cast<x4*>(&Vd[pl][skey * HD + 4 * sv4]) = vp[pl];
    }
  };
  const int tr_off = ((lane & 15) >> 2) * HD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  auto tile = [&](int kt, const E (&Kr)[NP][TK * KSB], const E (&Vr)[NP][TK * HD], auto first, auto full) __attribute__((always_inline)) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // K and V^T fragments: read once, they meet both query blocks
    x8 kf[NS][NP], vf[2][NP];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) kf[s][pl] = *reinterpret_cast<const x8*>(&Kr[pl][l31 * KSB + 16 * s + 8 * hi]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const E* base = &Vr[pl][(16 * t + 4 * hi) * HD + tr_off];
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base));
        const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + 8 * HD));
        const u32x2 aw = __builtin_bit_cast(u32x2, a), cw = __builtin_bit_cast(u32x2, c);
        const u32x4 w = {aw[0], aw[1], cw[0], cw[1]};
        vf[t][pl] = __builtin_bit_cast(x8, w);
      }
#pragma unroll
    for (int jq = 0; jq < QB; ++jq) {
      __builtin_amdgcn_sched_barrier(0);                      // (one query block at a time)
      x8 qf[NS][NP];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) qf[s][pl] = *reinterpret_cast<const x8*>(&Qs[wave][jq][s][pl][lane * 8]);
      f32x16 S = zero16;
#pragma unroll
      for (int s = 0; s < NS; ++s) S = mfma_terms<F>(kf[s], qf[s], S);
      float mx = -INFINITY;
      if constexpr (decltype(full)::value) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float svv = key < nk ? S[r] : -INFINITY;
          S[r] = svv;
          mx = fmaxf(mx, svv);
        }
      }
      mx = xhalf_max(mx);
      mx = h2_reference(mx * cinv);
      const float mn = fmaxf(m[jq], mx);
      const float alpha = __builtin_amdgcn_exp2f(m[jq] - mn);
      float rs = 0.f;
      x8 pf[2][NP];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(S[r], cinv, -mn));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(S[r + 1], cinv, -mn));
        rs += p0 + p1;
        x2 pl[NP];
        F::split(p0, p1, pl);
        const int t = r >> 3, j = r & 7;
#pragma unroll
        for (int q = 0; q < NP; ++q) { pf[t][q][j] = pl[q][0]; pf[t][q][j + 1] = pl[q][1]; }
      }
      rs = xhalf_sum(rs);
      l[jq] = l[jq] * alpha + rs;
      m[jq] = mn;
      const bool rescale = __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0;
      if constexpr (decltype(first)::value) {
        O[jq] += T[jq];
        if (rescale) {
#pragma unroll
          for (int r = 0; r < 16; ++r) O[jq][r] *= alpha;
        }
        T[jq] = mfma_terms<F>(vf[1], pf[1], mfma_terms<F>(vf[0], pf[0], zero16));
      } else {
        if (rescale) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { O[jq][r] *= alpha; T[jq][r] *= alpha; }
        }
        T[jq] = mfma_terms<F>(vf[1], pf[1], mfma_terms<F>(vf[0], pf[0], T[jq]));
      }
    }
  };

  int th0 = 0, th1 = nt > 1 ? 1 : 0;
  gload(kreg0, vreg0, th0);
  gload(kreg1, vreg1, th1);
  lstore(Kt0, Vt0, kreg0, vreg0, th0);
  __syncthreads();
  for (int kt = 0; kt < nt; kt += 2) {
    th0 = kt + 2 < nt ? kt + 2 : kt;
    gload(kreg0, vreg0, th0);
    if (kt * 32 + 32 <= nk) tile(kt, Kt0, Vt0, BoolC<true>{}, BoolC<true>{});          // block-uniform
    else tile(kt, Kt0, Vt0, BoolC<true>{}, BoolC<false>{});
    lstore(Kt1, Vt1, kreg1, vreg1, th1);
    __syncthreads();
    if (kt + 1 < nt) {
      th1 = kt + 3 < nt ? kt + 3 : kt;
      gload(kreg1, vreg1, th1);
      if (kt * 32 + 64 <= nk) tile(kt + 1, Kt1, Vt1, BoolC<false>{}, BoolC<true>{});
      else tile(kt + 1, Kt1, Vt1, BoolC<false>{}, BoolC<false>{});
      lstore(Kt0, Vt0, kreg0, vreg0, th0);
      __syncthreads();
    }
  }
#pragma unroll
  for (int jq = 0; jq < QB; ++jq) {
    if (q0 + 32 * QB * wave + 32 * jq >= Nqp) break;                  // (wave-uniform: blocks past the padded rows store nothing)
    const int qrow = qrow0 + 32 * jq;
    O[jq] += T[jq];
    const float inv = (l[jq] > 0.f && qrow < nq) ? (1.0f / l[jq]) * svinv : 0.f;
    float* op = p.out + (qbase + qrow) * p.d + head * HD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(O[jq][4 * g] * inv, O[jq][4 * g + 1] * inv, O[jq][4 * g + 2] * inv, O[jq][4 * g + 3] * inv);
      *reinterpret_cast<float4*>(op + 8 * g + 4 * hi) = v;
    }
  }
}

// The pipelined form (used for HD = 64: the straight form above needs 2 x the V / P / O registers there and spills).
template <int HD, class F>
__global__ __launch_bounds__(256, 2) void attention_x3p_kernel(AttnArgs p, float scale) {
  typedef typename F::T E;
  typedef typename F::x8 x8;
  typedef typename F::x4 x4;
  typedef typename F::x2 x2;
  constexpr int NP = F::NP, NT = F::NT;
  constexpr int TK = 32;
  constexpr int KSB = HD + 8;         // K plane row stride (bf16): 80 / 144 bytes = 5 / 9 sixteen-byte slots (odd)
  constexpr int OB = HD / 32;         // output blocks of 32 dims
  constexpr int NS = HD / 16;         // 16-dim steps of K.Q^T
  constexpr int V4 = HD / 4;          // float4 per K / V row
  constexpr int ITER = (TK * V4) / 256;
  // four buffers, separate objects: in one iteration tile k+1's K and tile k-1's V are read while tile k+2 is written, and the
  // compiler must see that those never alias (one scheduling region)
  __shared__ __attribute__((aligned(16))) E Kt0[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Kt1[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Kt2[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Kt3[NP][TK * KSB];
  __shared__ __attribute__((aligned(16))) E Vt0[NP][TK * HD];
  __shared__ __attribute__((aligned(16))) E Vt1[NP][TK * HD];
  __shared__ __attribute__((aligned(16))) E Vt2[NP][TK * HD];
  __shared__ __attribute__((aligned(16))) E Vt3[NP][TK * HD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const AttnBlock blk = attn_block();
  const int head = blk.y;
  const int side = blk.z / p.B, b = blk.z % p.B;
  const int kside = p.cross ? 1 - side : side;
  const int Nqp = side ? p.N1p : p.N0p, Nkp = kside ? p.N1p : p.N0p;
  const int q0 = blk.x * 128;
  if (q0 >= Nqp) return;
  const int nq = side ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const int nk = kside ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const size_t qbase = (side ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nqp;
  const size_t kbase = (kside ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nkp;
  const int ld = 3 * p.d;
  const int qrow = q0 + 32 * wave + l31;
  const bool wave_active = (q0 + 32 * wave) < Nqp;   // waves past the padded row range compute on a clamped row and store nothing

  float sq = 1.f, sk = 1.f, sv = 1.f, cinv = 1.f, svinv = 1.f;      // FmtH2's powers of two: see attention_x3_kernel
  if constexpr (F::SCALED) {
    sq = pow2_scale(p.amax[(side * p.B + b) * 4]);
    sk = pow2_scale(p.amax[(kside * p.B + b) * 4 + 1]);
    sv = pow2_scale(p.amax[(kside * p.B + b) * 4 + 2]);
    cinv = 1.0f / (sq * sk);
    svinv = 1.0f / sv;
  }
  // Q^T fragments (B operand): lane (query, kb = hi) holds dims 16 s + 8 hi .. + 7, pre-scaled (1/sqrt(HD) and log2 e), split
  x8 qf[NS][NP];
  {
    const float* qp = p.qkv + (qbase + min(qrow, Nqp - 1)) * ld + head * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s), c = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
      float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, c[0] * scale, c[1] * scale, c[2] * scale, c[3] * scale};
      if constexpr (F::SCALED) {        // rows past the keypoint count are not covered by amax
        if (qrow >= nq) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        x2 pl[NP];
        if constexpr (F::SCALED) F::split(v[j] * sq, v[j + 1] * sq, pl);
        else F::split(v[j], v[j + 1], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) { qf[s][q][j] = pl[q][0]; qf[s][q][j + 1] = pl[q][1]; }
      }
    }
  }

  // K/V rows of this (pair, side) through a buffer descriptor (rows past the padded count read as zeros, never used)
  const unsigned long long kaddr = (unsigned long long)(p.qkv + kbase * ld);
  const unsigned long long kaddr_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(kaddr >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((unsigned)kaddr);
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kaddr_u, 0, __builtin_amdgcn_readfirstlane(Nkp * ld * 4), 0x00020000);
  int kvo[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int e = tid + it * 256, key = e / V4, v4 = e % V4;
    kvo[it] = (key * ld + head * HD + 4 * v4 + p.d) * 4;
  }
  const int nt = (nk + TK - 1) / TK;
  f32x4 kreg[ITER], vreg[ITER];
  int tload = 0;                                   // the tile kreg / vreg hold
  auto gload = [&](int kt) __attribute__((always_inline)) {
    tload = kt;
    const int so = __builtin_amdgcn_readfirstlane(kt * TK * ld * 4);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      kreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it], so, 0));
      vreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it] + p.d * 4, so, 0));
    }
  };
  // split the staged tile into the planes
  auto lstore = [&](E (&Kd)[NP][TK * KSB], E (&Vd)[NP][TK * HD]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int e = tid + it * 256, key = e / V4, v4 = e % V4;
      x4 kp[NP], vp[NP];
      f32x4 kq = kreg[it], vq = vreg[it];
      if constexpr (F::SCALED) {        // keys past the valid count are zeroed (amax does not cover them: 0 x inf), the rest scaled
        if (tload * TK + TK > nk) {     // (block-uniform: see attention_x3_kernel)
          asm volatile("" ::: "memory");
          if (tload * TK + key >= nk) { kq = (f32x4){0.f, 0.f, 0.f, 0.f}; vq = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
        kq *= sk; vq *= sv;
      }
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        x2 pl[NP];
        F::split(kq[j], kq[j + 1], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) { kp[q][j] = pl[q][0]; kp[q][j + 1] = pl[q][1]; }
        F::split(vq[j], vq[j + 1], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) { vp[q][j] = pl[q][0]; vp[q][j + 1] = pl[q][1]; }
      }
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        *reinterpret_cast<x4*>(&Kd[pl][key * KSB + 4 * v4]) = kp[pl];
        *reinterpret_cast<x4*>(&Vd[pl][key * HD + 4 * v4]) = vp[pl];
      }
    }
  };

  // transposed-read address pattern of this lane inside a [4 keys][16 dims] block of a V plane
  const int tr_off = ((lane & 15) >> 2) * HD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  f32x16 O[OB], Tq[OB], S;  // running output; the product of tile k-2 (folded in this iteration); the scores of tile k
  x8 pf[2][NP];             // P^T fragments of tile k-1 (B operand): step t holds keys r = 8 t .. 8 t + 7 of the lane's sixteen
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) { O[o][r] = 0.f; Tq[o][r] = 0.f; }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[t][pl][j] = (E)0.f;
  float m = -INFINITY, l = 0.f, a1 = 1.f, a2 = 1.f;     // a1 / a2: the rescale factors that go with the products of tiles k-1 / k-2

  // One iteration = tile k, k = 0 .. nt (the last one only drains).  A three-deep software pipeline inside the wave, so that the
  // MFMAs of an iteration never wait for its VALU work and the two MFMA chains alternate (a dependent bf16 MFMA issued right
  // behind its producer, with other instructions in between, stalls the pipe):
  //   MFMA:  scores of tile k+1 (K.Q^T)  interleaved with  the product of tile k-1 (V^T.P^T, from a zero accumulator)
  //   VALU:  fold the product of tile k-2 into the output (O = O * alpha + T: two-level accumulation at every tile, one fma per
  //          element), softmax + split of tile k's scores, split + store of tile k+2 (requested one iteration earlier)
  // one barrier per tile, no branch in the body except the key mask of the last tile.
  auto body = [&](int k, const E (&Kn)[NP][TK * KSB], const E (&Vp)[NP][TK * HD], E (&Kd)[NP][TK * KSB],
                  E (&Vd)[NP][TK * HD]) __attribute__((always_inline)) {
    // ---- operands of this iteration's MFMAs: K of tile k+1 (A), Q (B, registers); V^T of tile k-1 (A), P^T of tile k-1 (B)
    x8 kf[NS][NP], vf[OB][2][NP];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) kf[s][pl] = *reinterpret_cast<const x8*>(&Kn[pl][l31 * KSB + 16 * s + 8 * hi]);
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          const E* base = &Vp[pl][(16 * t + 4 * hi) * HD + 32 * o + tr_off];
          const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base));
          const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + 8 * HD));
          const u32x2 aw = __builtin_bit_cast(u32x2, a), cw = __builtin_bit_cast(u32x2, c);
          const u32x4 w = {aw[0], aw[1], cw[0], cw[1]};
          vf[o][t][pl] = __builtin_bit_cast(x8, w);
        }
    f32x16 Sn = zero16, Tn[OB];
#pragma unroll
    for (int o = 0; o < OB; ++o) Tn[o] = zero16;
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
      if (i < NT * NS) Sn = F::mfma(kf[i / NT][F::pa(i % NT)], qf[i / NT][F::pb(i % NT)], Sn);
#pragma unroll
      for (int o = 0; o < OB; ++o) Tn[o] = F::mfma(vf[o][i / NT][F::pa(i % NT)], pf[i / NT][F::pb(i % NT)], Tn[o]);
    }
    if constexpr (NS > 2) {
#pragma unroll
      for (int i = 2 * NT; i < NT * NS; ++i) Sn = F::mfma(kf[i / NT][F::pa(i % NT)], qf[i / NT][F::pb(i % NT)], Sn);
    }
    // ---- fold the product of tile k-2: O = O * a2 + Tq
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[o][r] = fmaf(O[o][r], a2, Tq[o][r]);
    // ---- online softmax over tile k's 32 keys (16 here, 16 in lane^32), log2 domain
    // This branch also keeps the body in TWO scheduling regions -- the MFMAs clustered in the first, the softmax / split VALU work
    // in the second.  Round 4 measured the alternative (branch outside the body, one region: hipcc then interleaves one MFMA with
    // 7-8 VALU instructions, also when pinned with sched_group_barrier): 3-9 % SLOWER at HD = 32 and HD = 64.
    if (k * 32 + 32 > nk) {               // block-uniform: the last tile (or the draining iteration, whose scores are not used)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = (k * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi < nk) ? S[r] : -INFINITY;
    }
    const bool live = k < nt;             // block-uniform; the draining iteration leaves m and l alone
    float mx = S[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
    mx = xhalf_max(mx);
    if constexpr (F::SCALED) mx = h2_reference(mx * cinv);                // the scores carry sq sk (a power of two: exact)
    const float mn = live ? fmaxf(m, mx) : m;
    const float alpha = live ? __builtin_amdgcn_exp2f(m - mn) : 1.f;      // m = -inf on the first tile -> 0
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 mn2 = {mn, mn}, cinv2 = {cinv, cinv};
    f32x2 rs2 = {0.f, 0.f};
    x8 pn[2][NP];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f32x2 dd;
      if constexpr (F::SCALED) dd = __builtin_elementwise_fma((f32x2){S[r], S[r + 1]}, cinv2, -mn2);
      else dd = (f32x2){S[r], S[r + 1]} - mn2;
      const f32x2 pp = {__builtin_amdgcn_exp2f(dd[0]), __builtin_amdgcn_exp2f(dd[1])};
      rs2 += pp;
      x2 pl[NP];
      F::split(pp[0], pp[1], pl);
      const int t = r >> 3, j = r & 7;
#pragma unroll
      for (int q = 0; q < NP; ++q) { pn[t][q][j] = pl[q][0]; pn[t][q][j + 1] = pl[q][1]; }
    }
    const float rs = xhalf_sum(rs2[0] + rs2[1]);
    l = live ? l * alpha + rs : l;
    m = mn;
    // ---- tile k+2: split, store; request tile k+3
    lstore(Kd, Vd);
    gload(min(k + 3, nt - 1));
    // ---- shift the pipeline
    a2 = a1;
    a1 = alpha;
    S = Sn;
#pragma unroll
    for (int o = 0; o < OB; ++o) Tq[o] = Tn[o];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) pf[t][pl] = pn[t][pl];
    __syncthreads();
  };

  gload(0);
  lstore(Kt0, Vt0);
  gload(nt > 1 ? 1 : 0);
  lstore(Kt1, Vt1);
  gload(nt > 2 ? 2 : 0);
  // iteration 0 multiplies P = 0 by the V buffer of "tile -1": it must hold finite values
  for (int e = tid; e < NP * TK * HD / 8; e += 256) reinterpret_cast<u32x4*>(&Vt3[0][0])[e] = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();
  {                                       // scores of tile 0
    S = zero16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      x8 kf[NP];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) kf[pl] = *reinterpret_cast<const x8*>(&Kt0[pl][l31 * KSB + 16 * s + 8 * hi]);
      S = mfma_terms<F>(kf, qf[s], S);
    }
  }
  // iteration k reads K buffer (k+1) % 4 and V buffer (k-1) % 4 (zeros x anything at k = 0: P is zero) and writes buffer (k+2) % 4
  for (int k = 0; k <= nt; k += 4) {
    body(k, Kt1, Vt3, Kt2, Vt2);
    if (k + 1 <= nt) body(k + 1, Kt2, Vt0, Kt3, Vt3);       // block-uniform
    if (k + 2 <= nt) body(k + 2, Kt3, Vt1, Kt0, Vt0);
    if (k + 3 <= nt) body(k + 3, Kt0, Vt2, Kt1, Vt1);
  }

  if (wave_active) {
    const float inv = (l > 0.f && qrow < nq) ? (1.0f / l) * svinv : 0.f;   // rows past the valid count: zeros
    float* op = p.out + (qbase + qrow) * p.d + head * HD;
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int g = 0; g < 4; ++g) {      // accumulator registers 4g..4g+3 = dims 8g + 4hi ..; the last tile's product is folded here
        float4 v = make_float4(fmaf(O[o][4 * g], a2, Tq[o][4 * g]) * inv, fmaf(O[o][4 * g + 1], a2, Tq[o][4 * g + 1]) * inv,
                               fmaf(O[o][4 * g + 2], a2, Tq[o][4 * g + 2]) * inv, fmaf(O[o][4 * g + 3], a2, Tq[o][4 * g + 3]) * inv);
        *reinterpret_cast<float4*>(op + o * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

// max |q|, |k|, |v| over the VALID rows of every (side, pair) (FmtH2's scales).  grid (chunks, 2 B), 3 d / 4 threads: a thread owns ONE float4
// column of the rows chunk, chunk + chunks, ..., so a half-wave (d = 128) or a wave (d = 256) stays inside one third of the row
// (q | k | v); the maxima go out as bit patterns through atomicMax (|x| orders as an unsigned integer; the result does not depend
// on the order).  Rows past a pair's keypoint count never reach the attention kernels' softmax and are not looked at here either.
__global__ __launch_bounds__(192) void qkv_amax_kernel(AttnArgs p, unsigned* amax) {
  const int side = blockIdx.y / p.B, b = blockIdx.y % p.B;
  const int Np = side ? p.N1p : p.N0p;
  const int n = side ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const size_t base = (side ? (size_t)p.B * p.N0p : 0) + (size_t)b * Np;
  const int ld4 = 3 * p.d / 4, c = threadIdx.x;                     // 96 / 192 float4 per row
  const u32x4* src = reinterpret_cast<const u32x4*>(p.qkv) + base * ld4 + c;
  unsigned mx = 0;
#pragma unroll 4
  for (int r = blockIdx.x; r < n; r += gridDim.x) {
    const u32x4 v = src[(size_t)r * ld4];
    const unsigned m01 = max(v[0] & 0x7fffffffu, v[1] & 0x7fffffffu), m23 = max(v[2] & 0x7fffffffu, v[3] & 0x7fffffffu);
    mx = max(mx, max(m01, m23));
  }
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if ((threadIdx.x & 31) == 0 && mx) atomicMax(amax + blockIdx.y * 4 + c / (ld4 / 3), mx);
}
}  // namespace

bool attention_x3_supported(const AttnArgs& a) {
  const int hd = a.heads > 0 ? a.d / a.heads : 0;
  return hd == 32 || hd == 64;
}

// amax: [2 B][4] words, zeroed by the caller (once per forward: one table per layer)
hipError_t launch_qkv_amax(const AttnArgs& a, unsigned* amax, hipStream_t s) {
  if (a.d != 128 && a.d != 256) return hipErrorInvalidValue;      // a half-wave must stay inside one third of a row
  const int nmax = a.N0p > a.N1p ? a.N0p : a.N1p;
  const int chunks = nmax >= 512 ? 16 : nmax >= 64 ? 4 : 1;
  hipLaunchKernelGGL(qkv_amax_kernel, dim3((unsigned)chunks, (unsigned)(2 * a.B)), dim3((unsigned)(3 * a.d / 4)), 0, s, a, amax);
  return hipGetLastError();
}

hipError_t launch_attention_x3(const AttnArgs& a, hipStream_t s) {
  if (!attention_x3_supported(a)) return hipErrorInvalidValue;
  const int hd = a.d / a.heads;
  const int nmax = a.N0p > a.N1p ? a.N0p : a.N1p;
  dim3 grid((unsigned)((nmax + 127) / 128), (unsigned)a.heads, (unsigned)(2 * a.B));
  const float scale = (float)(1.4426950408889634 / sqrt((double)hd));   // log2(e)/sqrt(HD)
  if (a.amax) {                   // two fp16 planes, three term products per k-step; needs the launch's q / k / v maxima
    last_form = "attention_h2:f16x2";
    // two query blocks per wave: whole 256-query workgroups only; "auto" where one block per wave would still leave two workgroups
    // per slot of the chip (at one or two pairs the smaller workgroups fill more of it)
    const bool q2 = a.qblocks == 2 || (a.qblocks <= 0 && (long)((nmax + 127) / 128) * a.heads * 2 * a.B >= 1024);
    if (hd == 32 && q2 && nmax % 256 == 0) {
      dim3 grid2((unsigned)(nmax / 256), (unsigned)a.heads, (unsigned)(2 * a.B));
      hipLaunchKernelGGL(attention_h2q2_kernel, grid2, dim3(256), 0, s, a, scale);
    } else if (hd == 32) hipLaunchKernelGGL((attention_x3_kernel<32, FmtH2>), grid, dim3(256), 0, s, a, scale);
    else hipLaunchKernelGGL((attention_x3p_kernel<64, FmtH2>), grid, dim3(256), 0, s, a, scale);
    return hipGetLastError();
  }
  last_form = "attention_x3:bf16x3";
  if (hd == 32) hipLaunchKernelGGL((attention_x3_kernel<32, FmtX3>), grid, dim3(256), 0, s, a, scale);
  else hipLaunchKernelGGL((attention_x3p_kernel<64, FmtX3>), grid, dim3(256), 0, s, a, scale);
  return hipGetLastError();
}

}  // namespace imx
