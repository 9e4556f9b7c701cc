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

namespace {

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

// x = h + m + l, two values at a time (split3.h: the conversions round to nearest even and pack, the residuals are one
// v_dot2c_f32_bf16 each)
__device__ __forceinline__ void split2(float x0, float x1, bf16x2& h, bf16x2& m, bf16x2& l) { split3_pair(x0, x1, h, m, l); }

// six term products, smallest first: planes (A, B) = (m,m) (h,l) (l,h) (h,m) (m,h) (h,h)
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attention_x3_kernel(AttnArgs p, float scale) {
  constexpr int TK = 32;
  constexpr int KSB = HD + 8;         // K plane row stride (bf16): 80 / 144 bytes = 5 / 9 sixteen-byte slots (odd)
  constexpr int OB = HD / 32;         // output blocks of 32 dims
  constexpr int NS = HD / 16;         // 16-dim steps of K.Q^T
  constexpr int V4 = HD / 4;          // float4 per K / V row
  constexpr int ITER = (TK * V4) / 256;
  // separate objects per buffer: the stores of tile t+1 must be seen not to alias the loads of tile t (one scheduling region)
  __shared__ __attribute__((aligned(16))) __bf16 Kt0[3][TK * KSB];
  __shared__ __attribute__((aligned(16))) __bf16 Kt1[3][TK * KSB];
  __shared__ __attribute__((aligned(16))) __bf16 Vt0[3][TK * HD];
  __shared__ __attribute__((aligned(16))) __bf16 Vt1[3][TK * HD];

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

  // Q^T fragments (B operand): lane (query, kb = hi) holds dims 16 s + 8 hi .. + 7, pre-scaled (1/sqrt(HD) and log2 e), split
  bf16x8 qf[NS][3];
  {
    const float* qp = p.qkv + (qbase + min(qrow, Nqp - 1)) * ld + head * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s), c = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
      const float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, c[0] * scale, c[1] * scale, c[2] * scale, c[3] * scale};
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        bf16x2 h, m, l;
        split2(v[j], v[j + 1], h, m, l);
        qf[s][0][j] = h[0]; qf[s][0][j + 1] = h[1];
        qf[s][1][j] = m[0]; qf[s][1][j + 1] = m[1];
        qf[s][2][j] = l[0]; qf[s][2][j + 1] = l[1];
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
  // split a staged tile into the three planes
  auto lstore = [&](__bf16 (&Kd)[3][TK * KSB], __bf16 (&Vd)[3][TK * HD], const f32x4 (&kr)[ITER], const f32x4 (&vr)[ITER])
      __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int e = tid + it * 256, key = e / V4, v4 = e % V4;
      bf16x4 kp[3], vp[3];
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        bf16x2 h, mm, ll;
        split2(kr[it][j], kr[it][j + 1], h, mm, ll);
        kp[0][j] = h[0]; kp[0][j + 1] = h[1]; kp[1][j] = mm[0]; kp[1][j + 1] = mm[1]; kp[2][j] = ll[0]; kp[2][j + 1] = ll[1];
        split2(vr[it][j], vr[it][j + 1], h, mm, ll);
        vp[0][j] = h[0]; vp[0][j + 1] = h[1]; vp[1][j] = mm[0]; vp[1][j + 1] = mm[1]; vp[2][j] = ll[0]; vp[2][j + 1] = ll[1];
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        *reinterpret_cast<bf16x4*>(&Kd[pl][key * KSB + 4 * v4]) = kp[pl];
        *reinterpret_cast<bf16x4*>(&Vd[pl][key * HD + 4 * v4]) = vp[pl];
      }
    }
  };

  // transposed-read address pattern of this lane inside a [4 keys][16 dims] block of a V plane
  const int tr_off = ((lane & 15) >> 2) * HD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  // one 32-key tile: S^T = K.Q^T, online softmax, O^T += V^T.P^T
  // `first`: this tile opens a 64-key group (folds the finished group T into O, then starts T from a zero accumulator)
  // `full`:  every key of the tile is valid (no masking)
  auto tile = [&](int kt, const __bf16 (&Kr)[3][TK * KSB], const __bf16 (&Vr)[3][TK * HD], auto first, auto full)
      __attribute__((always_inline)) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- S^T = K . Q^T
    f32x16 S = zero16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x8 kf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(&Kr[pl][l31 * KSB + 16 * s + 8 * hi]);
      S = mfma6(kf, qf[s], S);
    }
    // ---- V^T fragments (A operand of the second product), requested before the softmax
    bf16x8 vf[OB][2][3];
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const __bf16* base = &Vr[pl][(16 * t + 4 * hi) * HD + 32 * o + tr_off];
          const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base));
          const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + 8 * HD));
          const u32x2 aw = __builtin_bit_cast(u32x2, a), cw = __builtin_bit_cast(u32x2, c);
          const u32x4 w = {aw[0], aw[1], cw[0], cw[1]};
          vf[o][t][pl] = __builtin_bit_cast(bf16x8, w);
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
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);      // m = -inf on the first tile -> 0
    float rs = 0.f;
    bf16x8 pf[2][3];                                         // P^T fragments (B operand): step t holds S[8 t .. 8 t + 7]
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __builtin_amdgcn_exp2f(S[r] - mn), p1 = __builtin_amdgcn_exp2f(S[r + 1] - mn);
      rs += p0 + p1;
      bf16x2 h, mm, ll;
      split2(p0, p1, h, mm, ll);
      const int t = r >> 3, j = r & 7;
      pf[t][0][j] = h[0]; pf[t][0][j + 1] = h[1];
      pf[t][1][j] = mm[0]; pf[t][1][j + 1] = mm[1];
      pf[t][2][j] = ll[0]; pf[t][2][j + 1] = ll[1];
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
      for (int o = 0; o < OB; ++o) T[o] = mfma6(vf[o][1], pf[1], mfma6(vf[o][0], pf[0], zero16));
    } else {
      if (rescale) {
#pragma unroll
        for (int o = 0; o < OB; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) { O[o][r] *= alpha; T[o][r] *= alpha; }
      }
#pragma unroll
      for (int o = 0; o < OB; ++o) T[o] = mfma6(vf[o][1], pf[1], mfma6(vf[o][0], pf[0], T[o]));
    }
  };
  // HD = 64 folds every 32-key tile (as attention.hip does: C5's 2048 keys)
  constexpr bool EVERY = HD >= 64;

  // tile kt+2 is requested while tile kt is multiplied and tile kt+1 (requested one iteration earlier) is split and stored:
  // two static register sets, two static LDS buffers, loop unrolled by two
  gload(kreg0, vreg0, 0);
  gload(kreg1, vreg1, nt > 1 ? 1 : 0);
  lstore(Kt0, Vt0, kreg0, vreg0);
  __syncthreads();
  for (int kt = 0; kt < nt; kt += 2) {
    gload(kreg0, vreg0, kt + 2 < nt ? kt + 2 : kt);
    if (kt * 32 + 32 <= nk) tile(kt, Kt0, Vt0, BoolC<true>{}, BoolC<true>{});          // block-uniform
    else tile(kt, Kt0, Vt0, BoolC<true>{}, BoolC<false>{});
    lstore(Kt1, Vt1, kreg1, vreg1);                // tile kt+1
    __syncthreads();
    if (kt + 1 < nt) {                             // block-uniform
      gload(kreg1, vreg1, kt + 3 < nt ? kt + 3 : kt);
      if (kt * 32 + 64 <= nk) tile(kt + 1, Kt1, Vt1, BoolC<EVERY>{}, BoolC<true>{});
      else tile(kt + 1, Kt1, Vt1, BoolC<EVERY>{}, BoolC<false>{});
      lstore(Kt0, Vt0, kreg0, vreg0);              // tile kt+2
      __syncthreads();
    }
  }

  if (wave_active) {
#pragma unroll
    for (int o = 0; o < OB; ++o) O[o] += T[o];                    // the last group
    const float inv = (l > 0.f && qrow < nq) ? 1.0f / l : 0.f;   // rows past the valid count: zeros
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
// The pipelined form (used for HD = 64: the straight form above needs 2 x the V / P / O registers there and spills).
template <int HD>
__global__ __launch_bounds__(256, 2) void attention_x3p_kernel(AttnArgs p, float scale) {
  constexpr int TK = 32;
  constexpr int KSB = HD + 8;         // K plane row stride (bf16): 80 / 144 bytes = 5 / 9 sixteen-byte slots (odd)
  constexpr int OB = HD / 32;         // output blocks of 32 dims
  constexpr int NS = HD / 16;         // 16-dim steps of K.Q^T
  constexpr int V4 = HD / 4;          // float4 per K / V row
  constexpr int ITER = (TK * V4) / 256;
  // four buffers, separate objects: in one iteration tile k+1's K and tile k-1's V are read while tile k+2 is written, and the
  // compiler must see that those never alias (one scheduling region)
  __shared__ __attribute__((aligned(16))) __bf16 Kt0[3][TK * KSB];
  __shared__ __attribute__((aligned(16))) __bf16 Kt1[3][TK * KSB];
  __shared__ __attribute__((aligned(16))) __bf16 Kt2[3][TK * KSB];
  __shared__ __attribute__((aligned(16))) __bf16 Kt3[3][TK * KSB];
  __shared__ __attribute__((aligned(16))) __bf16 Vt0[3][TK * HD];
  __shared__ __attribute__((aligned(16))) __bf16 Vt1[3][TK * HD];
  __shared__ __attribute__((aligned(16))) __bf16 Vt2[3][TK * HD];
  __shared__ __attribute__((aligned(16))) __bf16 Vt3[3][TK * HD];

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

  // Q^T fragments (B operand): lane (query, kb = hi) holds dims 16 s + 8 hi .. + 7, pre-scaled (1/sqrt(HD) and log2 e), split
  bf16x8 qf[NS][3];
  {
    const float* qp = p.qkv + (qbase + min(qrow, Nqp - 1)) * ld + head * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s), c = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
      const float v[8] = {a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale, c[0] * scale, c[1] * scale, c[2] * scale, c[3] * scale};
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        bf16x2 h, m, l;
        split2(v[j], v[j + 1], h, m, l);
        qf[s][0][j] = h[0]; qf[s][0][j + 1] = h[1];
        qf[s][1][j] = m[0]; qf[s][1][j + 1] = m[1];
        qf[s][2][j] = l[0]; qf[s][2][j + 1] = l[1];
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
  auto gload = [&](int kt) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(kt * TK * ld * 4);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      kreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it], so, 0));
      vreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it] + p.d * 4, so, 0));
    }
  };
  // split the staged tile into the three planes
  auto lstore = [&](__bf16 (&Kd)[3][TK * KSB], __bf16 (&Vd)[3][TK * HD]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int e = tid + it * 256, key = e / V4, v4 = e % V4;
      bf16x4 kp[3], vp[3];
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        bf16x2 h, mm, ll;
        split2(kreg[it][j], kreg[it][j + 1], h, mm, ll);
        kp[0][j] = h[0]; kp[0][j + 1] = h[1]; kp[1][j] = mm[0]; kp[1][j + 1] = mm[1]; kp[2][j] = ll[0]; kp[2][j + 1] = ll[1];
        split2(vreg[it][j], vreg[it][j + 1], h, mm, ll);
        vp[0][j] = h[0]; vp[0][j + 1] = h[1]; vp[1][j] = mm[0]; vp[1][j + 1] = mm[1]; vp[2][j] = ll[0]; vp[2][j + 1] = ll[1];
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        *reinterpret_cast<bf16x4*>(&Kd[pl][key * KSB + 4 * v4]) = kp[pl];
        *reinterpret_cast<bf16x4*>(&Vd[pl][key * HD + 4 * v4]) = vp[pl];
      }
    }
  };

  // transposed-read address pattern of this lane inside a [4 keys][16 dims] block of a V plane
  const int tr_off = ((lane & 15) >> 2) * HD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // the six term products, smallest first: planes (A, B)
  constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};

  f32x16 O[OB], Tq[OB], S;  // running output; the product of tile k-2 (folded in this iteration); the scores of tile k
  bf16x8 pf[2][3];          // P^T fragments of tile k-1 (B operand): step t holds keys r = 8 t .. 8 t + 7 of the lane's sixteen
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) { O[o][r] = 0.f; Tq[o][r] = 0.f; }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[t][pl][j] = (__bf16)0.f;
  float m = -INFINITY, l = 0.f, a1 = 1.f, a2 = 1.f;     // a1 / a2: the rescale factors that go with the products of tiles k-1 / k-2

  // One iteration = tile k, k = 0 .. nt (the last one only drains).  A three-deep software pipeline inside the wave, so that the
  // MFMAs of an iteration never wait for its VALU work and the two MFMA chains alternate (a dependent bf16 MFMA issued right
  // behind its producer, with other instructions in between, stalls the pipe):
  //   MFMA:  scores of tile k+1 (K.Q^T)  interleaved with  the product of tile k-1 (V^T.P^T, from a zero accumulator)
  //   VALU:  fold the product of tile k-2 into the output (O = O * alpha + T: two-level accumulation at every tile, one fma per
  //          element), softmax + split of tile k's scores, split + store of tile k+2 (requested one iteration earlier)
  // one barrier per tile, no branch in the body except the key mask of the last tile.
  auto body = [&](int k, const __bf16 (&Kn)[3][TK * KSB], const __bf16 (&Vp)[3][TK * HD], __bf16 (&Kd)[3][TK * KSB],
                  __bf16 (&Vd)[3][TK * HD]) __attribute__((always_inline)) {
    // ---- operands of this iteration's MFMAs: K of tile k+1 (A), Q (B, registers); V^T of tile k-1 (A), P^T of tile k-1 (B)
    bf16x8 kf[NS][3], vf[OB][2][3];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) kf[s][pl] = *reinterpret_cast<const bf16x8*>(&Kn[pl][l31 * KSB + 16 * s + 8 * hi]);
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const __bf16* base = &Vp[pl][(16 * t + 4 * hi) * HD + 32 * o + tr_off];
          const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base));
          const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + 8 * HD));
          const u32x2 aw = __builtin_bit_cast(u32x2, a), cw = __builtin_bit_cast(u32x2, c);
          const u32x4 w = {aw[0], aw[1], cw[0], cw[1]};
          vf[o][t][pl] = __builtin_bit_cast(bf16x8, w);
        }
    f32x16 Sn = zero16, Tn[OB];
#pragma unroll
    for (int o = 0; o < OB; ++o) Tn[o] = zero16;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < 6 * NS) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i / 6][PA[i % 6]], qf[i / 6][PB[i % 6]], Sn, 0, 0, 0);
#pragma unroll
      for (int o = 0; o < OB; ++o)
        Tn[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[o][i / 6][PA[i % 6]], pf[i / 6][PB[i % 6]], Tn[o], 0, 0, 0);
    }
    if constexpr (NS > 2) {
#pragma unroll
      for (int i = 12; i < 6 * NS; ++i) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i / 6][PA[i % 6]], qf[i / 6][PB[i % 6]], Sn, 0, 0, 0);
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
    const float mn = live ? fmaxf(m, mx) : m;
    const float alpha = live ? __builtin_amdgcn_exp2f(m - mn) : 1.f;      // m = -inf on the first tile -> 0
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 mn2 = {mn, mn};
    f32x2 rs2 = {0.f, 0.f};
    bf16x8 pn[2][3];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 dd = (f32x2){S[r], S[r + 1]} - mn2;
      const f32x2 pp = {__builtin_amdgcn_exp2f(dd[0]), __builtin_amdgcn_exp2f(dd[1])};
      rs2 += pp;
      bf16x2 h, mm, ll;
      split2(pp[0], pp[1], h, mm, ll);
      const int t = r >> 3, j = r & 7;
      pn[t][0][j] = h[0]; pn[t][0][j + 1] = h[1];
      pn[t][1][j] = mm[0]; pn[t][1][j + 1] = mm[1];
      pn[t][2][j] = ll[0]; pn[t][2][j + 1] = ll[1];
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
      for (int pl = 0; pl < 3; ++pl) pf[t][pl] = pn[t][pl];
    __syncthreads();
  };

  gload(0);
  lstore(Kt0, Vt0);
  gload(nt > 1 ? 1 : 0);
  lstore(Kt1, Vt1);
  gload(nt > 2 ? 2 : 0);
  // iteration 0 multiplies P = 0 by the V buffer of "tile -1": it must hold finite values
  for (int e = tid; e < 3 * TK * HD / 8; e += 256) reinterpret_cast<u32x4*>(&Vt3[0][0])[e] = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();
  {                                       // scores of tile 0
    S = zero16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x8 kf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(&Kt0[pl][l31 * KSB + 16 * s + 8 * hi]);
      S = mfma6(kf, qf[s], S);
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
    const float inv = (l > 0.f && qrow < nq) ? 1.0f / l : 0.f;   // rows past the valid count: zeros
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
}  // namespace

bool attention_x3_supported(const AttnArgs& a) {
  const int hd = a.heads > 0 ? a.d / a.heads : 0;
  return hd == 32 || hd == 64;
}

hipError_t launch_attention_x3(const AttnArgs& a, hipStream_t s) {
  if (!attention_x3_supported(a)) return hipErrorInvalidValue;
  const int hd = a.d / a.heads;
  const int nmax = a.N0p > a.N1p ? a.N0p : a.N1p;
  dim3 grid((unsigned)((nmax + 127) / 128), (unsigned)a.heads, (unsigned)(2 * a.B));
  const float scale = (float)(1.4426950408889634 / sqrt((double)hd));   // log2(e)/sqrt(HD)
  last_form = "attention_x3:bf16x3";
  if (hd == 32) hipLaunchKernelGGL((attention_x3_kernel<32>), grid, dim3(256), 0, s, a, scale);
  else hipLaunchKernelGGL((attention_x3p_kernel<64>), grid, dim3(256), 0, s, a, scale);
  return hipGetLastError();
}

}  // namespace imx
