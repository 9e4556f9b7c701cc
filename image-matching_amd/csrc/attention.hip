// attention.hip — multi-head self/cross attention of SuperGlue's GNN, flash style, on the fp32
// matrix cores.  Replaces attention()/MultiHeadedAttention.forward's einsum-softmax-einsum
// (superglue_test.py:85-89, :102-106); the N x M probability matrix is never materialised.
//
// Layout: qkv rows = keypoints ([side0: B*N0p][side1: B*N1p]), ld = 3d, columns [q | k | v], each
// re-ordered head-major (head*HD + dim) when the projection weights are loaded (the reference's
// view(b, dim, heads, n) is dim-major/head-minor, :104).  Output rows likewise, ld = d.
//
// Workgroup = 4 waves; a wave owns 32 queries and walks all keys in tiles of 32 (K/V tile staged
// in LDS, double buffered, one barrier per tile).  Per tile and wave:
//   S^T = K.Q^T      HD/2 x v_mfma_f32_32x32x2_f32: lane (q = lane&31) ends up holding 16 keys'
//                    scores of ITS query (the other 16 live in lane^32) -> the softmax row
//                    reductions are in-lane plus one cross-half shuffle;
//   online softmax   running max m, denominator l, rescale factor all lane-local per query;
//   O^T += V^T.P^T   16 MFMAs per 32 output dims: the MFMA k-index is only a summation index, so
//                    step s pairs key (s&3)+8(s>>2) (lanes 0-31) with that key +4 (lanes 32-63):
//                    exactly the keys whose probabilities sit in accumulator register s — P is fed
//                    to the matrix core straight from the S accumulators, no transposes, no LDS.
#include "imx_kernels.h"
#include <math.h>
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vectors: arrays of HIP's float4 struct land in scratch

namespace {

// DEEP: two K/V tiles in flight in registers (a global load can take longer than one tile's MFMAs), same LDS.
// TK: keys per staged K/V tile and barrier (32 or 64; 64 = two 32-key sub-tiles multiplied back to back).
// XCD-aware work mapping: workgroups are dispatched round-robin over the 8 XCDs (each with its own L2), x fastest.  The
// query blocks of one (pair side, head) all stream the same K/V rows, so they should meet in ONE L2: workgroup L takes work
// item (L % 8) * (total / 8) + L / 8 -- consecutive items land on the same XCD, close in time (measured with rocprofv3 --pmc FETCH_SIZE: 0.56 GB -> 0.10 GB
// fetched per launch at 32 pairs).
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

template <int HD, bool DEEP, int TK = 32>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs p, float scale) {
  constexpr int KS = HD + 4;          // K tile row stride: rows 16-byte aligned, 8 lanes of a b128 read cover all 32 banks
  constexpr int OB = HD / 32;         // output blocks of 32 dims
  __shared__ __attribute__((aligned(16))) float Kt[2][TK * KS];
  __shared__ __attribute__((aligned(16))) float Vt[2][TK * HD];

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
  const bool wave_active = (q0 + 32 * wave) < Nqp;   // whole wave inside the padded row range

  // Q fragment: lane holds Q[q][16*hi*(HD/32) ...]: dims [hi*HD/2, hi*HD/2 + HD/2)
  float q[HD / 2];
  if (wave_active) {
    const float* qp = p.qkv + (qbase + qrow) * ld + head * HD + hi * (HD / 2);
#pragma unroll
    for (int t = 0; t < HD / 2; t += 4) {
      float4 v = *reinterpret_cast<const float4*>(qp + t);
      // fold 1/sqrt(HD) and log2(e) into Q: scores come out of the MFMA in the log2 domain
      q[t] = v.x * scale; q[t + 1] = v.y * scale; q[t + 2] = v.z * scale; q[t + 3] = v.w * scale;
    }
  } else {
#pragma unroll
    for (int t = 0; t < HD / 2; ++t) q[t] = 0.f;
  }

  f32x16 O[OB];
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[o][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int nt = (nk + TK - 1) / TK;        // staged tiles
  // staging map: 32 keys x HD dims, float4 per thread-iteration
  constexpr int V4 = HD / 4;                 // float4 per row
  constexpr int ITER = (TK * V4) / 256;      // 1..4
  f32x4 kreg[2][ITER], vreg[2][ITER];
#define IMX_GLOAD(set_, kt_)                                                                   \
  _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                        \
    const int e = tid + it * 256, key = e / V4, v4 = e % V4;                                   \
    const int krow = min((kt_) * TK + key, Nkp - 1);      /* rows past the padded count are never used */ \
    const float* base = p.qkv + (kbase + (size_t)krow) * ld + head * HD + 4 * v4;              \
    kreg[set_][it] = *reinterpret_cast<const f32x4*>(base + p.d);                              \
    vreg[set_][it] = *reinterpret_cast<const f32x4*>(base + 2 * p.d);                          \
  }
#define IMX_LSTORE(set_, buf_)                                                                 \
  _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                        \
    const int e = tid + it * 256, key = e / V4, v4 = e % V4;                                   \
    float* kd = &Kt[buf_][key * KS + 4 * v4];                                                  \
    *reinterpret_cast<f32x4*>(kd) = kreg[set_][it];                                            \
    *reinterpret_cast<f32x4*>(&Vt[buf_][key * HD + 4 * v4]) = vreg[set_][it];                  \
  }

  // one key tile: S^T = K.Q^T, online softmax, O^T = O^T*alpha + V^T.P^T
  auto tile32 = [&](int kt, int buf, int sub) __attribute__((always_inline)) {      // kt = 32-key tile index
    if (wave_active) {
      // ---- fetch this tile's K and V fragments from LDS up front (V lands during the S MFMAs)
      const float* kp = &Kt[buf][(sub * 32 + l31) * KS + hi * (HD / 2)];
      const float* vp = &Vt[buf][(sub * 32 + 4 * hi) * HD + l31];
      float kf[HD / 2], vf[OB][16];
#pragma unroll
      for (int t = 0; t < HD / 2; t += 4) {           // a lane's HD/2 values of its key row are contiguous: b128 reads
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + t);
        kf[t] = kv[0]; kf[t + 1] = kv[1]; kf[t + 2] = kv[2]; kf[t + 3] = kv[3];
      }
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int st = 0; st < 16; ++st) vf[o][st] = vp[((st & 3) + 8 * (st >> 2)) * HD + o * 32];
      __builtin_amdgcn_sched_barrier(0);
      // ---- S^T = K . Q^T
      // (the first MFMA takes a literal zero accumulator: no 16-register clear per tile)
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[0], q[0], zero16, 0, 0, 0);
#pragma unroll
      for (int t = 1; t < HD / 2; ++t) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], q[t], S, 0, 0, 0);
      // ---- online softmax over this tile's 32 keys (16 here, 16 in lane^32), log2 domain:
      //      p = 2^(s2 - m2) = e^(s - m); |abs err| of the one-multiply form <= 6e-8*max|x e^x| ~ 2e-8
      float mx = -INFINITY;
      if (kt * 32 + 32 <= nk) {           // full tile (block-uniform): no key masking
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
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);      // m = -inf on the first tile -> 0
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 mn2 = {mn, mn};
      f32x2 rs2 = {0.f, 0.f};                          // shift and row sum in packed ops (8 + 8 instead of 16 + 16)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 dd = (f32x2){S[r], S[r + 1]} - mn2;
        const f32x2 pp = {__builtin_amdgcn_exp2f(dd[0]), __builtin_amdgcn_exp2f(dd[1])};
        S[r] = pp[0];
        S[r + 1] = pp[1];
        rs2 += pp;
      }
      float rs = rs2[0] + rs2[1];
      rs += __shfl_xor(rs, 32);
      l = l * alpha + rs;
      m = mn;
      // ---- O^T = O^T * alpha + V^T . P^T ; the rescale is skipped when no lane of the wave saw a larger maximum
      //      (alpha == 1 exactly -- the common case after the first tiles)
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
        for (int o = 0; o < OB; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[o][r] *= alpha;
      }
#pragma unroll
      for (int o = 0; o < OB; ++o) {
#pragma unroll
        for (int st = 0; st < 16; ++st)
          O[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[o][st], S[st], O[o], 0, 0, 0);
      }
    }
  };

  auto tile = [&](int kts, int buf) __attribute__((always_inline)) {               // one staged tile = TK/32 sub-tiles
#pragma unroll
    for (int sub = 0; sub < TK / 32; ++sub) {
      const int kt = kts * (TK / 32) + sub;
      if (kt * 32 < nk) tile32(kt, buf, sub);          // block-uniform
    }
  };

  if constexpr (!DEEP) {
    if (nt > 0) { IMX_GLOAD(0, 0) IMX_LSTORE(0, 0) }
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
      const int buf = kt & 1;
      { IMX_GLOAD(0, kt + 1 < nt ? kt + 1 : kt) }   // branch-free prefetch (last tile re-fetches itself)
      tile(kt, buf);
      { IMX_LSTORE(0, buf ^ 1) }
      __syncthreads();
    }
  } else {
    // tile kt+2 is requested while tile kt is multiplied and tile kt+1 (requested one iteration earlier) moves from
    // registers to LDS at the end of the iteration: two static register sets, loop unrolled by two.
    if (nt > 0) { IMX_GLOAD(0, 0) IMX_LSTORE(0, 0) }
    { IMX_GLOAD(1, nt > 1 ? 1 : 0) }
    __syncthreads();
    for (int kt = 0; kt < nt; kt += 2) {
      { IMX_GLOAD(0, kt + 2 < nt ? kt + 2 : kt) }
      tile(kt, 0);
      { IMX_LSTORE(1, 1) }                          // tile kt+1
      __syncthreads();
      if (kt + 1 < nt) {                            // block-uniform
        { IMX_GLOAD(1, kt + 3 < nt ? kt + 3 : kt) }
        tile(kt + 1, 1);
        { IMX_LSTORE(0, 0) }                        // tile kt+2
        __syncthreads();
      }
    }
  }

  if (wave_active) {
    const float inv = (l > 0.f && qrow < nq) ? 1.0f / l : 0.f;   // rows past the valid count: zeros
    float* op = p.out + (qbase + qrow) * p.d + head * HD;
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(O[o][4 * g] * inv, O[o][4 * g + 1] * inv, O[o][4 * g + 2] * inv, O[o][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + o * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

#undef IMX_GLOAD
#undef IMX_LSTORE

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined form (default).  Same arithmetic, same per-key pairing; what changes is the ORDER inside a wave:
// iteration t issues the score MFMAs of tile t+1, then the P.V MFMAs of tile t with the softmax VALU work of tile
// t+1 interleaved between them (independent chains: S(t+1) -> P(t+1) vs O += V(t).P(t)), so the exponentials no longer
// sit in series with the matrix pipe.  K/V tiles live in a ring of three LDS buffers (tile t's V and tile t+1's K are
// needed together while tile t+2 is being staged); still one barrier per tile.  The running-max rescale of O and l is
// skipped when no lane of the wave saw a larger maximum (alpha == 1 exactly; common after the first tiles).
template <int HD>
__global__ __launch_bounds__(256) void attention2_kernel(AttnArgs p, float scale) {
  constexpr int KS = HD + 1;          // K tile row stride (odd: conflict-free column reads)
  constexpr int OB = HD / 32;         // output blocks of 32 dims
  __shared__ __attribute__((aligned(16))) float Kt[3][32 * KS];
  __shared__ __attribute__((aligned(16))) float Vt[3][32 * HD];

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
  const bool wave_active = (q0 + 32 * wave) < Nqp;

  float q[HD / 2];
  if (wave_active) {
    const float* qp = p.qkv + (qbase + qrow) * ld + head * HD + hi * (HD / 2);
#pragma unroll
    for (int t = 0; t < HD / 2; t += 4) {
      float4 v = *reinterpret_cast<const float4*>(qp + t);
      q[t] = v.x * scale; q[t + 1] = v.y * scale; q[t + 2] = v.z * scale; q[t + 3] = v.w * scale;
    }
  } else {
#pragma unroll
    for (int t = 0; t < HD / 2; ++t) q[t] = 0.f;
  }

  f32x16 O[OB];
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[o][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int nt = (nk + 31) / 32;
  constexpr int V4 = HD / 4;
  constexpr int ITER = (32 * V4) / 256;
  f32x4 kreg[ITER], vreg[ITER];
#define IMX_GLOAD(kt_)                                                                         \
  _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                        \
    const int e = tid + it * 256, key = e / V4, v4 = e % V4;                                   \
    const float* base = p.qkv + (kbase + (size_t)(kt_) * 32 + key) * ld + head * HD + 4 * v4;  \
    kreg[it] = *reinterpret_cast<const f32x4*>(base + p.d);                                   \
    vreg[it] = *reinterpret_cast<const f32x4*>(base + 2 * p.d);                               \
  }
#define IMX_LSTORE(buf_)                                                                       \
  _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                        \
    const int e = tid + it * 256, key = e / V4, v4 = e % V4;                                   \
    float* kd = &Kt[buf_][key * KS + 4 * v4];                                                  \
    kd[0] = kreg[it][0]; kd[1] = kreg[it][1]; kd[2] = kreg[it][2]; kd[3] = kreg[it][3];            \
    *reinterpret_cast<f32x4*>(&Vt[buf_][key * HD + 4 * v4]) = vreg[it];                       \
  }
  // scores of one tile: S^T = K.Q^T, masked past nk (tile index tile_)
#define IMX_SCORES(S_, slot_, tile_)                                                           \
  {                                                                                            \
    const float* kp = &Kt[slot_][l31 * KS + hi * (HD / 2)];                                    \
    float kf[HD / 2];                                                                          \
    _Pragma("unroll") for (int t = 0; t < HD / 2; ++t) kf[t] = kp[t];                          \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) S_[r] = 0.f;                                \
    _Pragma("unroll") for (int t = 0; t < HD / 2; ++t)                                         \
        S_ = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], q[t], S_, 0, 0, 0);                   \
  }
  // online softmax of a tile's scores (log2 domain): S_ -> probabilities, updates m, l; alpha_ = rescale of the past.
  // Branch-free (keys >= lim_ are masked by select, lim_ = 0 turns the whole tile into a no-op: alpha 1, P 0) so the
  // whole thing can be scheduled between the P.V MFMAs of the previous tile.
#define IMX_SOFTMAX(S_, tile_, lim_, alpha_)                                                   \
  {                                                                                            \
    float mx = -INFINITY;                                                                      \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                           \
      const int key = (tile_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                          \
      const float sv = key < (lim_) ? S_[r] : -INFINITY;                                       \
      S_[r] = sv;                                                                              \
      mx = fmaxf(mx, sv);                                                                      \
    }                                                                                          \
    mx = fmaxf(mx, __shfl_xor(mx, 32));                                                        \
    const float mn = fmaxf(m, mx);                                                             \
    alpha_ = __builtin_amdgcn_exp2f(m - mn);                                                   \
    float rs = 0.f;                                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                           \
      const float pr = __builtin_amdgcn_exp2f(S_[r] - mn);                                     \
      S_[r] = pr;                                                                              \
      rs += pr;                                                                                \
    }                                                                                          \
    rs += __shfl_xor(rs, 32);                                                                  \
    l = l * alpha_ + rs;                                                                       \
    m = mn;                                                                                    \
  }

  if (nt > 0) { IMX_GLOAD(0) IMX_LSTORE(0) }
  if (nt > 1) { IMX_GLOAD(1) IMX_LSTORE(1) }
  __syncthreads();

  f32x16 P, Sn;            // probabilities of the current tile, scores -> probabilities of the next one
  float alpha = 1.f;       // rescale that P's maximum imposed on everything accumulated before it
#pragma unroll
  for (int r = 0; r < 16; ++r) { P[r] = 0.f; Sn[r] = 0.f; }
  if (wave_active && nt > 0) {
    IMX_SCORES(P, 0, 0)
    IMX_SOFTMAX(P, 0, nk, alpha)
  }
  int s0 = 0;              // ring slot of tile kt
  for (int kt = 0; kt < nt; ++kt) {
    const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
    const int lim = kt + 1 < nt ? nk : 0;      // past the last tile the look-ahead softmax is a no-op
    { IMX_GLOAD(kt + 2 < nt ? kt + 2 : kt) }   // branch-free prefetch (the tail re-fetches an old tile, never consumed)
    if (wave_active) {
      // rescale by this tile's alpha only if some lane's maximum moved (alpha == 1 exactly otherwise)
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
        for (int o = 0; o < OB; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[o][r] *= alpha;
      }
      // ---- one straight-line block: scores of tile kt+1, then O^T += V^T . P^T of tile kt with tile kt+1's softmax
      //      scheduled into the MFMA shadows
      const float* vp = &Vt[s0][(4 * hi) * HD + l31];
      float vf[OB][16];
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int st = 0; st < 16; ++st) vf[o][st] = vp[((st & 3) + 8 * (st >> 2)) * HD + o * 32];
      IMX_SCORES(Sn, s1, kt + 1)
      __builtin_amdgcn_sched_barrier(0);
      // P.V MFMAs of tile kt, with tile kt+1's softmax hand-sliced between them (the order is pinned with scheduling
      // fences: left alone, the compiler issues all MFMAs back to back and the exponentials after them).
      //   MFMAs 0-3: masked maximum (4 elements each) | after 3: cross-half max, new maximum, alpha
      //   MFMAs 4-11: exponentials (2 elements each)  | after 11: cross-half sum, l, m
      float mx = -INFINITY, rs = 0.f, mn = m, alpha_n = 1.f;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
#pragma unroll
        for (int o = 0; o < OB; ++o) O[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[o][st], P[st], O[o], 0, 0, 0);
        if (st < 4) {
#pragma unroll
          for (int r = 4 * st; r < 4 * st + 4; ++r) {
            const int key = (kt + 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float sv = key < lim ? Sn[r] : -INFINITY;
            Sn[r] = sv;
            mx = fmaxf(mx, sv);
          }
          if (st == 3) {
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            mn = fmaxf(m, mx);
            alpha_n = __builtin_amdgcn_exp2f(m - mn);
          }
        } else if (st < 12) {
#pragma unroll
          for (int r = 2 * (st - 4); r < 2 * (st - 4) + 2; ++r) {
            const float pr = __builtin_amdgcn_exp2f(Sn[r] - mn);
            Sn[r] = pr;
            rs += pr;
          }
          if (st == 11) {
            rs += __shfl_xor(rs, 32);
            l = l * alpha_n + rs;
            m = mn;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      alpha = alpha_n;
      P = Sn;
    }
    { IMX_LSTORE(s2) }
    __syncthreads();
    s0 = s1;
  }

  if (wave_active) {
    const float inv = (l > 0.f && qrow < nq) ? 1.0f / l : 0.f;
    float* op = p.out + (qbase + qrow) * p.d + head * HD;
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(O[o][4 * g] * inv, O[o][4 * g + 1] * inv, O[o][4 * g + 2] * inv, O[o][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + o * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

#undef IMX_GLOAD
#undef IMX_LSTORE
#undef IMX_SCORES
#undef IMX_SOFTMAX
}  // namespace

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  const int hd = a.d / a.heads;
  const int nmax = a.N0p > a.N1p ? a.N0p : a.N1p;
  dim3 grid((unsigned)((nmax + 127) / 128), (unsigned)a.heads, (unsigned)(2 * a.B));
  const float scale = (float)(1.4426950408889634 / sqrt((double)hd));   // log2(e)/sqrt(HD)
  // A/B switch IMX_ATTN: 1 = one K/V tile in flight, 2 = software-pipelined softmax (attention2_kernel), 3 = two tiles in
  // flight (default).  Measured on MI355X, whole step, C3 (HD=32, 64 pairs) / C5 (HD=64, 8 pairs), pairs/s:
  // 1: 1172 / 203.1, 2: 1163 / 203.6, 3: 1185 / 207.6 -- VALU work placed between MFMAs is not free (in-order issue),
  // so the pipelined form gains nothing; keeping two tiles of K/V in flight hides the L2 latency of the staging loads.
  const char* env = getenv("IMX_ATTN");        // read per launch (tests switch it within one process)
  const int mode = env ? atoi(env) : 3;
  if (hd == 32) {
    if (mode == 1) hipLaunchKernelGGL((attention_kernel<32, false>), grid, dim3(256), 0, s, a, scale);
    else if (mode == 2) hipLaunchKernelGGL(attention2_kernel<32>, grid, dim3(256), 0, s, a, scale);
    else if (mode == 4) hipLaunchKernelGGL((attention_kernel<32, true, 64>), grid, dim3(256), 0, s, a, scale);
    else hipLaunchKernelGGL((attention_kernel<32, true>), grid, dim3(256), 0, s, a, scale);
  } else if (hd == 64) {
    if (mode == 1) hipLaunchKernelGGL((attention_kernel<64, false>), grid, dim3(256), 0, s, a, scale);
    else if (mode == 2) hipLaunchKernelGGL(attention2_kernel<64>, grid, dim3(256), 0, s, a, scale);
    else if (mode == 4) hipLaunchKernelGGL((attention_kernel<64, true, 64>), grid, dim3(256), 0, s, a, scale);
    else hipLaunchKernelGGL((attention_kernel<64, true>), grid, dim3(256), 0, s, a, scale);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace imx
