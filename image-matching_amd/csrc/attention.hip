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
// Accumulation order (round 2).  v_mfma_f32_* accumulates exactly like a sequential RNE fma chain (tools/ubench/mfma_round.hip),
// and P >= 0, so one running accumulator over all N keys is a 1024-long chain whose partial sums only grow: emulating that
// order on the CPU (tools/accuracy_emul.py) puts scores_in 2.2x further from a float64 evaluation than the reference's own
// fp32 result, exactly what the kernel measured (2.1x at C3, 2.5x at C5).  The product is therefore accumulated in two levels:
// the P.V MFMAs of a GROUP of 64 keys start from a zero accumulator T, and T is folded into the running O with packed adds
// (8 v_pk_add_f32 per 32 output dims and group): 0.92x of the reference's error in the emulation, ~1 % of the tile's time.
#include "imx_kernels.h"
#include <math.h>
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vectors: arrays of HIP's float4 struct land in scratch

namespace {

// Two K/V tiles are in flight in registers (a global load can take longer than one tile's MFMAs).
// TK: keys per staged K/V tile and barrier (32; 64 at head dim 16 = two 32-key sub-tiles multiplied back to back, so that the
// staging map stays one float4 of K and of V per thread).
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

// max / sum over a lane and its partner lane^32
// (one v_permlane32_swap instead of the ds_bpermute round trip of __shfl_xor was measured: no difference, 10.2-10.4 ms per
//  step either way; its __builtin_amdgcn_permlane32_swap form clobbered a live register under hipcc 7.2 -- memory fault)
__device__ __forceinline__ float xhalf_max(float x) { return fmaxf(x, __shfl_xor(x, 32)); }
__device__ __forceinline__ float xhalf_sum(float x) { return x + __shfl_xor(x, 32); }

template <bool V>
struct BoolC { static constexpr bool value = V; };

template <int HD, int TK = 32>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs p, float scale) {
  constexpr int KS = HD + 4;          // K tile row stride: rows 16-byte aligned, 8 lanes of a b128 read cover all 32 banks
  // HD = 16 (descriptor_dim 64, reference README.md:134-140): the P.V product still runs on the 32x32x2 MFMA, whose output
  // block is 32 dims wide -- the V tile is staged 32 columns wide with columns 16..31 zero (written once), and only the
  // first 16 output dims are stored.  Half of that product is padding; the configuration is a quarter of C3's work anyway.
  constexpr int HV = HD < 32 ? 32 : HD;   // V tile row width
  constexpr int OB = HV / 32;         // output blocks of 32 dims
  __shared__ __attribute__((aligned(16))) float Kt[2][TK * KS];
  __shared__ __attribute__((aligned(16))) float Vt[2][TK * HV];

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

  f32x16 O[OB], T[OB];      // running output, and the current 64-key group's partial product (two-level accumulation)
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) { O[o][r] = 0.f; T[o][r] = 0.f; }
  float m = -INFINITY, l = 0.f;

  const int nt = (nk + TK - 1) / TK;        // staged tiles
  // staging map: 32 keys x HD dims, float4 per thread-iteration
  constexpr int V4 = HD / 4;                 // float4 per row
  constexpr int ITER = (TK * V4) / 256;      // 1..4
  f32x4 kreg[2][ITER], vreg[2][ITER];
  // K/V rows of this (pair, side) through a buffer descriptor: per-thread byte offsets are loop invariant, the tile offset
  // is an SGPR, rows past the padded count read as zeros (never used) -- no address arithmetic in the loop (12 VALU
  // instructions per tile with global loads; the buffer form pays for the two-level accumulation: 10.7 -> 10.3 ms)
  // (the base is block-uniform, but it is a phi over divergent control flow for the compiler: without the explicit
  //  readfirstlane every load becomes a waterfall loop)
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
#define IMX_GLOAD(set_, kt_)                                                                   \
  {                                                                                            \
    const int so = __builtin_amdgcn_readfirstlane((kt_) * TK * ld * 4);                        \
    _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                      \
      kreg[set_][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it], so, 0));            \
      vreg[set_][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[it] + p.d * 4, so, 0));  \
    }                                                                                          \
  }
#define IMX_LSTORE(set_, buf_)                                                                 \
  _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                        \
    const int e = tid + it * 256, key = e / V4, v4 = e % V4;                                   \
    float* kd = &Kt[buf_][key * KS + 4 * v4];                                                  \
    *reinterpret_cast<f32x4*>(kd) = kreg[set_][it];                                            \
    *reinterpret_cast<f32x4*>(&Vt[buf_][key * HV + 4 * v4]) = vreg[set_][it];                  \
  }

  // one key tile: S^T = K.Q^T, online softmax, O^T = O^T*alpha + V^T.P^T
  // `first`: this tile opens a 64-key group (folds the finished group T into O, then starts T from a zero accumulator)
  auto tile32 = [&](int kt, int buf, int sub, auto first) __attribute__((always_inline)) {      // kt = 32-key tile index
    if (wave_active) {
      // ---- fetch this tile's K and V fragments from LDS up front (V lands during the S MFMAs)
      const float* kp = &Kt[buf][(sub * 32 + l31) * KS + hi * (HD / 2)];
      const float* vp = &Vt[buf][(sub * 32 + 4 * hi) * HV + l31];
      float kf[HD / 2], vf[OB][16];
#pragma unroll
      for (int t = 0; t < HD / 2; t += 4) {           // a lane's HD/2 values of its key row are contiguous: b128 reads
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + t);
        kf[t] = kv[0]; kf[t + 1] = kv[1]; kf[t + 2] = kv[2]; kf[t + 3] = kv[3];
      }
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int st = 0; st < 16; ++st) vf[o][st] = vp[((st & 3) + 8 * (st >> 2)) * HV + o * 32];
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
      mx = xhalf_max(mx);
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
      float rs = xhalf_sum(rs2[0] + rs2[1]);
      l = l * alpha + rs;
      m = mn;
      // ---- (O^T + T) * alpha + V^T . P^T, two-level: the rescale is skipped when no lane of the wave saw a larger maximum
      //      (alpha == 1 exactly -- the common case after the first tiles)
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
        for (int o = 0; o < OB; ++o) {
          T[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[o][0], S[0], zero16, 0, 0, 0);
#pragma unroll
          for (int st = 1; st < 16; ++st) T[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[o][st], S[st], T[o], 0, 0, 0);
        }
      } else {
        if (rescale) {
#pragma unroll
          for (int o = 0; o < OB; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) { O[o][r] *= alpha; T[o][r] *= alpha; }
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) {
#pragma unroll
          for (int st = 0; st < 16; ++st) T[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[o][st], S[st], T[o], 0, 0, 0);
        }
      }
    }
  };

  // one staged tile = TK/32 sub-tiles.  Groups of two 32-key tiles: with 64-key staged tiles a group is one staged tile,
  // with 32-key staged tiles the even staged tiles open a group (`even` is a compile-time tag at every call site)
  // HD = 64 (C5: 2048 keys, d = 256) folds every 32-key tile: in the emulation of this order C5's scores_in goes from 1.48x
  // to 1.23x of the reference's own fp32 distance to float64, for 16 more packed adds per 4096 MFMA cycles.
  constexpr bool EVERY = HD >= 64;
  auto tile = [&](int kts, int buf, auto even) __attribute__((always_inline)) {
    if constexpr (TK == 64) {
      if (kts * 64 < nk) tile32(kts * 2, buf, 0, BoolC<true>{});            // block-uniform
      if (kts * 64 + 32 < nk) tile32(kts * 2 + 1, buf, 1, BoolC<EVERY>{});
    } else {
      if (kts * 32 < nk) tile32(kts, buf, 0, BoolC<EVERY || decltype(even)::value>{});
    }
  };

  if constexpr (HV != HD) {                 // zero columns HD..HV-1 of both V buffers, never written again
    for (int e = tid; e < 2 * TK * (HV - HD); e += 256) {
      const int row = e / (HV - HD), c = HD + e % (HV - HD);
      (&Vt[0][0])[row * HV + c] = 0.f;
    }
  }
  // tile kt+2 is requested while tile kt is multiplied and tile kt+1 (requested one iteration earlier) moves from
  // registers to LDS at the end of the iteration: two static register sets, loop unrolled by two.
  if (nt > 0) { IMX_GLOAD(0, 0) IMX_LSTORE(0, 0) }
  { IMX_GLOAD(1, nt > 1 ? 1 : 0) }
  __syncthreads();
  for (int kt = 0; kt < nt; kt += 2) {
    { IMX_GLOAD(0, kt + 2 < nt ? kt + 2 : kt) }
    tile(kt, 0, BoolC<true>{});
    { IMX_LSTORE(1, 1) }                          // tile kt+1
    __syncthreads();
    if (kt + 1 < nt) {                            // block-uniform
      { IMX_GLOAD(1, kt + 3 < nt ? kt + 3 : kt) }
      tile(kt + 1, 1, BoolC<false>{});
      { IMX_LSTORE(0, 0) }                        // tile kt+2
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
      for (int g = 0; g < (HD < 32 ? HD / 8 : 4); ++g) {      // accumulator registers 4g..4g+3 = dims 8g + 4hi ..
        float4 v = make_float4(O[o][4 * g] * inv, O[o][4 * g + 1] * inv, O[o][4 * g + 2] * inv, O[o][4 * g + 3] * inv);
        *reinterpret_cast<float4*>(op + o * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

#undef IMX_GLOAD
#undef IMX_LSTORE

// ---------------------------------------------------------------------------------------------------------------
// Key-split form for SMALL grids (single pair, BASELINE configs[2]): with one pair the kernel above launches
// (N/128) x heads x 2 = 64 workgroups on 256 CUs, each walking all N keys -- 47 us per launch, 18 launches per pair, the
// largest single item of the single-pair latency.  Here a workgroup owns 32 queries and its four waves split the KEYS:
// a 128-key super tile is staged cooperatively (one buffer, two barriers), wave w multiplies sub-tile w, and the four
// (max, sum, output) partials are merged through LDS at the end.  4x the workgroups, 1/4 of the serial work per wave.
// Same per-tile arithmetic as attention_kernel (tile maximum, log2-domain softmax, P fed from the S accumulators); the
// accumulation order differs (four partial sums), so results agree with the throughput form to rounding, not bit for bit.
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) void attention_split_kernel(AttnArgs p, float scale) {
  constexpr int KS = HD + 4, HV = HD < 32 ? 32 : HD, OB = HV / 32, V4 = HD / 4;
  constexpr int NT = 64 * NW;                   // threads: NW waves, each multiplying its own 32-key sub-tile
  constexpr int TKS = 32 * NW;                  // keys per staged super tile
  constexpr int ITER = (TKS * V4) / NT;         // float4 of K (and of V) per thread and super tile: 2 / 4 / 8
  extern __shared__ __attribute__((aligned(16))) float asm_[];
  float* Kt = asm_;                              // [128][KS]
  float* Vt = asm_ + TKS * KS;                   // [128][HV]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const AttnBlock blk = attn_block();
  const int head = blk.y;
  const int side = blk.z / p.B, b = blk.z % p.B;
  const int kside = p.cross ? 1 - side : side;
  const int Nqp = side ? p.N1p : p.N0p, Nkp = kside ? p.N1p : p.N0p;
  const int q0 = blk.x * 32;
  if (q0 >= Nqp) return;
  const int nq = side ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const int nk = kside ? (p.n1 ? p.n1[b] : p.N1) : (p.n0 ? p.n0[b] : p.N0);
  const size_t qbase = (side ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nqp;
  const size_t kbase = (kside ? (size_t)p.B * p.N0p : 0) + (size_t)b * Nkp;
  const int ld = 3 * p.d;
  const int qrow = q0 + l31;                     // every wave: the same 32 queries (N?p is a multiple of 32)

  float q[HD / 2];
  {
    const float* qp = p.qkv + (qbase + qrow) * ld + head * HD + hi * (HD / 2);
#pragma unroll
    for (int t = 0; t < HD / 2; t += 4) {
      float4 v = *reinterpret_cast<const float4*>(qp + t);
      q[t] = v.x * scale; q[t + 1] = v.y * scale; q[t + 2] = v.z * scale; q[t + 3] = v.w * scale;
    }
  }
  f32x16 O[OB];
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[o][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  if constexpr (HV != HD) {
    for (int e = tid; e < TKS * (HV - HD); e += NT) Vt[(e / (HV - HD)) * HV + HD + e % (HV - HD)] = 0.f;
  }
  const unsigned long long kaddr = (unsigned long long)(p.qkv + kbase * ld);
  const unsigned long long kaddr_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(kaddr >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((unsigned)kaddr);
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kaddr_u, 0, __builtin_amdgcn_readfirstlane(Nkp * ld * 4), 0x00020000);
  int kvo[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int e = tid + it * NT, key = e / V4, v4 = e % V4;
    // rows past the padded count get an out-of-range VGPR offset (the SGPR tile offset is not bounds-checked): zeros
    kvo[it] = (key * ld + head * HD + 4 * v4 + p.d) * 4;
  }
  f32x4 kreg[ITER], vreg[ITER];
  auto gload = [&](int st) {
    const int so = __builtin_amdgcn_readfirstlane(st * TKS * ld * 4);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int key = (tid + it * NT) / V4;
      const int vo = st * TKS + key < Nkp ? kvo[it] : 0x7ffffff0;
      kreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, vo, so, 0));
      vreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, vo + p.d * 4, so, 0));
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int e = tid + it * NT, key = e / V4, v4 = e % V4;
      *reinterpret_cast<f32x4*>(&Kt[key * KS + 4 * v4]) = kreg[it];
      *reinterpret_cast<f32x4*>(&Vt[key * HV + 4 * v4]) = vreg[it];
    }
  };

  const int nst = (nk + TKS - 1) / TKS;
  if (nst > 0) gload(0);
  for (int st = 0; st < nst; ++st) {
    __syncthreads();                   // the previous super tile's readers are done
    lstore();
    __syncthreads();
    if (st + 1 < nst) gload(st + 1);   // block-uniform; in flight during the MFMAs below
    const int k0 = st * TKS + 32 * wave;              // this wave's 32 keys
    if (k0 < nk) {
      const float* kp = &Kt[(32 * wave + l31) * KS + hi * (HD / 2)];
      const float* vp = &Vt[(32 * wave + 4 * hi) * HV + l31];
      float kf[HD / 2], vf[OB][16];
#pragma unroll
      for (int t = 0; t < HD / 2; t += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + t);
        kf[t] = kv[0]; kf[t + 1] = kv[1]; kf[t + 2] = kv[2]; kf[t + 3] = kv[3];
      }
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int st2 = 0; st2 < 16; ++st2) vf[o][st2] = vp[((st2 & 3) + 8 * (st2 >> 2)) * HV + o * 32];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[0], q[0], zero16, 0, 0, 0);
#pragma unroll
      for (int t = 1; t < HD / 2; ++t) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], q[t], S, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float sv = key < nk ? S[r] : -INFINITY;
        S[r] = sv;
        mx = fmaxf(mx, sv);
      }
      mx = xhalf_max(mx);
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(S[r] - mn);
        S[r] = pr;
        rs += pr;
      }
      rs = xhalf_sum(rs);
      l = l * alpha + rs;
      m = mn;
      // per-tile product from a zero accumulator, folded into O with the rescale (two-level accumulation as above)
#pragma unroll
      for (int o = 0; o < OB; ++o) {
        f32x16 T = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[o][0], S[0], zero16, 0, 0, 0);
#pragma unroll
        for (int st2 = 1; st2 < 16; ++st2) T = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[o][st2], S[st2], T, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) O[o][r] = O[o][r] * alpha + T[r];
      }
    }
  }
  // ---- merge the four key ranges: wave w publishes (m, l) per query and its unnormalised O; every thread then combines
  //      one (query, 4 dims) group:  out = sum_w 2^(m_w - m*) O_w / sum_w 2^(m_w - m*) l_w
  __syncthreads();
  float* Om = asm_;                               // [NW][32 queries][HD + 4]   (the K/V staging area is free now)
  float* ml = asm_ + NW * 32 * (HD + 4);          // [NW][32][2]
  constexpr int OS = HD + 4;
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int g = 0; g < (HD < 32 ? HD / 8 : 4); ++g) {
      const f32x4 v = {O[o][4 * g], O[o][4 * g + 1], O[o][4 * g + 2], O[o][4 * g + 3]};
      *reinterpret_cast<f32x4*>(&Om[(wave * 32 + l31) * OS + o * 32 + 8 * g + 4 * hi]) = v;
    }
  if (hi == 0) { ml[(wave * 32 + l31) * 2] = m; ml[(wave * 32 + l31) * 2 + 1] = l; }
  __syncthreads();
  for (int e = tid; e < 32 * (HD / 4); e += NT) {
    const int qi = e / (HD / 4), d4 = (e % (HD / 4)) * 4;
    float mw[NW], ms = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) { mw[w] = ml[(w * 32 + qi) * 2]; ms = fmaxf(ms, mw[w]); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float lt = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = mw[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw[w] - ms);
      lt += f * ml[(w * 32 + qi) * 2 + 1];
      acc += f * *reinterpret_cast<const f32x4*>(&Om[(w * 32 + qi) * OS + d4]);
    }
    const float inv = (lt > 0.f && q0 + qi < nq) ? 1.0f / lt : 0.f;        // rows past the valid count: zeros
    *reinterpret_cast<f32x4*>(p.out + (qbase + q0 + qi) * p.d + head * HD + d4) = acc * inv;
  }
}

}  // namespace

// the form choice of launch_attention, for callers that prepare something form-specific (the q / k / v maxima of the two-plane fp16 form)
static bool attention_splits_keys(const AttnArgs& a) {
  const int nmax = a.N0p > a.N1p ? a.N0p : a.N1p;
  const long wgs = (long)((nmax + 127) / 128) * a.heads * 2 * a.B;
  return a.latency_forms >= 0 ? a.latency_forms != 0 : (wgs <= 256 && nmax >= 256);
}
bool attention_takes_x3(const AttnArgs& a) { return !attention_splits_keys(a) && !a.mfma_f32 && attention_x3_supported(a); }

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  const int hd = a.d / a.heads;
  const int nmax = a.N0p > a.N1p ? a.N0p : a.N1p;
  dim3 grid((unsigned)((nmax + 127) / 128), (unsigned)a.heads, (unsigned)(2 * a.B));
  const float scale = (float)(1.4426950408889634 / sqrt((double)hd));   // log2(e)/sqrt(HD)
  // Three forms, each with its reason:
  //   attention_split  small grids (one to four pairs): 32-query workgroups whose waves split the KEYS -- latency;
  //   attention_x3     head dim 32 / 64, the throughput form: both products on the 16-bit matrix pipe -- as three fp16 term products of
  //                    two-plane operands when the caller passes the q / k / v maxima (AttnArgs::amax), else six bf16 term products;
  //   attention_kernel fp32 MFMA: head dim 16 (descriptor_dim 64) and the "mfma" = "f32" A/B reference of the parity tests.
  // "latency_forms" = "off" keeps the throughput form for every batch size (then results do not depend on the batch size bit
  // for bit), "on" forces the key-split form.
  const bool split = attention_splits_keys(a);
  if (split) {
    dim3 sgrid((unsigned)((nmax + 31) / 32), (unsigned)a.heads, (unsigned)(2 * a.B));
    auto launch = [&](auto kern, int hdv, int nw) {
      const int ks = hdv + 4, hv = hdv < 32 ? 32 : hdv;
      const size_t stage = (size_t)32 * nw * (ks + hv) * 4, merge = (size_t)(nw * 32 * (hdv + 4) + nw * 32 * 2) * 4;
      const size_t lds = stage > merge ? stage : merge;
      static unsigned long long attr[3] = {0, 0, 0};
      raise_lds_limit(reinterpret_cast<const void*>(kern), (int)lds, attr[hdv == 16 ? 0 : hdv == 32 ? 1 : 2]);
      hipLaunchKernelGGL(kern, sgrid, dim3(64 * nw), lds, s, a, scale);
    };
    last_form = "attention_split:f32";
    if (hd == 16) launch(attention_split_kernel<16, 8>, 16, 8);
    else if (hd == 32) launch(attention_split_kernel<32, 8>, 32, 8);
    else if (hd == 64) launch(attention_split_kernel<64, 4>, 64, 4);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (!a.mfma_f32 && attention_x3_supported(a)) return launch_attention_x3(a, s);
  last_form = "attention:f32";
  if (hd == 32) hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(256), 0, s, a, scale);
  else if (hd == 64) hipLaunchKernelGGL((attention_kernel<64>), grid, dim3(256), 0, s, a, scale);
  else if (hd == 16) hipLaunchKernelGGL((attention_kernel<16, 64>), grid, dim3(256), 0, s, a, scale);   // 64 keys x 4 float4 = one float4 of K and of V per thread and tile
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace imx
