// sg_misc.hip — SuperGlue stages that are bandwidth/latency bound (no matrix work):
//   normalize_keypoints + first keypoint-encoder layer   superglue_test.py:63-82
//   descriptor gather into the row layout
//   log-domain Sinkhorn with implicit dustbins           superglue_test.py:141-170
//   mutual-nearest-neighbour match extraction            superglue_test.py:268-285
#include "imx_kernels.h"
#include <cstdlib>
#include <math.h>

namespace imx {
namespace {

// Online log-sum-exp with ONE v_exp_f32 per element (the running sum is rescaled only when the
// maximum moves):   d = x - m;  e = exp(-|d|) = 2^(-|d|*log2 e);  s = d > 0 ? s*e + 1 : s + e;  m = max(m, x)
// The maximum stays exact (natural domain); the one-multiply exponent form has absolute error
// <= 6e-8 * max|t 2^t| ~ 2e-8 per term (terms far below the maximum are negligible anyway).
constexpr float LOG2E = 1.4426950408889634f;
struct LSE { float m, s; };   // running max and sum of exp(x - m)

__device__ __forceinline__ float exp_neg(float t) {     // exp(-t), t >= 0 (t = +inf -> 0)
  return __builtin_amdgcn_exp2f(-t * LOG2E);
}
__device__ __forceinline__ void lse_add(LSE& a, float x) {
  const float d = x - a.m;                         // +inf on the first element (a.m = -inf)
  const float e = exp_neg(fabsf(d));
  a.s = d > 0.f ? fmaf(a.s, e, 1.0f) : a.s + e;
  a.m = fmaxf(a.m, x);
}
__device__ __forceinline__ LSE lse_merge(const LSE& a, const LSE& b) {
  const float nm = fmaxf(a.m, b.m);
  const float ea = (a.m == -INFINITY) ? 0.f : exp_neg(nm - a.m);
  const float eb = (b.m == -INFINITY) ? 0.f : exp_neg(nm - b.m);
  return LSE{nm, fmaf(b.s, eb, a.s * ea)};          // (explicit: left to contraction, two instantiations of one kernel fused different products)
}
// b.m finite (a slab's column always holds a row): ONE exponential -- of the two factors of lse_merge one is exp(0) = 1.  An empty
// accumulator (a.m = -inf, a.s = 0): d = -inf, e = 0, the sum is b's.
__device__ __forceinline__ LSE lse_merge1(const LSE& a, const LSE& b) {
  const float d = a.m - b.m;
  const float e = exp_neg(fabsf(d));
  return LSE{fmaxf(a.m, b.m), d >= 0.f ? fmaf(b.s, e, a.s) : fmaf(a.s, e, b.s)};
}
__device__ __forceinline__ LSE wave_lse(LSE a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    LSE b{__shfl_xor(a.m, o), __shfl_xor(a.s, o)};
    a = lse_merge(a, b);
  }
  return a;
}
__device__ __forceinline__ float lse_value(const LSE& a) { return a.m + logf(a.s); }
// Wave-wide max / sum of the slab kernel's row pass as DPP operations (round 6): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
// leave every lane of a 16-lane row with the row's result, row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry it to the last row,
// v_readlane 63 hands it back as a scalar: 6 VALU instructions and no LDS round trip, where __shfl_xor was 6 x (ds_bpermute + its
// address arithmetic + a wait on the LDS queue).  All 64 lanes must be active (the row pass branches on the wave index only).
// (inline assembly: through __builtin_amdgcn_update_dpp every step became v_mov + v_mov_dpp + a canonicalising v_max + the operation;
// s_nop 1 = the two wait states between a VALU write and a DPP read of the same register)
#define IMX_WAVE_REDUCE_DPP(OP)                                                   \
  asm volatile("s_nop 1\n\t"                                                      \
               OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
               "s_nop 1\n\t"                                                      \
               OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
               "s_nop 1\n\t"                                                      \
               OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
               "s_nop 1\n\t"                                                      \
               OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
               "s_nop 1\n\t"                                                      \
               OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
               "s_nop 1\n\t"                                                      \
               OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
               "s_nop 1"                                                          \
               : "+v"(v))
__device__ __forceinline__ float wave_max_u(float v) {
  IMX_WAVE_REDUCE_DPP("v_max_f32_dpp");
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_u(float v) {
  IMX_WAVE_REDUCE_DPP("v_add_f32_dpp");
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float uniform_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
// The two-pass forms below evaluate exp(t - mx) as exp2(fma(t, LOG2E, ml)) with ml = -(mx * LOG2E) ROUNDED: every term of a sum
// then carries the common factor 2^d, d = fma(mx, LOG2E, ml) = the product's rounding error (exact; up to 1.5e-5 at |mx| ~ 250,
// i.e. a 1e-5 bias of that log-sum-exp -- round 3: on a 7 x 64 transport problem with |Z| ~ 230 it put Z 3.3x further from float64
// than the oracle's fp32 result).  It is removed once per sum: 2^-d = 1 - d ln 2 to 1e-10.
__device__ __forceinline__ float lse_unbias(float sum, float mx, float ml) { return fmaf(sum, -0.69314718f * fmaf(mx, LOG2E, ml), sum); }

// ------------------------------------------------------------------ kenc layer 0
__global__ __launch_bounds__(256) void gather_desc_kernel(const float* __restrict__ src, long sb, long sc, long sn,
                                                          int B, int N, int Np, int d, float* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Np * d;
  if (e >= total) return;
  const int c = (int)(e % d);
  const long row = e / d;
  const int i = (int)(row % Np), b = (int)(row / Np);
  out[e] = i < N ? src[b * sb + c * sc + i * sn] : 0.f;
}

// both sides' gather + kenc layer 0 in one grid: blocks [0, g0) gather side 0, [g0, k0) kenc0 side 0, [k0, g1) gather side 1,
// [g1, ...) kenc0 side 1 -- the element functions above, unchanged
__device__ __forceinline__ void kenc0_element(const Kenc0Args& a, long e) {
  const long total = (long)a.B * a.Np * a.C1;
  if (e >= total) return;
  const int c = (int)(e % a.C1);
  const long row = e / a.C1;
  const int i = (int)(row % a.Np), b = (int)(row / a.Np);
  float v = 0.f;
  if (i < a.N) {
    const float* kp = a.kpts + ((size_t)b * a.N + i) * 2;
    const float xn = (kp[0] - a.cx) / a.scaling;          // normalize_keypoints (:63-70)
    const float yn = (kp[1] - a.cy) / a.scaling;
    const float sc = a.scores[(size_t)b * a.N + i];
    v = a.bias[c];
    v = fmaf(a.w[c], xn, v);
    v = fmaf(a.w[a.C1 + c], yn, v);
    v = fmaf(a.w[2 * a.C1 + c], sc, v);
    v = fmaxf(v, 0.f);
  }
  a.out[e] = v;
}
__global__ __launch_bounds__(256) void sg_prologue_kernel(SgPrologueArgs a, unsigned g0, unsigned k0, unsigned g1) {
  const unsigned blk = blockIdx.x;
  const int side = blk >= k0 ? 1 : 0;
  const unsigned base = side ? k0 : 0u, gend = side ? g1 : g0;
  const Kenc0Args& k = a.k[side];
  if (blk < gend) {
    const long e = (long)(blk - base) * 256 + threadIdx.x;
    const long total = (long)k.B * k.Np * a.d;
    if (e >= total) return;
    const int c = (int)(e % a.d);
    const long row = e / a.d;
    const int i = (int)(row % k.Np), b = (int)(row / k.Np);
    a.xrow[side][e] = i < k.N ? a.desc[side][b * a.sb[side] + c * a.sc[side] + i * a.sn[side]] : 0.f;
  } else {
    kenc0_element(k, (long)(blk - gend) * 256 + threadIdx.x);
  }
}

// ------------------------------------------------------------------ Sinkhorn
// Implicit couplings (:157-160): Z[i][j] = S[i][j] (i<m, j<n), alpha on the dustbin row/column.
// u has m+1 entries (u[m] = dustbin row), v has n+1.
__device__ __forceinline__ void counts(const SinkhornArgs& a, int b, int& m, int& n) {
  m = a.n0 ? a.n0[b] : a.N0;
  n = a.n1 ? a.n1[b] : a.N1;
}

// one wave per row i in [0, m]:  u[i] = log_mu[i] - logsumexp_j(Z[i][j] + v[j])      (:145)
__global__ __launch_bounds__(256) void sinkhorn_rows(SinkhornArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  int m, n;
  counts(a, b, m, n);
  if (m == 0 || n == 0 || i > m) return;
  const float* v = a.v + (size_t)b * (a.N1p + 1);
  LSE acc{-INFINITY, 0.f};
  if (i < m) {
    const float* Srow = a.S + ((size_t)b * a.N0p + i) * a.N1p;
    for (int j = lane; j < n; j += 64) lse_add(acc, Srow[j] + v[j]);
  } else {
    for (int j = lane; j < n; j += 64) lse_add(acc, a.alpha + v[j]);
  }
  if (lane == 0) lse_add(acc, a.alpha + v[n]);
  acc = wave_lse(acc);
  if (lane == 0) {
    const float norm = -logf((float)(m + n));                                          // (:162)
    const float log_mu = i < m ? norm : logf((float)n) + norm;                         // (:163)
    a.u[(size_t)b * (a.N0p + 1) + i] = log_mu - lse_value(acc);
  }
}

// block = 64 columns x 16 row groups:  v[j] = log_nu[j] - logsumexp_i(Z[i][j] + u[i])  (:146)
__global__ __launch_bounds__(1024) void sinkhorn_cols(SinkhornArgs a) {
  __shared__ float pm[16][64], ps[16][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const int b = blockIdx.y;
  int m, n;
  counts(a, b, m, n);
  if (m == 0 || n == 0) return;
  const float* u = a.u + (size_t)b * (a.N0p + 1);
  LSE acc{-INFINITY, 0.f};
  if (j < n) {
    const float* Scol = a.S + (size_t)b * a.N0p * a.N1p + j;
    for (int i = g; i < m; i += 16) lse_add(acc, Scol[(size_t)i * a.N1p] + u[i]);
  } else if (j == n) {
    for (int i = g; i < m; i += 16) lse_add(acc, a.alpha + u[i]);
  }
  if (g == 0 && j <= n) lse_add(acc, a.alpha + u[m]);
  pm[g][c] = acc.m;
  ps[g][c] = acc.s;
  __syncthreads();
  if (g == 0 && j <= n) {
    LSE t{pm[0][c], ps[0][c]};
#pragma unroll
    for (int k = 1; k < 16; ++k) t = lse_merge(t, LSE{pm[k][c], ps[k][c]});
    const float norm = -logf((float)(m + n));
    const float log_nu = j < n ? norm : logf((float)m) + norm;                         // (:164)
    a.v[(size_t)b * (a.N1p + 1) + j] = log_nu - lse_value(t);
  }
}

// ---- v[j] = log_nu[j] - logsumexp over the slab groups' partial (max, sum) pairs, for one block of 64 columns of one pair: the body of
// sinkhorn_vmerge, shared with the slab kernel's fused tail (round 6).  NT threads = 64 columns x GG chains; the arithmetic is that of 16
// chains per column whatever GG is (a thread walks chains g, g + GG, ...), so both callers produce the same bits.
// COH: the partials and u were written by OTHER workgroups of the same launch -- agent-scope (sc1) loads, which do not hit a stale line
// of this XCD's L2.
template <bool COH>
__device__ __forceinline__ float2 ld_part(const float2* p) {
  if constexpr (COH) {
    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (unsigned)w), __builtin_bit_cast(float, (unsigned)(w >> 32)));
  } else {
    return *p;
  }
}
template <bool COH>
__device__ __forceinline__ float ld_f(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
// pm / ps: [16][64] floats each, dm / ds: [16] each (LDS).  All NT threads of the workgroup call it (barriers inside).
template <int GG, bool COH>
__device__ __forceinline__ void vmerge_block(const SinkhornArgs& a, const float* __restrict__ part, int ngroup_max, int b, int m, int n, int ngroups,
                                             int cb, float* pm, float* ps, float* dm, float* ds) {
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6, j = cb * 64 + c;
  // two-pass merges: the maximum of the (max, sum) pairs first, then ONE exp per pair -- a chain of pairwise lse_merge()
  // spends two exps (quarter-rate instructions) and two selects per pair
  for (int vg = g; vg < 16; vg += GG) {
    LSE t{-INFINITY, 0.f};
    if (j < n) {
      const float2* pb = reinterpret_cast<const float2*>(part + (size_t)b * ngroup_max * (a.N1p + 1) * 2) + j;
      for (int s0 = vg; s0 < ngroups; s0 += 64) {
        float2 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int sl = s0 + 16 * k;
          q[k] = sl < ngroups ? ld_part<COH>(pb + (size_t)sl * (a.N1p + 1)) : make_float2(-INFINITY, 0.f);
        }
        const float nm = fmaxf(fmaxf(t.m, fmaxf(q[0].x, q[1].x)), fmaxf(q[2].x, q[3].x));
        if (nm > -INFINITY) {
          float sum = t.s * __builtin_amdgcn_exp2f((t.m - nm) * LOG2E);          // exp2(-inf) = 0 for an empty accumulator
#pragma unroll
          for (int k = 0; k < 4; ++k) sum = fmaf(q[k].y, __builtin_amdgcn_exp2f((q[k].x - nm) * LOG2E), sum);
          t = LSE{nm, sum};
        }
      }
    }
    pm[vg * 64 + c] = t.m;
    ps[vg * 64 + c] = t.s;
  }
  // The dustbin column j = n holds alpha in every row i <= m: its log-sum-exp over (alpha + u[i]) needs no S and no slab partial (round 6:
  // thread 0 of every slab workgroup used to walk it, 8 LDS reads and ~80 instructions per slab on the wave the others wait for).  The
  // block that owns column n reduces it here: 64-row chunks, chunk ch to chain ch % 16 (wave maximum and sum, two-pass), chains merged below.
  const bool dust_block = cb == n / 64;                    // (workgroup-uniform)
  if (dust_block) {
    const float* u = a.u + (size_t)b * (a.N0p + 1);
    for (int vw = g; vw < 16; vw += GG) {
      LSE d{-INFINITY, 0.f};
      for (int i0 = vw * 64; i0 <= m; i0 += 1024) {
        const int i = i0 + c;
        const float x = i <= m ? a.alpha + ld_f<COH>(u + i) : -INFINITY;
        const float mx = wave_max_u(x);
        if (mx > -INFINITY) {                              // (wave-uniform)
          const float sum = wave_sum_u(__builtin_amdgcn_exp2f((x - mx) * LOG2E));
          d = lse_merge(d, LSE{mx, sum});
        }
      }
      if (c == 0) { dm[vw] = d.m; ds[vw] = d.s; }
    }
  }
  __syncthreads();
  if (g == 0 && j <= n) {
    float nm = pm[c];
#pragma unroll
    for (int k = 1; k < 16; ++k) nm = fmaxf(nm, pm[k * 64 + c]);
    float sum = 0.f;                                                               // (nm finite for j < n: group 0 always contributes)
#pragma unroll
    for (int k = 0; k < 16; ++k) sum = fmaf(ps[k * 64 + c], __builtin_amdgcn_exp2f((pm[k * 64 + c] - nm) * LOG2E), sum);
    LSE t{nm, sum};
    if (j == n) {
      t = LSE{dm[0], ds[0]};
#pragma unroll
      for (int k = 1; k < 16; ++k) t = lse_merge(t, LSE{dm[k], ds[k]});
    }
    const float norm = -logf((float)(m + n));
    const float log_nu = j < n ? norm : logf((float)m) + norm;
    a.v[(size_t)b * (a.N1p + 1) + j] = log_nu - lse_value(t);
  }
  __syncthreads();                                         // (pm / ps / dm / ds are reused by the caller's next block)
}

// Slab form of one Sinkhorn iteration: a workgroup (16 waves) keeps R rows of S in LDS, computes their
// u (row pass, :145) and immediately the partial column log-sum-exps of (Z + u) over those rows
// (:146), so S is read from HBM/L2 once per iteration instead of twice.  Both passes hold their operands in
// registers, so each log-sum-exp is two-pass (exact max, then one fma + v_exp per element).  A second, small kernel merges
// the per-slab partials into v.  Slab `m / R` also owns the dustbin row i = m.
// NW = waves per workgroup: 16 (a row split over 16/R waves) or, when a row fits one wave's batch (N1p <= 1024), 8 -- twice
// as many workgroups resident per CU to cover each other's load latency and barriers.
// Round 5: a workgroup walks G consecutive slabs and merges their column partials in registers before writing them: the partials
// were 2 x 68 MB of an iteration's traffic at C3 (written here, read by sinkhorn_vmerge) beside the 268 MB of S.
// PF ("sinkhorn_prefetch" = on; off by default): once a slab's row segment has been written to LDS, the wave loads ITS segment of the NEXT
// slab's row into the same registers, where it stays in flight through the reductions, both barriers and the column pass.  Same loads, same
// arithmetic: bit-identical.  Round 5's form only touched the lines (4.20 -> 4.03 ms at C5 with ONE workgroup resident per CU); with two
// resident (below) neither form pays: C5 3.78 ms without, 3.95 with (G = 2, equal occupancy), C3 1.98 / 1.99 -- the other workgroups of
// the CU already cover the latency, and the 16 registers cost the G = 4 form its eighth wave (profiles/r06_sinkhorn_trace.txt, sk22).
// Scalar registers capped at 80 (round 6): a CU holds eight waves per SIMD only while a wave's scalar allocation is <= 80 -- measured with
// the workgroup-life stamps below (tools/sinkhorn_trace.py, profiles/r06_sinkhorn_trace.txt): at 82-87 scalar registers the compiler still
// reports occupancy 8, but the chip ran 3 of the 8-wave workgroups per CU at C3 instead of 4 and ONE of the 16-wave workgroups at C5
// instead of two (the C5 launch was two rounds of 256 workgroups).  With the cap the few spilled scalars cost one vector register
// (hence `cdust` in LDS: 64 -> 62 registers before the cap) and the C5 iteration went from 34 to 29 us (8 pairs), the C3 one from 60 to 54.
#ifndef SK_SGPRS
#define SK_SGPRS 80
#endif
#define SK_SGPR_ATTR __attribute__((amdgpu_num_sgpr(SK_SGPRS)))
template <int R, int NW, int G, bool PF, bool FM>
__global__ __launch_bounds__(64 * NW) SK_SGPR_ATTR void sinkhorn_slab(SinkhornArgs a, float* __restrict__ part, int ngroup_max) {   // (eight waves per SIMD: <= 64 registers)
  // (16-byte aligned: the dynamic array starts behind the static ones, and at an 8-byte offset the b128 accesses of `tile` run at half
  // rate -- measured: two more static floats took the C5 iteration from 3.98 to 6.26 ms)
  extern __shared__ __align__(16) float sm[];
  float* tile = sm;                       // [R][N1p]
  float* vs = tile + R * a.N1p;           // [N1p + 1]
  __shared__ float uu[R];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: row pointers in scalar registers)
  const int b = blockIdx.y, grp = blockIdx.x;
  int m, n;
  counts(a, b, m, n);
  if (m == 0 || n == 0 || grp * G * R > m) return;
#ifdef SK_TRACE
  // workgroup lives: the 100-MHz constant clock at entry, after each slab and at exit, and where the workgroup ran
  unsigned long long* const trw = a.trace ? a.trace + ((size_t)b * ngroup_max + grp) * 8 : nullptr;
  if (trw && tid == 0) {
    trw[0] = __builtin_amdgcn_s_memrealtime();
    trw[6] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);   // XCC_ID | HW_ID
  }
#endif
  const float* v = a.v + (size_t)b * (a.N1p + 1);
  for (int j = tid; j <= n; j += 64 * NW) vs[j] = v[j];
  __syncthreads();
  // (scalars: left to the compiler, both logarithms were re-evaluated by every wave in every slab -- 35 instructions of a slab's 270)
  const float norm = uniform_f(-logf((float)(m + n)));
  const float log_mu_dust = uniform_f(logf((float)n) + norm);
  constexpr int MAXC = R == 4 ? 4 : 2;    // real columns per thread: N1p / (64 NW) -- N1p <= 1024 with 8 waves, <= 2048 with 16 (R >= 8), <= 4096 with 16 (R = 4)
  LSE cacc[MAXC];                         // (the dustbin column j = n needs no S: sinkhorn_vmerge takes it from u, round 6)
#pragma unroll
  for (int c = 0; c < MAXC; ++c) cacc[c] = LSE{-INFINITY, 0.f};
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  [[maybe_unused]] f32x4 xr[4];           // the fast row path's batch of S; with PF it may already hold the NEXT slab's segment
  [[maybe_unused]] bool have = false;     // (PF) xr was loaded for this slab by the previous one (wave-uniform)
  // (G is a template parameter and the loop fully unrolled: as a run-time loop hipcc keeps 40 more vector and 50 more scalar registers
  // across the slabs -- 81 / 105 against 43 / 58 -- and the kernel falls off its eight waves per SIMD)
#pragma unroll
  for (int g = 0; g < G; ++g) {
  const int slab = grp * G + g, i0 = slab * R;
  if (i0 > m) break;                      // (block-uniform)
  if (g) __syncthreads();                 // the previous slab's column pass has read tile / uu
  // Row pass: all 16 waves work whatever R is -- a row is split over W = 16/R waves (segments of N1p/W <= 1024
  // columns, one batch of four float4 loads per lane); their partial (max, sum) pairs are merged through LDS.
  constexpr int W = NW / R;
  __shared__ float pm[NW], ps[NW];
  {
    const int r = wave / W, seg = wave % W, i = i0 + r;
    const int seglen = a.N1p / W, jlo = seg * seglen, jhi = jlo + seglen;     // N1p % 32 == 0: float4-aligned segments
    LSE acc{-INFINITY, 0.f};
    if (i < m) {
      // four float4 loads per lane are issued before any of them is consumed, so the wave keeps 4 KB in flight
      const float* Srow = a.S + ((size_t)b * a.N0p + i) * a.N1p;
      float* trow = tile + r * a.N1p;
      // The segment (<= 1024 columns) is one batch of four float4 per lane, so the row's values sit in registers and the
      // log-sum-exp can be the two-pass form (exact maximum first, then ONE fma + v_exp per element) instead of the
      // online form's ~8 VALU per element; the maximum is wave-wide, as in torch.logsumexp.
      if (seglen == 1024 && n == a.N1p) {
        // Fast path (uniform): this wave's segment is exactly one batch with every column real (the saturated max_keypoints case):
        // no bounds or count masks, and the adds / fmas / partial sums as packed f32x4 operations -- on this SIMD every VALU
        // instruction costs matrix-pipe-free but real issue time, and the masked form spends a third of its instructions on
        // compares and selects.
        f32x4 t4[4];
        if (!(PF && have)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) xr[k] = *reinterpret_cast<const f32x4*>(Srow + jlo + k * 256 + lane * 4);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = jlo + k * 256 + lane * 4;
          *reinterpret_cast<f32x4*>(trow + j) = xr[k];
          t4[k] = xr[k] + *reinterpret_cast<const f32x4*>(vs + j);
          mx = fmaxf(fmaxf(mx, fmaxf(t4[k][0], t4[k][1])), fmaxf(t4[k][2], t4[k][3]));
        }
        if constexpr (PF && G > 1) {
          // xr is dead: the NEXT slab's segment of this wave goes into it now and stays in flight through the reductions, both barriers
          // and the column pass (round 6; the round-5 form only touched the lines, which left the L2 -> register latency after the barrier)
          have = g + 1 < G && i + R < m;
          __builtin_amdgcn_sched_barrier(0);         // (not before xr's last use: hoisted, the loads take a second set of registers)
          if (have) {
#pragma unroll
            for (int k = 0; k < 4; ++k) xr[k] = *reinterpret_cast<const f32x4*>(Srow + (size_t)R * a.N1p + jlo + k * 256 + lane * 4);
          }
        }
        mx = wave_max_u(mx);
        const float ml = -mx * LOG2E;                 // mx is finite: real scores
        const f32x4 l2e = {LOG2E, LOG2E, LOG2E, LOG2E}, ml4 = {ml, ml, ml, ml};
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 a4 = __builtin_elementwise_fma(t4[k], l2e, ml4);
          s4 += (f32x4){__builtin_amdgcn_exp2f(a4[0]), __builtin_amdgcn_exp2f(a4[1]), __builtin_amdgcn_exp2f(a4[2]), __builtin_amdgcn_exp2f(a4[3])};
        }
        float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        sum = wave_sum_u(sum);
        acc = LSE{mx, lse_unbias(sum, mx, ml)};
      } else if (jlo < n) {
        have = false;
        float4 x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = jlo + k * 256 + lane * 4;
          x[k] = j < jhi ? *reinterpret_cast<const float4*>(Srow + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float t[16];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = jlo + k * 256 + lane * 4;
          float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < jhi) {
            *reinterpret_cast<float4*>(trow + j) = x[k];
            vv = *reinterpret_cast<const float4*>(vs + j);
          }
          t[4 * k + 0] = (j < jhi && j + 0 < n) ? x[k].x + vv.x : -INFINITY;
          t[4 * k + 1] = (j < jhi && j + 1 < n) ? x[k].y + vv.y : -INFINITY;
          t[4 * k + 2] = (j < jhi && j + 2 < n) ? x[k].z + vv.z : -INFINITY;
          t[4 * k + 3] = (j < jhi && j + 3 < n) ? x[k].w + vv.w : -INFINITY;
#pragma unroll
          for (int c = 0; c < 4; ++c) mx = fmaxf(mx, t[4 * k + c]);
        }
        mx = wave_max_u(mx);
        float sum = 0.f;
        if (mx > -INFINITY) {
          const float ml = -mx * LOG2E;
#pragma unroll
          for (int e = 0; e < 16; ++e) sum += __builtin_amdgcn_exp2f((t[e] - mx) * LOG2E);      // exp(t - mx); -inf -> 0
          (void)ml;
        }
        sum = wave_sum_u(sum);
        acc = LSE{mx, sum};
      }
    } else if (i == m) {
      for (int j = jlo + lane; j < jhi && j < n; j += 64) lse_add(acc, a.alpha + vs[j]);
    }
    if (i == m) acc = wave_lse(acc);             // (real rows are already wave-reduced)
    if (i <= m && seg == W - 1 && lane == 0) lse_add(acc, a.alpha + vs[n]);      // the dustbin column, once per row
    if (lane == 0) { pm[wave] = acc.m; ps[wave] = acc.s; }
  }
  __syncthreads();
  if (tid < R && i0 + tid <= m) {
    LSE t{pm[tid * W], ps[tid * W]};
#pragma unroll
    for (int k = 1; k < W; ++k) t = lse_merge(t, LSE{pm[tid * W + k], ps[tid * W + k]});
    const int i = i0 + tid;
    const float log_mu = i < m ? norm : log_mu_dust;
    const float ui = log_mu - lse_value(t);
    uu[tid] = ui;
    if constexpr (FM) __hip_atomic_store(a.u + (size_t)b * (a.N0p + 1) + i, ui, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by the pair's merging workgroup in this launch)
    else a.u[(size_t)b * (a.N0p + 1) + i] = ui;
  }
  __syncthreads();
  const int rows = min(R, m + 1 - i0);      // rows of this slab, the last may be the dustbin row
  // all R rows REAL (i0 + R <= m): `rows == R` alone also admits a slab whose last row is the dustbin row i == m
  // (m % R == R-1), and that row of `tile` is never written by the row pass
  const bool fastcol = i0 + R <= m && n == a.N1p;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  static_assert(R % 4 == 0, "slab rows in groups of four");
  const int ld = a.N1p;
  // the generic column: masks for the rows past m, the dustbin row and the dustbin column (alpha)
  auto generic_col = [&](int j) __attribute__((always_inline)) -> LSE {
    float tt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tt[r] = j < n ? tile[r * a.N1p + j] : 0.f;     // rows beyond `rows` hold stale data, masked below
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool real = (i0 + r < m) && (j < n);
      tt[r] = r < rows ? (real ? tt[r] : a.alpha) + uu[r] : -INFINITY;
      mx = fmaxf(mx, tt[r]);
    }
    float sum = 0.f;                                                  // rows >= 1: mx is finite
#pragma unroll
    for (int r = 0; r < R; ++r) sum += __builtin_amdgcn_exp2f((tt[r] - mx) * LOG2E);
    return LSE{mx, sum};
  };
  if (fastcol) {
    // Fast path (uniform): a full slab of real rows, every column real -- no masks
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      __builtin_amdgcn_sched_barrier(0);       // (one column at a time: interleaved, the unrolled bodies cost 40 registers and the eighth wave per SIMD)
      const int j = tid + c * 64 * NW;
      if (j < n) {
        f32x4 t4[R / 4];
        float mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < R / 4; ++q) {
          t4[q] = (f32x4){tile[(4 * q + 0) * ld + j], tile[(4 * q + 1) * ld + j], tile[(4 * q + 2) * ld + j], tile[(4 * q + 3) * ld + j]} +
                  (f32x4){uu[4 * q + 0], uu[4 * q + 1], uu[4 * q + 2], uu[4 * q + 3]};
          mx = fmaxf(fmaxf(mx, fmaxf(t4[q][0], t4[q][1])), fmaxf(t4[q][2], t4[q][3]));
        }
        const float ml = -mx * LOG2E;
        const f32x4 l2e = {LOG2E, LOG2E, LOG2E, LOG2E}, ml4 = {ml, ml, ml, ml};
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < R / 4; ++q) {
          const f32x4 a4 = __builtin_elementwise_fma(t4[q], l2e, ml4);
          s4 += (f32x4){__builtin_amdgcn_exp2f(a4[0]), __builtin_amdgcn_exp2f(a4[1]), __builtin_amdgcn_exp2f(a4[2]), __builtin_amdgcn_exp2f(a4[3])};
        }
        const LSE t{mx, lse_unbias((s4[0] + s4[1]) + (s4[2] + s4[3]), mx, ml)};
        cacc[c] = lse_merge1(cacc[c], t);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      __builtin_amdgcn_sched_barrier(0);
      const int j = tid + c * 64 * NW;
      if (j < n) {
        const LSE t = generic_col(j);
        cacc[c] = lse_merge1(cacc[c], t);
      }
    }
  }
#ifdef SK_TRACE
  if (trw && tid == 0 && g < 4) trw[1 + g] = __builtin_amdgcn_s_memrealtime();
#endif
  }  // slabs of the group
  float2* pb = reinterpret_cast<float2*>(part + ((size_t)b * ngroup_max + grp) * (a.N1p + 1) * 2);
  if constexpr (!FM) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int j = tid + c * 64 * NW;
      if (j < n) pb[j] = make_float2(cacc[c].m, cacc[c].s);
    }
  } else {
    // Fused merge ("sinkhorn_merge" = fused; round 6, VERDICT r5 next 5: one launch per iteration) -- built, bit-identical to the merge
    // kernel, and SLOWER: C3 x 64 pairs 1.70 -> 4.24 ms per 30 iterations, C5 x 8 pairs 3.22 -> 4.70 per 100 (profiles/r06_sinkhorn_trace.txt,
    // sk27); not the default.  The pair's workgroups take a ticket as they finish; the last min(groups, column blocks) arrivals stay, wait
    // until every group of the pair has arrived and merge one block of 64 columns each (more when there are fewer groups than blocks).
    // Data written by one workgroup and read by another in the same launch moves through agent-scope (sc1) stores and loads -- the
    // partials here, u in the row pass -- and the ticket is taken after this workgroup's stores have been acknowledged (s_waitcnt
    // vmcnt(0) + barrier): no L2 write-back fence.  Why it loses: half of a pair's workgroups now end their lives as mergers -- eight or
    // sixteen waves holding a slab slot while 64 x 16 chains wait on sc1 loads that are served past the L2 -- and every workgroup pays the
    // ticket's round trip; sinkhorn_vmerge does the same work in 8 us with the whole chip, from L2.
    // No deadlock: workgroups are dispatched in block order (x fastest: pair by pair) per XCD, so whatever a waiting workgroup waits
    // for is either running or at the head of a queue whose slots are held by workgroups that do finish -- the first
    // groups - min(groups, blocks) arrivals of every pair never wait.  The spin is bounded all the same (word [B] of the counters).
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int j = tid + c * 64 * NW;
      if (j < n) {
        const unsigned long long w = (unsigned long long)__builtin_bit_cast(unsigned, cacc[c].m) | ((unsigned long long)__builtin_bit_cast(unsigned, cacc[c].s) << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(pb + j), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned ticket;
    const int ng = m / (G * R) + 1;                         // the pair's groups that run (grp * G * R <= m)
    const unsigned base = (unsigned)a.it * (unsigned)ng;
    if (tid == 0) ticket = __hip_atomic_fetch_add(a.merge_cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int my = (int)(ticket - base);                    // arrival index 0 .. ng - 1
    const int ncb = n / 64 + 1, nmrg = min(ng, ncb);        // blocks of 64 columns over 0 .. n; mergers
    if (my >= ng - nmrg) {
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(a.merge_cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base < (unsigned)ng) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 24)) { __hip_atomic_store(a.merge_cnt + a.B, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
      }
      __syncthreads();
      float* pmm = sm;                                      // (the tile is free: 2 x 4 KB + 128 bytes, launch_slab_g sizes the dynamic LDS for it)
      for (int cb = my - (ng - nmrg); cb < ncb; cb += nmrg)
        vmerge_block<NW, true>(a, part, ngroup_max, b, m, n, ng, cb, pmm, pmm + 1024, pmm + 2048, pmm + 2064);
    }
  }
#ifdef SK_TRACE
  if (trw && tid == 0) trw[5] = __builtin_amdgcn_s_memrealtime();
#endif
}

// v[j] = log_nu[j] - logsumexp over the slabs' partial (max, sum) pairs.  64 columns x 16 slab groups per
// workgroup; each thread loads its (up to 4 at a time) partials before merging them, so the loads overlap.
__global__ __launch_bounds__(1024) void sinkhorn_vmerge(SinkhornArgs a, const float* __restrict__ part, int nslab_max, int R, int G) {
  __shared__ float pm[16 * 64], ps[16 * 64], dm[16], ds[16];
  const int b = blockIdx.y;
  int m, n;
  counts(a, b, m, n);
  if (m == 0 || n == 0) return;
  const int nslab = (m / R + 1 + G - 1) / G;        // groups of G slabs, merged by sinkhorn_slab (nslab_max = groups per pair)
  if ((int)blockIdx.x * 64 > n) return;             // (columns 0 .. n)
  vmerge_block<16, false>(a, part, nslab_max, b, m, n, nslab, (int)blockIdx.x, pm, ps, dm, ds);
}

// ------------------------------------------------------------------ matches
// Z'[i][j] = ((S[i][j] + u[i]) + v[j]) - norm   (:147, :169) — same operation order as the reference.
__device__ __forceinline__ void mcounts(const MatchArgs& a, int b, int& m, int& n) {
  m = a.n0 ? a.n0[b] : a.N0;
  n = a.n1 ? a.n1[b] : a.N1;
}

__global__ __launch_bounds__(256) void match_rowmax(MatchArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  int m, n;
  mcounts(a, b, m, n);
  if (m == 0 || n == 0 || i >= m) return;
  const float norm = -logf((float)(m + n));
  const float ui = a.u[(size_t)b * (a.N0p + 1) + i];
  const float* v = a.v + (size_t)b * (a.N1p + 1);
  const float* Srow = a.S + ((size_t)b * a.N0p + i) * a.N1p;
  float best = -INFINITY;
  int bj = 0x7fffffff;
  for (int j = lane; j < n; j += 64) {
    const float z = ((Srow[j] + ui) + v[j]) - norm;
    if (z > best) { best = z; bj = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oj = __shfl_xor(bj, o);
    if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
  }
  if (lane == 0) { a.max0[(size_t)b * a.N0p + i] = best; a.idx0[(size_t)b * a.N0p + i] = bj; }
}

__global__ __launch_bounds__(1024) void match_colmax(MatchArgs a) {
  __shared__ float pv[16][64];
  __shared__ int pi[16][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const int b = blockIdx.y;
  int m, n;
  mcounts(a, b, m, n);
  if (m == 0 || n == 0) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (j < n) {
    const float norm = -logf((float)(m + n));
    const float vj = a.v[(size_t)b * (a.N1p + 1) + j];
    const float* u = a.u + (size_t)b * (a.N0p + 1);
    const float* Scol = a.S + (size_t)b * a.N0p * a.N1p + j;
    // eight rows' loads in flight per thread (one at a time, a column of 1024 rows was 64 dependent L2 round trips: 24.6 us at
    // one pair, where the grid is 16 workgroups); compared in row order: the first maximum wins, as Tensor.max does
    int i = g;
    for (; i + 16 * 7 < m; i += 16 * 8) {
      float sv[8], uv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sv[e] = Scol[(size_t)(i + 16 * e) * a.N1p]; uv[e] = u[i + 16 * e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = ((sv[e] + uv[e]) + vj) - norm;
        if (z > best) { best = z; bi = i + 16 * e; }
      }
    }
    for (; i < m; i += 16) {
      const float z = ((Scol[(size_t)i * a.N1p] + u[i]) + vj) - norm;
      if (z > best) { best = z; bi = i; }
    }
  }
  pv[g][c] = best;
  pi[g][c] = bi;
  __syncthreads();
  if (g == 0 && j < n) {
    for (int k = 1; k < 16; ++k) {
      const float ob = pv[k][c];
      const int oi = pi[k][c];
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    a.max1[(size_t)b * a.N1p + j] = best;
    a.idx1[(size_t)b * a.N1p + j] = bi;
  }
}

__global__ __launch_bounds__(256) void match_finalize(MatchArgs a) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  int m, n;
  mcounts(a, b, m, n);
  const bool empty = (m == 0 || n == 0);                                               // (:235-242)
  const int* idx0 = a.idx0 + (size_t)b * a.N0p;
  const int* idx1 = a.idx1 + (size_t)b * a.N1p;
  const float* max0 = a.max0 + (size_t)b * a.N0p;
  if (t < a.N0) {
    long mi = -1;
    float ms = 0.f;
    if (!empty && t < m) {
      const int j = idx0[t];
      const bool mutual = idx1[j] == t;                                                // (:270)
      ms = mutual ? expf(max0[t]) : 0.f;                                               // (:273)
      if (mutual && ms > a.threshold) mi = j;                                          // (:275,277)
    }
    a.matches0[(size_t)b * a.N0 + t] = mi;
    a.ms0[(size_t)b * a.N0 + t] = ms;
  }
  if (t < a.N1) {
    long mi = -1;
    float ms = 0.f;
    if (!empty && t < n) {
      const int i = idx1[t];
      const bool mutual = idx0[i] == t;                                                // (:271)
      // mutual1 implies mutual0 at i, so mscores0[i] = exp(max0[i])                   (:274)
      ms = mutual ? expf(max0[i]) : 0.f;
      if (mutual && ms > a.threshold) mi = i;                                          // (:276,278)
    }
    a.matches1[(size_t)b * a.N1 + t] = mi;
    a.ms1[(size_t)b * a.N1 + t] = ms;
  }
}

}  // namespace

// rows per LDS slab (R * N1p floats <= 64 KB); 0 = use the two-pass kernels
int sinkhorn_slab_rows(int N1p) { return N1p <= 2048 ? 8 : N1p <= 4096 ? 4 : 0; }   // 16-row slabs measured slower (3.19 vs 2.36 ms at N = 1024)   // 8 rows even when 16 fit: 4 workgroups per CU (36 KB) hide the load latency better


hipError_t launch_sg_prologue(const SgPrologueArgs& a, hipStream_t s) {
  auto blocks = [](long n) { return (unsigned)((n + 255) / 256); };
  const unsigned g0 = blocks((long)a.k[0].B * a.k[0].Np * a.d), k0 = g0 + blocks((long)a.k[0].B * a.k[0].Np * a.k[0].C1);
  const unsigned g1 = k0 + blocks((long)a.k[1].B * a.k[1].Np * a.d), k1 = g1 + blocks((long)a.k[1].B * a.k[1].Np * a.k[1].C1);
  hipLaunchKernelGGL(sg_prologue_kernel, dim3(k1), dim3(256), 0, s, a, g0, k0, g1);
  return hipGetLastError();
}

hipError_t launch_gather_desc(const float* src, int64_t sb, int64_t sc, int64_t sn, int B, int N, int Np, int d,
                              float* out, hipStream_t s) {
  long total = (long)B * Np * d;
  hipLaunchKernelGGL(gather_desc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, (long)sb, (long)sc,
                     (long)sn, B, N, Np, d, out);
  return hipGetLastError();
}

template <int R, int NW, int G, bool PF, bool FM = false>
static void launch_slab_g(const SinkhornArgs& a, int nslab_max, hipStream_t s) {
  size_t lds = ((size_t)R * a.N1p + a.N1p + 1) * sizeof(float);
  if (FM && lds < 2080 * sizeof(float)) lds = 2080 * sizeof(float);      // the fused merge's chains live in the tile's LDS
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(sinkhorn_slab<R, NW, G, PF, FM>), 96 * 1024, attr);
  hipLaunchKernelGGL((sinkhorn_slab<R, NW, G, PF, FM>), dim3((unsigned)nslab_max, (unsigned)a.B), dim3(64 * NW), lds, s, a, a.part, nslab_max);
}
template <int R, int NW>
static void launch_slab_k(const SinkhornArgs& a, int nslab_max, int G, hipStream_t s) {
  // Round 6: with the scalar-register cap two 16-wave workgroups ARE resident per CU and cover each other: the next-slab prefetch is off
  // unless asked for (measurements at the kernel's head comment).  The fused merge ("sinkhorn_merge" = fused) is its own instantiation:
  // as a run-time branch its code cost the default kernel 9 registers and 3 % of its time
  const bool p = a.prefetch > 0;
  if (a.merge_cnt) {
    if (G == 4) launch_slab_g<R, NW, 4, false, true>(a, nslab_max, s);
    else if (G == 2) launch_slab_g<R, NW, 2, false, true>(a, nslab_max, s);
    else launch_slab_g<R, NW, 1, false, true>(a, nslab_max, s);
    return;
  }
  if (G == 4) { if (p) launch_slab_g<R, NW, 4, true>(a, nslab_max, s); else launch_slab_g<R, NW, 4, false>(a, nslab_max, s); }
  else if (G == 2) { if (p) launch_slab_g<R, NW, 2, true>(a, nslab_max, s); else launch_slab_g<R, NW, 2, false>(a, nslab_max, s); }
  else launch_slab_g<R, NW, 1, false>(a, nslab_max, s);
}

template <int R>
static void launch_slab_iter(const SinkhornArgs& a, int nslab_max, int G, hipStream_t s) {
  if constexpr (R == 8) {
    if (a.N1p <= 1024) launch_slab_k<8, 8>(a, nslab_max, G, s);  // a row fits one wave's batch: 8-wave workgroups, four per CU
    else launch_slab_k<8, 16>(a, nslab_max, G, s);
  } else {
    launch_slab_k<R, 16>(a, nslab_max, G, s);
  }
  if (!a.merge_cnt) hipLaunchKernelGGL(sinkhorn_vmerge, dim3((unsigned)((a.N1p + 1 + 63) / 64), (unsigned)a.B), dim3(1024), 0, s, a, a.part, nslab_max, R, G);
}

hipError_t launch_sinkhorn(const SinkhornArgs& a, hipStream_t s) {
  // v = 0 (:143), and u (its padding entries are never written).  The caller allocates them back to back (u, then v): one stream
  // operation instead of two (each is ~4.7 us on the single-pair path)
  hipError_t e;
  if (a.v == a.u + (size_t)a.B * (a.N0p + 1)) {
    e = hipMemsetAsync(a.u, 0, ((size_t)a.B * (a.N0p + 1) + (size_t)a.B * (a.N1p + 1)) * sizeof(float), s);
    if (e != hipSuccess) return e;
  } else {
    e = hipMemsetAsync(a.v, 0, (size_t)a.B * (a.N1p + 1) * sizeof(float), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.u, 0, (size_t)a.B * (a.N0p + 1) * sizeof(float), s);
    if (e != hipSuccess) return e;
  }
  const int R = sinkhorn_slab_rows(a.N1p);
  if (R > 0 && a.part) {
    // G slabs per workgroup (their column partials merged in registers): where a pair has enough slabs to fill its share of the
    // chip anyway (the handle option "sinkhorn_group" overrides: the A/B switch of the parity tests; 1 = a partial per slab, as before round 5)
    // (measured, tools/sinkhorn_time.py: C3, 129 slabs of 1024 columns x 64 pairs: 2.28 / 2.02 / 2.03 ms per 30 iterations with 1 / 2 / 4
    // slabs per workgroup; C5, 257 slabs of 2048 columns x 8 pairs: 4.49 / 4.63 / 4.18 per 100)
    // (round 6, eight waves per SIMD resident: C3 x 64 pairs 1.99 / 1.82 ms per 30 iterations with 2 / 4 slabs per workgroup -- four where
    // the groups of four still fill the 1024 workgroup slots of the 8-wave form)
    // the most slabs per workgroup (4, 2, 1) whose groups still fill the resident workgroup slots (1024 of the 8-wave form, 512 of the
    // 16-wave one): C3 x 1 pair 0.276 / 0.320 / 0.445 ms with 1 / 2 / 4, x 16 pairs 0.755 / 0.686 / 0.743, x 64 pairs - / 2.01 / 1.85
    const int nsl = a.N0p / R + 1;
    const long slots = a.N1p <= 1024 ? 1024 : 512;
    int G = (long)((nsl + 3) / 4) * a.B >= slots ? 4 : (long)((nsl + 1) / 2) * a.B >= slots ? 2 : 1;
    if (a.group == 1 || a.group == 2 || a.group == 4) G = a.group;
    const int nslab_max = (a.N0p / R + 1 + G - 1) / G;           // groups per pair = the partial buffer's rows per pair (<= N0p / R + 1: a.part's size)
    // (Round 5: walking the batch in groups whose score matrices fit the 256-MB Infinity Cache -- all iterations of a group back to
    // back -- was measured and is SLOWER: 2.22 ms for the 64 C3 pairs in one group, 2.58 / 2.88 / 3.16 / 4.25 ms with groups of 150 /
    // 112 / 72 / 40 MB.  An iteration of 64 pairs is 73 us over two launches: the loop is paced by launches and their tails, not by
    // the 268 MB an iteration reads, and smaller groups only multiply the launches.)
    SinkhornArgs ai = a;
    if (ai.merge_cnt) {
      e = hipMemsetAsync(ai.merge_cnt, 0, ((size_t)a.B + 1) * sizeof(unsigned), s);
      if (e != hipSuccess) return e;
    }
    for (int it = 0; it < a.iters; ++it) {
      ai.it = it;
      if (R == 16) launch_slab_iter<16>(ai, nslab_max, G, s);
      else if (R == 8) launch_slab_iter<8>(ai, nslab_max, G, s);
      else launch_slab_iter<4>(ai, nslab_max, G, s);
    }
    return hipGetLastError();
  }
  dim3 gr((unsigned)((a.N0p + 1 + 3) / 4), (unsigned)a.B);
  dim3 gc((unsigned)((a.N1p + 1 + 63) / 64), (unsigned)a.B);
  for (int it = 0; it < a.iters; ++it) {
    hipLaunchKernelGGL(sinkhorn_rows, gr, dim3(256), 0, s, a);
    hipLaunchKernelGGL(sinkhorn_cols, gc, dim3(1024), 0, s, a);
  }
  return hipGetLastError();
}

// One record row per pair (image-matching_amd/shard.py): [pair_id | n0 | n1 | kpts0 2K | kpts1 2K | matches0 K | matches1 K |
// mscores0 K | mscores1 K] as 32-bit words -- floats as bit patterns, int64 match indices narrowed to int32.  Rows >= B are
// padding (pair_id -1, zeros).  One workgroup per row, coalesced 4-byte stores.
__global__ __launch_bounds__(256) void pack_records_kernel(PackArgs a) {
  const int row = blockIdx.x, K = a.K, w = 3 + 8 * K;
  int* out = a.rec + (size_t)row * w;
  if (row >= a.B) {
    for (int i = threadIdx.x; i < w; i += 256) out[i] = i == 0 ? -1 : 0;
    return;
  }
  if (threadIdx.x == 0) { out[0] = a.pair_ids[row]; out[1] = a.counts0[row]; out[2] = a.counts1[row]; }
  const int* k0 = reinterpret_cast<const int*>(a.kpts0) + (size_t)row * 2 * K;
  const int* k1 = reinterpret_cast<const int*>(a.kpts1) + (size_t)row * 2 * K;
  const int* s0 = reinterpret_cast<const int*>(a.ms0) + (size_t)row * K;
  const int* s1 = reinterpret_cast<const int*>(a.ms1) + (size_t)row * K;
  const long long* m0 = a.matches0 + (size_t)row * K;
  const long long* m1 = a.matches1 + (size_t)row * K;
  for (int i = threadIdx.x; i < 2 * K; i += 256) { out[3 + i] = k0[i]; out[3 + 2 * K + i] = k1[i]; }
  for (int i = threadIdx.x; i < K; i += 256) {
    out[3 + 4 * K + i] = (int)m0[i];
    out[3 + 5 * K + i] = (int)m1[i];
    out[3 + 6 * K + i] = s0[i];
    out[3 + 7 * K + i] = s1[i];
  }
}

hipError_t launch_pack_records(const PackArgs& a, hipStream_t s) {
  if (a.rows < a.B || a.B < 0 || a.K <= 0) return hipErrorInvalidValue;
  if (a.rows == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)a.rows), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_matches(const MatchArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(match_rowmax, dim3((unsigned)((a.N0p + 3) / 4), (unsigned)a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(match_colmax, dim3((unsigned)((a.N1p + 63) / 64), (unsigned)a.B), dim3(1024), 0, s, a);
  const int nmax = a.N0 > a.N1 ? a.N0 : a.N1;
  hipLaunchKernelGGL(match_finalize, dim3((unsigned)((nmax + 255) / 256), (unsigned)a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace imx
