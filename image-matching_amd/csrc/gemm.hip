// gemm.hip — row-major fp32 GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// launch_gemm        the fp32-MFMA form of the linear layers: the A/B reference of the parity tests ("mfma" = "f32") and the
//                    fallback for shapes gemm_x3.hip rejects.
//                    out = act([a0|a1] . W + bias) (+ res): every 1x1 Conv2d / Conv1d(k=1) of the
//                    path — convPb/convDb (superpoint_test.py:78,83), the MLPs, projections and
//                    merges of superglue_test.py:49-60,92-119,214-216 — with BatchNorm folded into
//                    W/bias and the torch.cat([x, message]) of :119 expressed as a K split.
// launch_score_gemm  scores = mdesc0^T mdesc1 / sqrt(d)  (superglue_test.py:259-260).
//
// Workgroup = 256 threads -> 128 rows x NT columns; wave w owns rows [32w, 32w+32) x NT/32 N-blocks.
// K in chunks of 32: A tile [128][33] (odd stride: conflict-free 32-lane column reads),
// W tile [32][NT] (row reads, conflict free).
#include "imx_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BM = 128, CK = 32, SA = CK + 1;

template <int NT>
__global__ __launch_bounds__(256) void gemm_mfma(GemmArgs p) {
  constexpr int NB = NT / 32;
  constexpr int A_IT = BM * (CK / 4) / 256;   // 4 float4 per thread
  constexpr int W_IT = CK * NT / 4 / 256;     // 2 or 4
  constexpr int BUF = BM * SA + CK * NT;      // floats per stage (A tile then W tile; 4224 is 16-B aligned)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * BM, n0 = blockIdx.y * NT;
  const int K = p.K0 + p.K1, nchunk = K / CK;

  // two-level accumulation: the fp32 MFMA accumulates like one sequential fma chain (tools/ubench/mfma_round.hip), so blocks of
  // 128 k (four chunks) start from zero and are folded into `tot` (K = 512 at C5: 1.5x -> 1.2x of the reference's own fp32 error)
  f32x16 acc[NB], tot[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[n][r] = 0.f; tot[n][r] = 0.f; }

  // register-staged double buffering: chunk c+1 is fetched from L2/HBM while chunk c feeds the MFMAs.
  // Per-thread source pointers are set up once (rows >= M read a clamped row: their results are never stored), so a
  // chunk's fetch is 8 unconditional loads + pointer increments -- no zero-fill, bounds branches or 64-bit multiplies
  // in the K loop.
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 ar0, ar1, ar2, ar3, wr0, wr1, wr2, wr3;
  static_assert(A_IT == 4 && (W_IT == 2 || W_IT == 4), "staging map");
  const float* asrc0[4]; const float* asrc1[4]; const float* wsrc[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid + it * 256, row = min(r0 + e / (CK / 4), p.M - 1), v4 = e % (CK / 4);
    asrc0[it] = p.a0 + (size_t)row * p.lda0 + 4 * v4;
    asrc1[it] = p.a1 ? p.a1 + (size_t)row * p.lda1 + 4 * v4 : asrc0[it];
    const int idx = (tid + it * 256) * 4, k = idx / NT, col = idx % NT;
    wsrc[it] = p.w + (size_t)k * p.Npad + n0 + col;
  }
#define IMX_GLOAD(c0_)                                                                              \
  {                                                                                                 \
    const int c0v = (c0_);                                                                          \
    const bool first = c0v < p.K0;                                                                  \
    const int cc = first ? c0v : c0v - p.K0;                                                        \
    ar0 = *reinterpret_cast<const f32x4*>((first ? asrc0[0] : asrc1[0]) + cc);                      \
    ar1 = *reinterpret_cast<const f32x4*>((first ? asrc0[1] : asrc1[1]) + cc);                      \
    ar2 = *reinterpret_cast<const f32x4*>((first ? asrc0[2] : asrc1[2]) + cc);                      \
    ar3 = *reinterpret_cast<const f32x4*>((first ? asrc0[3] : asrc1[3]) + cc);                      \
    const size_t wo = (size_t)c0v * p.Npad;                                                         \
    wr0 = *reinterpret_cast<const f32x4*>(wsrc[0] + wo);                                            \
    wr1 = *reinterpret_cast<const f32x4*>(wsrc[1] + wo);                                            \
    if constexpr (W_IT == 4) {                                                                      \
      wr2 = *reinterpret_cast<const f32x4*>(wsrc[2] + wo);                                          \
      wr3 = *reinterpret_cast<const f32x4*>(wsrc[3] + wo);                                          \
    }                                                                                               \
  }
#define IMX_SA(reg_, it_)                                                                           \
  {                                                                                                 \
    const int e = tid + (it_) * 256, row = e / (CK / 4), v4 = e % (CK / 4);                         \
    float* d = at + row * SA + 4 * v4;                                                              \
    d[0] = reg_[0]; d[1] = reg_[1]; d[2] = reg_[2]; d[3] = reg_[3];                                 \
  }
#define IMX_LSTORE(buf_)                                                                            \
  {                                                                                                 \
    float* at = smem + (buf_) * BUF;                                                                \
    float* wt = at + BM * SA;                                                                       \
    IMX_SA(ar0, 0) IMX_SA(ar1, 1) IMX_SA(ar2, 2) IMX_SA(ar3, 3)                                     \
    *reinterpret_cast<f32x4*>(wt + (tid + 0 * 256) * 4) = wr0;                                     \
    *reinterpret_cast<f32x4*>(wt + (tid + 1 * 256) * 4) = wr1;                                     \
    if constexpr (W_IT == 4) {                                                                      \
      *reinterpret_cast<f32x4*>(wt + (tid + 2 * 256) * 4) = wr2;                                   \
      *reinterpret_cast<f32x4*>(wt + (tid + 3 * 256) * 4) = wr3;                                   \
    }                                                                                               \
  }

  IMX_GLOAD(0)
  IMX_LSTORE(0)
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    // branch-free prefetch: the last iteration re-fetches its own chunk into the idle buffer (harmless)
    IMX_GLOAD((c + 1 < nchunk ? c + 1 : c) * CK)
    const float* a_tile = smem + (c & 1) * BUF;
    const float* w_tile = a_tile + BM * SA;
    const float* ap = a_tile + (32 * wave + (lane & 31)) * SA + (lane >> 5);
    const float* bp = w_tile + (lane >> 5) * NT + (lane & 31);
    // operand fragments are fetched from LDS one k-step ahead of the MFMAs that consume them
    float af[2], bf[2][NB];
    af[0] = ap[0];
#pragma unroll
    for (int n = 0; n < NB; ++n) bf[0][n] = bp[n * 32];
#pragma unroll
    for (int kk = 0; kk < CK / 2; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < CK / 2) {
        af[nxt] = ap[2 * (kk + 1)];
#pragma unroll
        for (int n = 0; n < NB; ++n) bf[nxt][n] = bp[2 * (kk + 1) * NT + n * 32];
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
#pragma unroll
      for (int n = 0; n < NB; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur][n], acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((c & 3) == 3 || c + 1 == nchunk) {       // block-uniform: end of a 128-k block (or of the reduction)
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        tot[n] += acc[n];
        if (c + 1 < nchunk) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        } else {
          acc[n] = tot[n];
        }
      }
    }
    IMX_LSTORE((c + 1) & 1)
    __syncthreads();
  }
#undef IMX_GLOAD
#undef IMX_LSTORE
#undef IMX_SA

  // ---- epilogue through LDS (the operand buffers are free after the loop's last barrier): the
  //      accumulator layout gives a lane one column of 16 scattered rows, and per-lane dword stores
  //      at a row stride are store-issue bound (~250 cycles per wave store measured); staged, the tile
  //      leaves as float4 row segments with bias / ReLU / residual applied on whole rows.
  constexpr int OS = NT + 4;
  const int hi = lane >> 5;
  // this thread's column quad is the same for every row it stores (e % (NT/4) == tid % (NT/4)): its bias is loaded once,
  // up front -- inside the store loop each iteration waited a full global-load latency for it (16 x ~1.3k cycles measured)
  constexpr int ROWS_IT = BM * (NT / 4) / 256;          // rows per thread: 16 (NT = 128) / 8 (NT = 64)
  const int c4 = (tid % (NT / 4)) * 4, gcol = n0 + c4, row0 = tid / (NT / 4);
  const bool vec_ok = (p.ldo & 3) == 0 && (!p.res || (p.ldr & 3) == 0);
  float4 bsv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias && gcol < p.Npad) bsv = *reinterpret_cast<const float4*>(p.bias + gcol);
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int col = n * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) smem[(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi) * OS + col] = acc[n][r];
  }
  __syncthreads();
  if (gcol >= p.N) return;
  const bool fast = vec_ok && gcol + 3 < p.N;
  if (fast && r0 + BM <= p.M) {
    // Common case (whole tile inside M, aligned rows): straight-line stores -- no per-row bounds tests, row pointers
    // advance by a constant, ReLU / residual chosen once.  The epilogue runs beside the neighbour workgroup's MFMA loop,
    // where every instruction costs tens of cycles.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int RSTEP = 256 / (NT / 4);
    const f32x4 bias4 = {bsv.x, bsv.y, bsv.z, bsv.w};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float* op = p.out + (size_t)(r0 + row0) * p.ldo + gcol;
    const float* rp = p.res ? p.res + (size_t)(r0 + row0) * p.ldr + gcol : nullptr;
    const size_t ostep = (size_t)RSTEP * p.ldo, rstep = (size_t)RSTEP * p.ldr;
    const float* sp = smem + row0 * OS + c4;
    auto rows = [&](auto relu_c, auto res_c) __attribute__((always_inline)) {
      constexpr bool RELU = decltype(relu_c)::value, RES = decltype(res_c)::value;
#pragma unroll
      for (int b0 = 0; b0 < ROWS_IT; b0 += 8) {
        f32x4 v[8], rv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (RES) rv[i] = *reinterpret_cast<const f32x4*>(rp + (b0 + i) * rstep);
          v[i] = *reinterpret_cast<const f32x4*>(sp + (b0 + i) * RSTEP * OS);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          f32x4 o = v[i] + bias4;
          if constexpr (RELU) o = __builtin_elementwise_max(o, zero4);
          if constexpr (RES) o = rv[i] + o;
          *reinterpret_cast<f32x4*>(op + (b0 + i) * ostep) = o;
        }
      }
    };
    using T = std::true_type; using F = std::false_type;
    if (p.relu) { if (p.res) rows(T{}, T{}); else rows(T{}, F{}); }
    else { if (p.res) rows(F{}, T{}); else rows(F{}, F{}); }
    return;
  }
  // rows in batches of 8: all residual loads and LDS reads of a batch are issued before the first store
#pragma unroll
  for (int b0 = 0; b0 < ROWS_IT; b0 += 8) {
    float4 v[8], rv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = row0 + (b0 + i) * (256 / (NT / 4)), grow = r0 + row;
      rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (fast && p.res && grow < p.M) rv[i] = *reinterpret_cast<const float4*>(p.res + (size_t)grow * p.ldr + gcol);
      v[i] = *reinterpret_cast<const float4*>(smem + row * OS + c4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = row0 + (b0 + i) * (256 / (NT / 4)), grow = r0 + row;
      if (grow >= p.M) continue;
      float4 o = v[i];
      o.x += bsv.x; o.y += bsv.y; o.z += bsv.z; o.w += bsv.w;
      if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (fast) {
        o.x = rv[i].x + o.x; o.y = rv[i].y + o.y; o.z = rv[i].z + o.z; o.w = rv[i].w + o.w;
        *reinterpret_cast<float4*>(p.out + (size_t)grow * p.ldo + gcol) = o;
      } else {                                // ragged N (e.g. 65) or unaligned leading dimension
        const float vv[4] = {o.x, o.y, o.z, o.w};
        for (int j = 0; j < 4 && gcol + j < p.N; ++j)
          p.out[(size_t)grow * p.ldo + gcol + j] = (p.res ? p.res[(size_t)grow * p.ldr + gcol + j] : 0.f) + vv[j];
      }
    }
  }
}

// "NT" GEMM: both operands row-major [row][k]; tile 128 (i) x 64 (j).
__global__ __launch_bounds__(256) void score_mfma(ScoreArgs p) {
  constexpr int NT = 64, NB = 2;
  __shared__ __attribute__((aligned(16))) float smem[(BM + NT) * SA];
  float* a_tile = smem;
  float* b_tile = smem + BM * SA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * BM, j0 = blockIdx.y * NT, b = blockIdx.z;
  const float* m0 = p.m0 + (size_t)b * p.N0p * p.d;
  const float* m1 = p.m1 + (size_t)b * p.N1p * p.d;

  f32x16 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  for (int c0 = 0; c0 < p.d; c0 += CK) {
    __syncthreads();
    for (int e = tid; e < (BM + NT) * (CK / 4); e += 256) {
      int row = e / (CK / 4), v4 = e % (CK / 4);
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < BM) {
        if (i0 + row < p.N0p) val = *reinterpret_cast<const float4*>(m0 + (size_t)(i0 + row) * p.d + c0 + 4 * v4);
      } else {
        if (j0 + row - BM < p.N1p) val = *reinterpret_cast<const float4*>(m1 + (size_t)(j0 + row - BM) * p.d + c0 + 4 * v4);
      }
      float* d = smem + row * SA + 4 * v4;
      d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
    __syncthreads();
    const float* ap = a_tile + (32 * wave + (lane & 31)) * SA + (lane >> 5);
    const float* bp = b_tile + (lane & 31) * SA + (lane >> 5);
    float af[2], bf[2][NB];
    af[0] = ap[0];
#pragma unroll
    for (int n = 0; n < NB; ++n) bf[0][n] = bp[n * 32 * SA];
#pragma unroll
    for (int kk = 0; kk < CK / 2; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < CK / 2) {
        af[nxt] = ap[2 * (kk + 1)];
#pragma unroll
        for (int n = 0; n < NB; ++n) bf[nxt][n] = bp[n * 32 * SA + 2 * (kk + 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NB; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur][n], acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int hi = lane >> 5;
  float* out = p.out + (size_t)b * p.N0p * p.N1p;
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int j = j0 + n * 32 + (lane & 31);
    if (j >= p.N1p) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (i < p.N0p) out[(size_t)i * p.N1p + j] = acc[n][r] * p.scale;
    }
  }
}
}  // namespace

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.K0 % CK || a.K1 % CK || a.Npad % 64 || a.M <= 0) return hipErrorInvalidValue;
  const unsigned gm = (unsigned)((a.M + BM - 1) / BM);
  last_form = "gemm_tiled:f32";
  static unsigned long long attr[2] = {0, 0};
  if (a.Npad % 128 == 0) {
    raise_lds_limit(reinterpret_cast<const void*>(gemm_mfma<128>), 68 * 1024, attr[0]);
    hipLaunchKernelGGL(gemm_mfma<128>, dim3(gm, a.Npad / 128), dim3(256),
                       std::max<size_t>(2 * (BM * SA + CK * 128), BM * (128 + 4)) * sizeof(float), s, a);
  } else {
    hipLaunchKernelGGL(gemm_mfma<64>, dim3(gm, a.Npad / 64), dim3(256),
                       std::max<size_t>(2 * (BM * SA + CK * 64), BM * (64 + 4)) * sizeof(float), s, a);
  }
  return hipGetLastError();
}

hipError_t launch_score_gemm(const ScoreArgs& a, hipStream_t s) {
  if (a.d % CK) return hipErrorInvalidValue;
  dim3 grid((unsigned)((a.N0p + BM - 1) / BM), (unsigned)((a.N1p + 63) / 64), (unsigned)a.B);
  hipLaunchKernelGGL(score_mfma, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace imx
