// mlp_x3.hip — the AttentionalPropagation MLP of one GNN layer in ONE kernel (superglue_test.py:112-119):
//     x <- x + W2 . relu(W1 . cat[x, message] + b1) + b2          (attn.merge is folded into W1 at load, imx_api.cpp)
// with every fp32 product carried as six bf16 term products on the bf16 matrix pipe (gemm_x3.hip explains the split).
//
// As two gemm_x3 launches the 2d-wide hidden tensor makes a round trip through HBM (134 MB written and read per layer at 64
// pairs, of 536 MB for the two products) and both kernels are memory / latency bound.  Here a persistent 512-thread workgroup owns
// 64 rows at a time and the hidden tile never leaves the CU:
//   phase 1  hidden^T = W1^T . A^T  (K = 2d in chunks of 32; A rows arrive as fp32, are split in registers and stored as three
//            bf16 planes in LDS, double buffered, one barrier per chunk -- the gemm_x3 stream).  Wave w owns hidden columns
//            32 w .. 32 w + 31 of all 64 rows.  The product is computed TRANSPOSED (weight fragment as the A operand), so a lane
//            holds one row and four consecutive hidden columns per register group: bias, ReLU, split, and three 8-byte stores per
//            group put the hidden tile into LDS as the bf16 planes phase 2 reads (528-byte rows: 33 sixteen-byte slots, odd).
//   phase 2  out = hidden . W2 + b2 + x  (K = 2d from LDS, 16 steps): wave (rg, cg) owns rows 32 rg.. x columns 32 cg..; while it
//            runs, the A rows of the workgroup's NEXT tile are already being fetched and staged.
// Weight fragments come straight from global memory in the host-made fragment order of gemm_x3 (L2 / L1 resident), requested a
// step ahead.  d = 128 only (hidden 256: 99 KB of LDS); other widths take the two-launch path.
#include "imx_kernels.h"
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int D = 128, K1 = 2 * D, NH = 2 * D, K2 = NH, NO = D;   // phase 1: K1 -> NH hidden; phase 2: K2 -> NO outputs
constexpr int BM = 64, KC = 32, RS = 40;                         // rows per tile, k per chunk, A-plane row stride (bf16)
constexpr int HS = NH + 8;                                       // hidden-plane row stride (bf16): 528 bytes
constexpr int APL = BM * RS, HPL = BM * HS;                      // elements per plane
constexpr int NCH = K1 / KC;                                     // 8 chunks
constexpr int NST1 = K1 / 16, NST2 = K2 / 16;

__global__ __launch_bounds__(512, 1) void gnn_mlp_x3(MlpArgs p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  __bf16* As0 = lds;                      // [3][APL]
  __bf16* As1 = lds + 3 * APL;
  __bf16* Hs = lds + 6 * APL;             // [3][HPL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, kb = lane >> 5;
  const int G = (int)gridDim.x;
  const int first = (int)blockIdx.x;
  if (first >= ntiles) return;
  const __bf16* w1 = static_cast<const __bf16*>(p.w1x3);
  const __bf16* w2 = static_cast<const __bf16*>(p.w2x3);

  // ---- A loader: one float4 per thread and chunk: row tid >> 3, k 4 (tid & 7).  cat[x, message]: chunks 0..3 from x, 4..7 from msg
  f32x4 areg;
  auto gload = [&](int r0, int c) __attribute__((always_inline)) {
    const int k0 = c * KC;
    const float* src = k0 < D ? p.x + k0 : p.msg + (k0 - D);
    const int ld = k0 < D ? p.ldx : p.ldm;
    const int grow = min(r0 + (tid >> 3), p.M - 1);           // rows past M re-read the last row (never stored)
    areg = *reinterpret_cast<const f32x4*>(src + (size_t)grow * ld + (tid & 7) * 4);
  };
  auto lstore = [&](__bf16* Ad) __attribute__((always_inline)) {
    bf16x4 h, m, l;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float x = areg[t];
      h[t] = (__bf16)x;
      const float r1 = x - (float)h[t];
      m[t] = (__bf16)r1;
      l[t] = (__bf16)(r1 - (float)m[t]);
    }
    const int o = (tid >> 3) * RS + (tid & 7) * 4;
    *reinterpret_cast<bf16x4*>(Ad + o) = h;
    *reinterpret_cast<bf16x4*>(Ad + APL + o) = m;
    *reinterpret_cast<bf16x4*>(Ad + 2 * APL + o) = l;
  };
  // weight fragments: column block nb, step st, plane pl -> 512 elements at ((nb * nst + st) * 3 + pl) * 512
  auto wload = [&](bf16x8 (&wf)[3], const __bf16* w, int nb, int nst, int st) __attribute__((always_inline)) {
    const __bf16* wb = w + ((size_t)nb * nst + st) * (3 * 512) + lane * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(wb + pl * 512);
  };
  constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};

  // phase-2 role
  const int rg = wave >> 2, cg = wave & 3;

  int r0 = first * BM;
  gload(r0, 0);
  lstore(As0);
  gload(r0, 1);
  __syncthreads();
  for (int t = first; t < ntiles; t += G) {
    r0 = t * BM;
    const int rn = t + G < ntiles ? (t + G) * BM : r0;        // the workgroup's next tile (or this one again: harmless re-fetch)
    // ================================================================ phase 1: hidden^T block of wave w: 32 hidden columns x 64 rows
    f32x16 acc1[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[rb][r] = 0.f;
    bf16x8 wfa[3], wfb[3];
    wload(wfa, w1, wave, NST1, 0);
    auto chunk = [&](int c, const __bf16* Ar, __bf16* Ad) __attribute__((always_inline)) {
      wload(wfb, w1, wave, NST1, 2 * c + 1);
      if (c + 1 < NCH) {                                        // block-uniform
        lstore(Ad);                                             // chunk c + 1 (requested one chunk ago)
        if (c + 2 < NCH) gload(r0, c + 2);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 (&wf)[3] = s ? wfb : wfa;
        bf16x8 af[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) af[rb][pl] = *reinterpret_cast<const bf16x8*>(Ar + pl * APL + (rb * 32 + i) * RS + s * 16 + kb * 8);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) acc1[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PA[q]], af[rb][PB[q]], acc1[rb], 0, 0, 0);
        if (s == 0 && c + 1 < NCH) wload(wfa, w1, wave, NST1, 2 * c + 2);
      }
      __syncthreads();
    };
#pragma unroll 1
    for (int c = 0; c < NCH; c += 2) {
      chunk(c, As0, As1);
      chunk(c + 1, As1, As0);
    }
    // ---- hidden tile -> LDS planes: lane (row = i (+32 rb), kb) holds hidden columns 32 wave + 8 g + 4 kb + (0..3) in acc1[rb][4 g ..]
    {
      const int c0 = 32 * wave + 4 * kb;
      f32x4 b1[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b1[g] = *reinterpret_cast<const f32x4*>(p.b1 + c0 + 8 * g);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 h, m, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = fmaxf(acc1[rb][4 * g + e] + b1[g][e], 0.f);
            h[e] = (__bf16)x;
            const float r1 = x - (float)h[e];
            m[e] = (__bf16)r1;
            l[e] = (__bf16)(r1 - (float)m[e]);
          }
          const int o = (rb * 32 + i) * HS + c0 + 8 * g;
          *reinterpret_cast<bf16x4*>(Hs + o) = h;
          *reinterpret_cast<bf16x4*>(Hs + HPL + o) = m;
          *reinterpret_cast<bf16x4*>(Hs + 2 * HPL + o) = l;
        }
    }
    // the next tile's first chunks: fetched and staged while phase 2 runs (As0 / As1 are idle after the last phase-1 barrier)
    gload(rn, 0);
    __syncthreads();                                            // hidden planes complete
    // ================================================================ phase 2: out block of wave (rg, cg): 32 rows x 32 columns
    f32x16 acc2, acc2b;                      // the two steps of an iteration accumulate separately (no back-to-back dependent MFMAs)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc2[r] = 0.f; acc2b[r] = 0.f; }
    wload(wfa, w2, cg, NST2, 0);
#pragma unroll 1
    for (int st = 0; st < NST2; st += 2) {
      wload(wfb, w2, cg, NST2, st + 1);
      if (st == 2) { lstore(As0); gload(rn, 1); }               // next tile, chunk 0 staged; chunk 1 requested
      {
        bf16x8 hf0[3], hf1[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          hf0[pl] = *reinterpret_cast<const bf16x8*>(Hs + pl * HPL + (rg * 32 + i) * HS + st * 16 + kb * 8);
          hf1[pl] = *reinterpret_cast<const bf16x8*>(Hs + pl * HPL + (rg * 32 + i) * HS + (st + 1) * 16 + kb * 8);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf0[PA[q]], wfa[PB[q]], acc2, 0, 0, 0);
          acc2b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf1[PA[q]], wfb[PB[q]], acc2b, 0, 0, 0);
        }
        if (st + 2 < NST2) wload(wfa, w2, cg, NST2, st + 2);
      }
    }
    // ---- epilogue: lane (col = i, kb) holds rows (r & 3) + 8 (r >> 2) + 4 kb of column 32 cg + i; x may alias out: the sixteen
    //      residual loads are issued together, before the stores
    {
      const int col = 32 * cg + i;
      const int rbase = r0 + 32 * rg + 4 * kb;
      const float b2 = p.b2[col];
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        rv[r] = row < p.M ? p.x[(size_t)row * p.ldx + col] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < p.M) p.out[(size_t)row * p.ldo + col] = rv[r] + ((acc2[r] + acc2b[r]) + b2);
      }
    }
    __syncthreads();                                            // every wave is done with the hidden planes (and As0 holds chunk 0)
  }
}
}  // namespace

bool gnn_mlp_x3_supported(const MlpArgs& a) {
  return a.d == D && a.M > 0 && a.w1x3 && a.w2x3 && !(a.ldx & 3) && !(a.ldm & 3);
}

hipError_t launch_gnn_mlp_x3(const MlpArgs& a, hipStream_t s) {
  if (!gnn_mlp_x3_supported(a)) return hipErrorInvalidValue;
  const int ntiles = (a.M + BM - 1) / BM;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const size_t lds = (size_t)(6 * APL + 3 * HPL) * sizeof(__bf16);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gnn_mlp_x3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));
  hipLaunchKernelGGL(gnn_mlp_x3, grid, dim3(512), lds, s, a, ntiles);
  return hipGetLastError();
}

}  // namespace imx
