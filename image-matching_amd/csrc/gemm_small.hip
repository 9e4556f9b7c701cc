// gemm_small.hip — the 1x1-conv / linear product for SMALL row counts (a single pair: M = 2 x 1024 keypoint rows), where latency,
// not throughput, is what counts (BASELINE configs[2]; superglue_test.py:49-60,92-119,214-216).
//
// A weights-stationary form (a workgroup's W columns in registers for the whole K, amortised over many 64-row tiles; round 2's
// gemm_ws.hip, removed in round 3) has 32-96 workgroups of ONE tile each at M = 2048, every one of them first pulling 64-128 KB of
// W through 128 dword loads per lane: 11-16 us per launch, 54 launches per pair.  Here the work is cut the other way: a
// workgroup owns a 32 x 32 output tile (four waves of 16 x 16 on v_mfma_f32_16x16x4_f32), stages its whole A (32 x K) and W
// (K x 32) panels in LDS with every load in flight at once (one barrier), and multiplies from LDS: 256-768 workgroups per
// launch, K/4 MFMAs per wave.  Reductions longer than 128 are accumulated in two levels (128-k blocks folded into a running
// sum), like the throughput forms.
#include "imx_kernels.h"

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TM = 32, TN = 32;
constexpr int SW = 36;            // W panel row stride (floats): k rows 4 apart land 16 banks apart (conflict-free ds_read_b32)

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void gemm_small(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = p.K0 + p.K1;
  const int SA = K + 8;            // A panel row stride: (K/4 + 2) 16-byte slots = 2 mod 16 for K % 64 == 0 (conflict-free ds_read_b128)
  float* As = sm;                  // [32][SA]
  float* Ws = sm + TM * SA;        // [K][SW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * TM, n0 = blockIdx.y * TN;

  // ---- stage the panels: every load is issued before the first store (one global round trip for the whole tile)
  //      A: 32 rows x K floats = 8 K float4 -> K/32 per thread; rows past M read row M-1 (never stored)
  //      W: K rows x 32 floats = 8 K float4 -> K/32 per thread
  const int nv = K / 32;           // float4 per thread and panel (K % 32 == 0): 2 .. 16
  f32x4 av[16], wv[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    if (it < nv) {
      const int e = tid + it * 256;
      const int row = e / (K / 4), k4 = (e % (K / 4)) * 4;
      const int grow = min(r0 + row, p.M - 1);
      const float* src = k4 < p.K0 ? p.a0 + (size_t)grow * p.lda0 + k4 : p.a1 + (size_t)grow * p.lda1 + (k4 - p.K0);
      av[it] = *reinterpret_cast<const f32x4*>(src);
      const int wk = e / (TN / 4), wc = (e % (TN / 4)) * 4;
      wv[it] = *reinterpret_cast<const f32x4*>(p.w + (size_t)wk * p.Npad + n0 + wc);
    }
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    if (it < nv) {
      const int e = tid + it * 256;
      const int row = e / (K / 4), k4 = (e % (K / 4)) * 4;
      *reinterpret_cast<f32x4*>(As + row * SA + k4) = av[it];
      const int wk = e / (TN / 4), wc = (e % (TN / 4)) * 4;
      *reinterpret_cast<f32x4*>(Ws + wk * SW + wc) = wv[it];
    }
  }
  __syncthreads();

  // ---- wave (wr, wc) = 16 x 16 outputs; MFMA 16x16x4: A lane (i = lane&15, kq = lane>>4) holds A[i][4 kq' ...], and because the
  //      sum over k is order free the j-th MFMA of a 16-k group takes k = 16 t + 4 kq + j from BOTH operands: the lane's four A
  //      values are one ds_read_b128, its four W values four ds_read_b32 of consecutive rows
  const int wr = wave >> 1, wc = wave & 1, n = lane & 15, kq = lane >> 4;
  const float* ap = As + (16 * wr + n) * SA + 4 * kq;
  const float* bp = Ws + (4 * kq) * SW + 16 * wc + n;
  f32x4 tot = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < K / 16; ++t) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 16 * t);
    const float* b = bp + 16 * t * SW;
    const float b0 = b[0], b1 = b[SW], b2 = b[2 * SW], b3 = b[3 * SW];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], b3, acc, 0, 0, 0);
    if ((t & 7) == 7) {             // end of a 128-k block (two-level accumulation)
      tot += acc;
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  tot += acc;

  // ---- epilogue straight from registers: D lane (col = lane&15, g = lane>>4) holds rows 4 g .. 4 g + 3 of its column; sixteen
  //      lanes store sixteen consecutive floats of a row
  const int col = n0 + 16 * wc + n;
  if (col >= p.N) return;
  const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + 16 * wr + 4 * kq + r;
    if (row >= p.M) continue;
    float v = tot[r] + bias;
    if (RELU) v = fmaxf(v, 0.f);
    if (RES) v = p.res[(size_t)row * p.ldr + col] + v;
    p.out[(size_t)row * p.ldo + col] = v;
  }
}
}  // namespace

// K % 64 == 0 (conflict-free A stride), K <= 512, K0 % 4 == 0, float4-aligned A rows, Npad % 32 == 0
bool gemm_small_supported(const GemmArgs& a) {
  const int K = a.K0 + a.K1;
  if (K % 64 || K > 512 || a.K0 % 4 || a.Npad % TN || a.M <= 0) return false;
  if ((a.lda0 & 3) || (a.a1 && (a.lda1 & 3)) || (a.K1 && !a.a1)) return false;
  return true;
}

hipError_t launch_gemm_small(const GemmArgs& a, hipStream_t s) {
  if (!gemm_small_supported(a)) return hipErrorInvalidValue;
  const int K = a.K0 + a.K1;
  const size_t lds = (size_t)(TM * (K + 8) + K * SW) * sizeof(float);     // K = 256: 33.8 + 36.9 KB
  const dim3 grid((unsigned)((a.M + TM - 1) / TM), (unsigned)(a.Npad / TN));
  static unsigned long long attr[4] = {0, 0, 0, 0};
  last_form = "gemm_small:f32";
  auto go = [&](auto kern, int id) {
    raise_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024, attr[id]);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  };
  if (a.res) { if (a.relu) go(gemm_small<true, true>, 3); else go(gemm_small<true, false>, 2); }
  else { if (a.relu) go(gemm_small<false, true>, 1); else go(gemm_small<false, false>, 0); }
  return hipGetLastError();
}

}  // namespace imx
