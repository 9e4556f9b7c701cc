// gemm_ws.hip — weights-stationary, persistent fp32 GEMM for the tall-and-thin products of the path:
//   out[r][n] = act( sum_k [a0|a1][r][k] * W[k][n] + bias[n] ) (+ res[r][n]),   rows = every keypoint of every pair of the
// batch (10^5), K = 128..512, N = 128..768: the Conv1d(k=1) projections / MLPs of superglue_test.py:49-60,92-119,214-216 and
// convDb of superpoint_test.py:83 (BatchNorm folded, torch.cat([x, message]) as a K split).  Same contract as launch_gemm.
//
// Why a second form: with K this short, a tiled GEMM re-stages a K x 128 weight panel through LDS for every 128 rows and
// spends a third of its life in prologue/epilogue (gemm.hip: MFMA pipe 0.52-0.55 busy).  Here the WEIGHTS NEVER MOVE:
// a workgroup (4 waves) owns 128 output columns, a wave 32 of them; its W fragments for the whole K (K/4 VGPRs per 16-column
// block: 64 registers at K = 128, 128 at K = 256) are loaded once and stay in registers while the workgroup walks 64-row
// tiles of the activations.  Only A streams: 64 rows x 64 k stages through LDS (double buffered, one barrier per stage),
// fetched two stages ahead with buffer loads whose per-thread offsets are tile independent (tile and k offsets ride in the
// SGPR offset; rows past M are out of range and come back as zeros).  K is permuted so that a lane's A operands of four
// consecutive MFMAs are four consecutive k of its row: one ds_read_b128 feeds 4 x (columns blocks) MFMAs -- 16 LDS reads per
// 128 v_mfma_f32_16x16x4_f32 per wave and stage.  Bias / ReLU / residual / stores go through an LDS staging tile as whole
// float4 row segments.  LDS 2 x 17 KB + 33 KB, two workgroups per CU.
#include "imx_kernels.h"
#include <algorithm>
#include <cstdlib>

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int RT = 64;            // rows per tile
constexpr int KC = 64;            // k per stage
// A stage row stride: 72 floats = 18 sixteen-byte slots.  The LDS services a ds_read_b128 in four 16-lane groups that mix eight lanes
// of k-quad kq with the complementary eight of kq+1 ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS); a lane reads slot
// row*S + 4t + kq, and with S = 18 the sixteen slots of every group are distinct mod 16 (S = 17, round 1, left one 2-way
// conflict per group: SQ_LDS_BANK_CONFLICT was 34 % of the kernel's LDS cycles)
constexpr int AS = KC + 8;

// CBW = 16-column blocks per wave: 2 (128 columns per workgroup) for K <= 256; 1 (64 columns) for K = 512, where a wave's
// weights already fill 128 registers.
template <int K, bool RES, bool RELU, int CBW>
__global__ __launch_bounds__(256, 2) void gemm_ws(GemmArgs p, int ntile, int ncg) {
  constexpr int NTW = 64 * CBW;   // columns per workgroup
  constexpr int OSN = NTW + 4;
  constexpr int EIT = RT * (NTW / 4) / 256;      // float4 row segments per thread in the epilogue: 8 / 4
  constexpr int NC = K / KC;      // stages per tile
  constexpr int NQ = K / 16;      // operand quads over K
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;               // [2][RT * AS]
  float* Ot = smem + 2 * RT * AS; // [RT][OSN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  // XCD-aware placement: workgroups are dispatched round-robin over the 8 XCDs.  The ncg column groups of one row partition
  // stream the same A rows, so they sit on ONE XCD (same L2), next to each other in dispatch order: A is fetched from HBM once
  // instead of ncg times (nparts is a multiple of 8 whenever it is at least 8)
  const int nparts = (int)gridDim.x / ncg;
  int cg, part;
  if ((nparts & 7) == 0) {
    const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    cg = q % ncg;
    part = (q / ncg) * 8 + xcd;
  } else {
    cg = (int)blockIdx.x % ncg;
    part = (int)blockIdx.x / ncg;
  }
  const int n0 = cg * NTW;
  if (part >= ntile) return;

  // ---- this wave's weights, resident for the whole kernel: bq[cbk][quad t][j] = W[16 t + 4 kq + j][n0 + 32 wave + 16 cbk + n]
  f32x4 bq[CBW][NQ];
#pragma unroll
  for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bq[cbk][t][j] = p.w[(size_t)(16 * t + 4 * kq + j) * p.Npad + n0 + 16 * CBW * wave + 16 * cbk + n];

  // ---- loader: a stage is 64 rows x 16 float4; thread -> rows tid/16 + 16 it (it = 0..3), float4 tid % 16
  int vo0[4], vo1[4], ld_dst[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (tid >> 4) + 16 * it, v4 = tid & 15;
    vo0[it] = (row * p.lda0 + 4 * v4) * 4;
    vo1[it] = (row * (p.a1 ? p.lda1 : p.lda0) + 4 * v4) * 4;
    ld_dst[it] = row * AS + 4 * v4;
  }
  f32x4 lr[4];
  int ltile = part, lc = 0;        // loader cursor (tile, stage)
  // (the tile offset rides in the SGPR operand, which is NOT bounds-checked: rows past M in the last tile are READ from whatever
  //  follows the M rows inside the workspace arena -- harmless, their results are never stored; the stores are range-checked)
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a0, 0, p.M * p.lda0 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a1 ? p.a1 : p.a0), 0, p.M * (p.a1 ? p.lda1 : p.lda0) * 4, 0x00020000);
  auto issue_load = [&]() {
    const int kb = lc * KC;
    const bool second = kb >= p.K0;
    const int so = __builtin_amdgcn_readfirstlane((ltile * RT * (second ? (p.a1 ? p.lda1 : p.lda0) : p.lda0) + (second ? kb - p.K0 : kb)) * 4);
    const __amdgpu_buffer_rsrc_t rs = second ? rs1 : rs0;      // branch-free: every issue is exactly four loads (exact vmcnt bookkeeping)
#pragma unroll
    for (int it = 0; it < 4; ++it) lr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, second ? vo1[it] : vo0[it], so, 0));
    if (++lc == NC) { lc = 0; ltile += nparts; }       // past the last tile: garbage or zeros, never used
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(As + buf * (RT * AS) + ld_dst[it]) = lr[it];
  };

  // epilogue: thread -> column quad tid % 32 (fixed), rows tid/32 + 8 i.  Residual loads and stores are buffer operations
  // over M rows: rows past M read zeros / are dropped by the hardware, so the epilogue is branch free too
  const int c4 = (tid % (NTW / 4)) * 4, erow = tid / (NTW / 4);
  constexpr int ERS = 256 / (NTW / 4);           // rows per epilogue pass: 8 / 16
  // (descriptors are built per tile in the epilogue: the SGPR offset operand of a buffer access is NOT bounds-checked, so the
  //  tile's row offset must be part of the descriptor's base for rows past M to be out of range)
  // per-thread byte offsets inside a pass of ERS rows (the pass's first row is the base of its buffer descriptor)
  const int eo0 = (erow * p.ldo + n0 + c4) * 4, er0 = (erow * (RES ? p.ldr : p.ldo) + n0 + c4) * 4;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + n0 + c4);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- pipeline fill
  issue_load();
  store_stage(0);
  issue_load();
  __builtin_amdgcn_s_waitcnt(0x0F70);       // drain: the loop's waitcnt bookkeeping then sees only its own in-order loads
  int par = 0;
  const int aoff = n * AS + 4 * kq;          // A operand of row block rb, quad t: As[(16 rb + n) * AS + 16 t + 4 kq]

  for (int tile = part; tile < ntile; tile += nparts) {
    // Two-level accumulation for the long reductions (K = 512: C5's MLPs): v_mfma_f32 accumulates like a sequential fma chain
    // (tools/ubench/mfma_round.hip), whose rounding error grows with the chain; blocks of 128 k start from zero and are folded
    // into `tot` with packed adds (16 per 128 MFMAs).  With it C5's scores_in is 1.2x (without: 1.5x) further from a float64
    // evaluation than the reference's fp32 result in the CPU emulation of this order (tools/accuracy_emul.py).  K <= 256 keeps
    // one level: the K = 256 form has no registers to spare (248) and C3 measures 1.06x as it is.
    constexpr bool TWO = K >= 512;
    f32x4 acc[4][CBW], tot[TWO ? 4 : 1][CBW];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cbk = 0; cbk < CBW; ++cbk) {
        acc[rb][cbk] = zero4;
        if constexpr (TWO) tot[rb][cbk] = zero4;
      }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      __syncthreads();             // stage `par` complete; the other buffer's readers are done
      const float* ab = As + par * (RT * AS) + aoff;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int t = 0; t < KC / 16; ++t) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(ab + rb * 16 * AS + 16 * t);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int cbk = 0; cbk < CBW; ++cbk)
              acc[rb][cbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bq[cbk][c * (KC / 16) + t][j], acc[rb][cbk], 0, 0, 0);
          if (rb == 0 && t == 1) {   // the next stage: registers -> the idle buffer, then fetch the stage after it
            store_stage(par ^ 1);
            issue_load();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      par ^= 1;
      if constexpr (TWO) {
        if ((c & 1) == 1) {          // two stages = 128 k
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cbk = 0; cbk < CBW; ++cbk) {
              tot[rb][cbk] += acc[rb][cbk];
              if (c + 1 < NC) acc[rb][cbk] = zero4; else acc[rb][cbk] = tot[rb][cbk];
            }
        }
      }
    }
    // ---- tile done: stage the 64 x 128 result, then bias / ReLU / residual on whole float4 row segments.
    //      acc[rb][cbk][r]: row 16 rb + 4 kq + r, column 32 wave + 16 cbk + n.
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ot[(16 * rb + 4 * kq + r) * OSN + 16 * CBW * wave + 16 * cbk + n] = acc[rb][cbk][r];
    __syncthreads();
    {
      const int ldr_ = RES ? p.ldr : p.ldo;
      const int trow = __builtin_amdgcn_readfirstlane(tile * RT);
      // one descriptor per epilogue pass (scalar arithmetic only): base = first row of the pass, range = what is left of the M
      // rows from there, so a thread's row inside the pass (VGPR offset, checked) past M is dropped / reads zero
      auto pass_rsrc = [&](const float* ptr, int ld, int i) {
        const int row0 = trow + ERS * i;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(ptr + (size_t)row0 * ld), 0, row0 < p.M ? (p.M - row0) * ld * 4 : 0, 0x00020000);
      };
      f32x4 v[EIT], rv[EIT];
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        if constexpr (RES) rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pass_rsrc(p.res, ldr_, i), er0, 0, 0));
        v[i] = *reinterpret_cast<const f32x4*>(Ot + (erow + ERS * i) * OSN + c4);
      }
#pragma unroll
      for (int i = 0; i < EIT; ++i) {
        f32x4 o = v[i] + bias4;
        if constexpr (RELU) o = __builtin_elementwise_max(o, zero4);      // compile-time: a run-time flag costs a v_cndmask per value
        if constexpr (RES) o = rv[i] + o;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), pass_rsrc(p.out, p.ldo, i), eo0, 0, 0);
      }
    }
  }
}

template <int K, bool RES, bool RELU, int CBW>
hipError_t launch_k(const GemmArgs& a, hipStream_t s) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  constexpr int NTW = 64 * CBW, OSN = NTW + 4;
  const int ntile = (a.M + RT - 1) / RT, ncg = a.Npad / NTW;
  int nparts = 2 * ncu / ncg;
  if (nparts > ntile) nparts = ntile;
  if (nparts >= 8) nparts &= ~7;             // multiple of 8: XCD-aware placement (see the kernel)
  if (nparts < 1) nparts = 1;
  const size_t lds = (size_t)(2 * RT * AS + RT * OSN) * sizeof(float);
  auto k = gemm_ws<K, RES, RELU, CBW>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)(nparts * ncg)), dim3(256), lds, s, a, ntile, ncg);
  return hipGetLastError();
}
}  // namespace

bool gemm_ws_supported(const GemmArgs& a) {
  const int K = a.K0 + a.K1;
  if (K != 128 && K != 256 && K != 512) return false;
  if (a.K0 % KC || a.K1 % KC || a.Npad % (K == 512 ? 64 : 128) || a.N != a.Npad || !a.bias) return false;
  if ((a.ldo & 3) || (a.res && (a.ldr & 3)) || (a.lda0 & 3) || (a.a1 && (a.lda1 & 3))) return false;
  const int ldmax = std::max(std::max(a.lda0, a.a1 ? a.lda1 : 0), std::max(a.ldo, a.res ? a.ldr : 0));
  if ((size_t)a.M * (size_t)ldmax * 4 >= 0x7fffffffull) return false;    // 31-bit buffer offsets
  return a.M > 0;
}

hipError_t launch_gemm_ws(const GemmArgs& a, hipStream_t s) {
  if (!gemm_ws_supported(a)) return hipErrorInvalidValue;
  const int K = a.K0 + a.K1, sel = (a.res ? 2 : 0) + (a.relu ? 1 : 0);
#define IMX_WS(K_, CBW_)                                                  \
  switch (sel) {                                                          \
    case 0: return launch_k<K_, false, false, CBW_>(a, s);                \
    case 1: return launch_k<K_, false, true, CBW_>(a, s);                 \
    case 2: return launch_k<K_, true, false, CBW_>(a, s);                 \
    default: return launch_k<K_, true, true, CBW_>(a, s);                 \
  }
  if (K == 128) { IMX_WS(128, 2) }
  if (K == 256) { IMX_WS(256, 2) }
  IMX_WS(512, 1)
#undef IMX_WS
}

}  // namespace imx
