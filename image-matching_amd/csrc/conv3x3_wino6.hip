// 3x3 convolution (pad 1) + folded BatchNorm + ReLU (+ fused 2x2 max-pool, + fused conv1a) as Winograd F(2x2,3x3)
// on the fp32 matrix cores -- persistent, wave-specialised form for gfx950 (MI355X).
//
// Replaces double_conv / inconv / down (superpoint/models/unet_parts.py:10-48) and convPa/convDa
// (superpoint_test.py:76-84); same arithmetic as conv3x3_wino.hip (U = G g G^T precomputed at weight load,
// V = B^T d B per tile, Y = A^T M A in-lane), different execution shape:
//
//   * one workgroup of 8 waves per CU, persistent over a list of (tile, 64-channel block) work items;
//   * waves 0-3 are CONSUMERS: nothing but v_mfma_f32_16x16x4_f32 (64 per 8-channel chunk), their A operands from
//     LDS (V, ds_read2st64) and B operands (U) from L2 into a 32-register panel refilled in place (8 buffer_load_dwordx4
//     per chunk), and the in-lane output transform;
//   * waves 4-7 are PRODUCERS: they fetch the next raw 10x18x8 input patches (or compute them: FIRST mode fuses
//     conv1a), run the input transform into the other V buffer, and write finished output tiles from an LDS
//     staging tile to HBM as whole 256-byte channel rows;
//   * ONE barrier per chunk.  Prologue, input transform, epilogue stores and index arithmetic all run in the shadow
//     of the consumers' MFMA stream instead of in series with it (conv3x3_wino.hip measured 2.05k MFMA cycles inside
//     a 5.1-6.0k-cycle chunk period plus a 13-17 % prologue/epilogue share).
//
// Tile = 4x8 Winograd tiles (8x16 output pixels) x 64 output channels; consumer wave cb owns channels cb*16..+16 for
// all 32 wtiles: acc[16 positions][2 row blocks] (128 accumulator registers).  LDS: V ring of 3 x 18 KB (the next
// chunk's V is complete one phase early, so its first A operands are prefetched across the barrier), raw[2] 17 KB,
// output staging 34 KB (+ 2 image patches, conv1a weights in FIRST mode) = 107 KB.  XCD-aware: the work list is cut into 8
// contiguous ranges (workgroup g runs on XCD g % 8), so the 32 workgroups of an XCD walk neighbouring tiles and share
// halos / weights in that XCD's L2.
//
// Two kernels: conv3x3_wino6 (8 input channels per phase; also the FIRST = fused conv1a form) and conv3x3_wino6x2
// (16 per phase, the default whenever Cin % 64 == 0; see its comment).
#include "imx_kernels.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
#ifndef EXP_NOB
#define EXP_NOB 0
#endif
#ifndef EXP_NOT
#define EXP_NOT 0
#endif
#ifndef W6_TRACE
#define W6_TRACE 0
#endif
#if W6_TRACE
__device__ unsigned long long w6_trace[8];
#define W6_T(var) const unsigned long long var = __builtin_readcyclecounter();
#else
#define W6_T(var)
#endif
constexpr int TR = 4, TC = 8, OH = 2 * TR, OW = 2 * TC;
constexpr int RH = OH + 2, RW = OW + 2, RS = 12;
constexpr int CK = 8, NT = 64;
constexpr int RAW = RH * RW * RS;        // 2160
constexpr int VK = 72;                    // V channel-pair stride (64 + 8 pad: conflict-free transform writes)
constexpr int QS = 320;                   // V position stride: 5 x 64 dwords, so the consumers' reads of different positions from one
                                          // base address fuse into ds_read2st64_b32 -- no per-read address VALU in the MFMA stream
constexpr int VSZ = 16 * QS;              // 5120
constexpr int USZ = 16 * 2 * 4 * 4 * 16;  // 8192
constexpr int IMG_H = RH + 2, IMG_W = RW + 2, IMG = IMG_H * IMG_W;   // 12 x 20
constexpr int OS = NT + 4;
constexpr int OTSZ = OH * OW * OS;        // 8704
constexpr int NXCD = 8;

struct Item { int b, y0, x0, cob; };

struct Sched {
  int tiles_x, tiles_y, ncob, total, per_xcd, first, limit, stride, count;
  __device__ __forceinline__ void init(const ConvArgs& p, int g, int G) {
    tiles_x = (p.W + OW - 1) / OW;
    tiles_y = (p.H + OH - 1) / OH;
    ncob = p.Cout / NT;
    total = tiles_x * tiles_y * p.B * ncob;
    per_xcd = (total + NXCD - 1) / NXCD;
    const int xcd = g % NXCD;
    stride = G / NXCD;
    first = xcd * per_xcd + g / NXCD;
    limit = min(total, (xcd + 1) * per_xcd);
    count = first < limit ? (limit - first + stride - 1) / stride : 0;
  }
  __device__ __forceinline__ Item item(int k) const {
    int w = first + k * stride;
    Item it;
    it.cob = w % ncob; w /= ncob;
    it.x0 = (w % tiles_x) * OW; w /= tiles_x;
    it.y0 = (w % tiles_y) * OH;
    it.b = w / tiles_y;
    return it;
  }
};

// ------------------------------------------------------------------------------------------------ consumer side
// B operands (U) come straight from L2 into ONE 32-register panel that is refilled in place with the next chunk's
// values, four registers at a time, right behind the MFMAs that read them (the refill sits in the NEXT scheduling
// region so it cannot be hoisted above those MFMAs; a second panel does not fit next to 128 accumulators).  Row (pos, k-step) of a wave's panel is 256 contiguous bytes: [pos][k-step][co-block][4 k][16 co].
// Buffer loads: descriptor + block offset live in SGPRs, the only VGPR is the lane offset.  U layout (wu6):
// [k-step][pos group][co-block][lane][4 pos] -- the four B registers of one MFMA group are ONE dwordx4 per lane, so a
// chunk's panel is 8 vector-memory instructions instead of 32 (their issue cost sits in the MFMA stream).
__device__ __forceinline__ f32x4 u_load(__amdgpu_buffer_rsrc_t ur, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, voff, soff, 0));
}
__device__ __forceinline__ void load_b_panel(f32x4 (&bf)[8], __amdgpu_buffer_rsrc_t ur, int uoff, int voff) {
#pragma unroll
  for (int g = 0; g < 8; ++g) bf[g] = u_load(ur, voff, uoff + g * 4096);
}

// 16 positions x 2 k-steps x 2 row blocks of v_mfma_f32_16x16x4_f32 in 8 groups of 8 (group g = k-step g>>2, positions
// 4(g&3)..+3); A operands from LDS one group ahead; bf[g] = group g's four B registers, refilled in place with the next
// chunk's right behind the MFMAs that read them (the refill sits in the NEXT scheduling region so it cannot be hoisted
// above those MFMAs; a second panel does not fit next to 128 accumulators).
// af[0] holds this chunk's group-0 A operands on entry (prefetched during the previous chunk, across the barrier: V is a
// ring of three buffers, so the next chunk's V is complete one phase early) and the next chunk's on exit.
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[16][2], const float* Vb, const float* Vnext, float (&af)[2][4][2], f32x4 (&bf)[8],
                                           __amdgpu_buffer_rsrc_t ur, int uoff, int voff, int lane) {
  // Four element offsets per chunk (k-step 0/1 x row block 0/1), made opaque to the optimiser so that it does not fold them
  // back into one base + large immediates (which costs a v_add per read): from each base, positions q and q' are
  // q*QS = q*5*64 dwords apart and fuse into ds_read2st64_b32 with no address arithmetic in the MFMA stream.
  const int vlane = (lane >> 5) * VK + (lane & 15) * 2 + ((lane >> 4) & 1);
  int o00 = vlane, o01 = vlane + 32, o10 = vlane + 2 * VK, o11 = vlane + 2 * VK + 32;
  asm volatile("" : "+v"(o00), "+v"(o01), "+v"(o10), "+v"(o11));      // (the pointers stay LDS pointers)
  const float* b00 = Vb + o00;
  const float* b01 = Vb + o01;
  const float* b10 = Vb + o10;
  const float* b11 = Vb + o11;
  const float* n00 = Vnext + o00;
  const float* n01 = Vnext + o01;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int cur = g & 1, nxt = cur ^ 1;
    if (g + 1 < 8) {
      const int s1 = (g + 1) >> 2, qb = ((g + 1) & 3) * 4;
      const float* r0 = s1 ? b10 : b00;
      const float* r1 = s1 ? b11 : b01;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[nxt][i][0] = r0[(qb + i) * QS];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[nxt][i][1] = r1[(qb + i) * QS];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[nxt][i][0] = n00[i * QS];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[nxt][i][1] = n01[i * QS];
    }
    const int q0 = (g & 3) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[q0 + i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i][0], bf[g][i], acc[q0 + i][0], 0, 0, 0);
      acc[q0 + i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][i][1], bf[g][i], acc[q0 + i][1], 0, 0, 0);
    }
    if (g > 0) bf[g - 1] = u_load(ur, voff, uoff + (g - 1) * 4096);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read (next group's A)
      __builtin_amdgcn_sched_group_barrier(0x026, 1, 0);     // VALU / SALU / VMEM read
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf[7] = u_load(ur, voff, uoff + 7 * 4096);
}

// Y = A^T M A in-lane, bias, ReLU, (2x2 max-pool) -> LDS staging tile Ot[pixel][OS].
template <bool POOL, bool RELU>
__device__ __forceinline__ void output_transform(f32x4 (&acc)[16][2], float* Ot, const float* bias, int n0, int cb, int lane) {
  asm volatile("" : "+v"(lane));      // opaque: keeps the staging addresses from being hoisted out of the chunk loop (they would
                                       // sit in ~16 registers next to 128 accumulators + two B panels and force spills)
  const int col = cb * 16 + (lane & 15);
  const float bs = bias[n0 + col];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
#define ACC(q_) acc[q_][jb]
    // The four accumulator registers of a (position, row block) are the same lane's four wtiles r = 0..3: the whole
    // transform runs on them as f32x4 values, i.e. as v_pk_add_f32 pairs -- half the VALU instructions of a scalar loop
    // over r, with no swizzles.
    f32x4 t0[4], t1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      t0[nu] = ACC(0 * 4 + nu) + ACC(1 * 4 + nu) + ACC(2 * 4 + nu);
      t1[nu] = ACC(1 * 4 + nu) - ACC(2 * 4 + nu) - ACC(3 * 4 + nu);
    }
    const f32x4 bs4 = {bs, bs, bs, bs}, zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 y00 = t0[0] + t0[1] + t0[2] + bs4, y01 = t0[1] - t0[2] - t0[3] + bs4;
    f32x4 y10 = t1[0] + t1[1] + t1[2] + bs4, y11 = t1[1] - t1[2] - t1[3] + bs4;
    if (RELU) {
      y00 = __builtin_elementwise_max(y00, zero4); y01 = __builtin_elementwise_max(y01, zero4);
      y10 = __builtin_elementwise_max(y10, zero4); y11 = __builtin_elementwise_max(y11, zero4);
    }
    f32x4 pooled = zero4;
    if constexpr (POOL) pooled = __builtin_elementwise_max(__builtin_elementwise_max(y00, y01), __builtin_elementwise_max(y10, y11));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = jb * 16 + 4 * (lane >> 4) + r;
      const int wr = w >> 3, wc = w & 7;
      if constexpr (POOL) {
        Ot[(wr * TC + wc) * OS + col] = pooled[r];
      } else {
        float* o = Ot + ((2 * wr) * OW + 2 * wc) * OS + col;
        o[0] = y00[r];
        o[OS] = y01[r];
        o[OW * OS] = y10[r];
        o[OW * OS + OS] = y11[r];
      }
    }
    // (the accumulators are cleared for the next tile)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q][jb] = zero4;
#undef ACC
  }
}

// ------------------------------------------------------------------------------------------------ producer side
__device__ __forceinline__ f32x4 nt_load4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }

// input transform V = B^T d B: thread = (channel tc, wtile tw); raw pixel stride 12 keeps the reads conflict free.
// Split into its LDS-read half and its compute + LDS-write half so other work can sit in the read latency.
template <int RSX = RS>
__device__ __forceinline__ void input_transform_load(const float* raw, int ptid, float (&d)[16]) {
  const int tc = ptid & 7, tw = ptid >> 3, twr = tw >> 3, twc = tw & 7;
  const float* rp = raw + ((2 * twr) * RW + 2 * twc) * RSX + tc;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) d[a * 4 + bb] = rp[(a * RW + bb) * RSX];
}
__device__ __forceinline__ void input_transform_finish(const float (&d)[16], float* V, int ptid) {
  const int tc = ptid & 7, tw = ptid >> 3;
  float tt[4][4];
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    tt[0][bb] = d[0 * 4 + bb] - d[2 * 4 + bb];
    tt[1][bb] = d[1 * 4 + bb] + d[2 * 4 + bb];
    tt[2][bb] = d[2 * 4 + bb] - d[1 * 4 + bb];
    tt[3][bb] = d[1 * 4 + bb] - d[3 * 4 + bb];
  }
  float* vp = V + (tc >> 1) * VK + tw * 2 + (tc & 1);
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    vp[(xi * 4 + 0) * QS] = tt[xi][0] - tt[xi][2];
    vp[(xi * 4 + 1) * QS] = tt[xi][1] + tt[xi][2];
    vp[(xi * 4 + 2) * QS] = tt[xi][2] - tt[xi][1];
    vp[(xi * 4 + 3) * QS] = tt[xi][1] - tt[xi][3];
  }
}
__device__ __forceinline__ void input_transform(const float* raw, float* V, int ptid) {
  float d[16];
  input_transform_load(raw, ptid, d);
  input_transform_finish(d, V, ptid);
}

// staged output tile -> HBM, whole channel rows as float4.  All LDS reads are issued before the first store (one LDS
// round trip per tile, not one per row).
template <bool POOL>
__device__ __forceinline__ void store_tile(const ConvArgs& p, const Item& it, const float* Ot, int ptid) {
  constexpr int PW = POOL ? TC : OW, PH = POOL ? TR : OH;
  constexpr int N = PH * PW * (NT / 4) / 256;
  const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
  const int oy0 = POOL ? it.y0 >> 1 : it.y0, ox0 = POOL ? it.x0 >> 1 : it.x0;
  f32x4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int e = ptid + i * 256;
    v[i] = *reinterpret_cast<const f32x4*>(Ot + (e / (NT / 4)) * OS + 4 * (e % (NT / 4)));
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int e = ptid + i * 256;
    const int pix = e / (NT / 4), v4 = e % (NT / 4);
    const int oy = oy0 + pix / PW, ox = ox0 + pix % PW;
    if (oy < Ho && ox < Wo)
      __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4*>(p.out + ((size_t)(it.b * Ho + oy) * Wo + ox) * p.Cout + it.cob * NT + 4 * v4));
  }
}

template <bool POOL, bool RELU, bool FIRST>
__global__ __launch_bounds__(512) void conv3x3_wino6(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;                    // [3][VSZ] ring
  float* raw = V + 3 * VSZ;           // [2][RAW]
  float* Ot = raw + 2 * RAW;          // [OTSZ]
  float* img = Ot + OTSZ;             // FIRST: [2][IMG] + conv1a weights

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: the role branch below is wave-uniform
  Sched sc;
  sc.init(p, blockIdx.x, gridDim.x);
  if (sc.count == 0) return;
  const int nchunk = p.Cin / CK;
  const int S = sc.count * nchunk;
  const int H = p.H, W = p.W, Cin = p.Cin;

  if (wave < 4) {
    // ======================================================================================== consumers
    const int cb = wave;
    f32x4 acc[16][2];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][j][r] = 0.f;
#if W6_TRACE
    unsigned long long tr_work = 0, tr_bar = 0, tr_epi = 0;
#endif
    f32x4 bf[8];
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wu6, 0, (p.Cout / NT) * nchunk * USZ * 4, 0x00020000);
    const int voff = (cb * 64 + lane) * 16;
    auto ustep = [&](int k, int c) __attribute__((always_inline)) -> int {      // byte offset of a step's U chunk (scalar)
      if (c >= nchunk) { c -= nchunk; ++k; }
      if (k >= sc.count) { k = sc.count - 1; c = nchunk - 1; }
      return __builtin_amdgcn_readfirstlane((sc.item(k).cob * nchunk + c) * (USZ * 4));
    };
    load_b_panel(bf, ur, ustep(0, 0), voff);
    if constexpr (FIRST) __syncthreads();     // producers' prologue: 3 phases (+1 in FIRST mode)
    __syncthreads();
    __syncthreads();
    __syncthreads();
    float af[2][4][2];
    {
      const float* va = V + (lane >> 5) * VK + (lane & 15) * 2 + ((lane >> 4) & 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[0][i][0] = va[i * QS];
        af[0][i][1] = va[i * QS + 32];
      }
    }
    int vi = 0;                                   // ring position of the chunk being multiplied
    for (int k = 0; k < sc.count; ++k) {
      const Item it = sc.item(k);
      for (int c = 0; c < nchunk; ++c) {
        const int vnx = vi == 2 ? 0 : vi + 1;
        W6_T(t0)
        mfma_chunk(acc, V + vi * VSZ, V + vnx * VSZ, af, bf, ur, ustep(k, c + 1), voff, lane);
        W6_T(t1)
        if (c + 1 == nchunk) output_transform<POOL, RELU>(acc, Ot, p.bias, it.cob * NT, cb, lane);
        W6_T(t2)
        __syncthreads();
#if W6_TRACE
        const unsigned long long t3 = __builtin_readcyclecounter();
        tr_work += t1 - t0; tr_epi += t2 - t1; tr_bar += t3 - t2;
#endif
        vi = vnx;
      }
    }
#if W6_TRACE
    if (tid == 0) { atomicAdd(&w6_trace[0], tr_work); atomicAdd(&w6_trace[1], tr_bar); atomicAdd(&w6_trace[2], tr_epi); atomicAdd(&w6_trace[3], (unsigned long long)S); }
#endif
    return;
  }

  // ========================================================================================== producers
  const int ptid = tid - 256;
  __builtin_amdgcn_s_setprio(3);      // the few producer instructions must not queue behind the consumers' MFMA stream

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  if constexpr (!FIRST) {
    // this thread's two raw-patch items: e < 360: pixel e>>1, channel half e&1.  Threads without a second item repeat
    // their first one; out-of-image pixels load from a clamped address and are zeroed at the LDS write -- so every
    // load and store below is unconditional and the compiler can count the fetches in flight.
    const int e0 = ptid, e1 = ptid + 256 < RH * RW * 2 ? ptid + 256 : ptid;
    const int rpy0 = (e0 >> 1) / RW, rpx0 = (e0 >> 1) % RW, rh0 = e0 & 1;
    const int rpy1 = (e1 >> 1) / RW, rpx1 = (e1 >> 1) % RW, rh1 = e1 & 1;
    const int rdst0 = (rpy0 * RW + rpx0) * RS + 4 * rh0, rdst1 = (rpy1 * RW + rpx1) * RS + 4 * rh1;
    // fetch cursor: (item, chunk) of the next patch fetch; past the last item it keeps re-reading the last one
    int fk = 0, fc = 0;
    // Patch fetches are buffer loads: the descriptor (this item's image) and the chunk offset live in SGPRs, the thread's
    // pixel offset is one VGPR per item, and an out-of-image pixel simply gets an out-of-range offset -- the hardware
    // returns zeros (conv padding) with no clamp, mask multiply or 64-bit address arithmetic in the producers' stream.
    __amdgpu_buffer_rsrc_t frc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
    int voff0 = 0, voff1 = 0;
    const int img_bytes = H * W * Cin * 4;
    auto raw_sources = [&](const Item& it) __attribute__((always_inline)) {
      const int gy0 = it.y0 + rpy0 - 1, gx0 = it.x0 + rpx0 - 1, gy1 = it.y0 + rpy1 - 1, gx1 = it.x0 + rpx1 - 1;
      frc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)it.b * H * W * Cin), 0, img_bytes, 0x00020000);
      voff0 = (gy0 >= 0 && gy0 < H && gx0 >= 0 && gx0 < W) ? ((gy0 * W + gx0) * Cin + 4 * rh0) * 4 : img_bytes;
      voff1 = (gy1 >= 0 && gy1 < H && gx1 >= 0 && gx1 < W) ? ((gy1 * W + gx1) * Cin + 4 * rh1) * 4 : img_bytes;
    };
    raw_sources(sc.item(0));
    // A global load takes longer (~1.5 us under load) than one ~2k-cycle phase: FOUR patch fetches are kept in flight,
    // each issued four phases before the LDS write that consumes it.  Static register sets, loop unrolled by 4.
    f32x4 ra[4], rb[4];
    auto issue = [&](auto set_c) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      const int soff = __builtin_amdgcn_readfirstlane(fc * CK * 4);
      ra[SET] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frc, voff0, soff, 2));     // slc: streaming
      rb[SET] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frc, voff1, soff, 2));
      if (++fc == nchunk) {
        fc = 0;
        if (fk + 1 < sc.count) raw_sources(sc.item(++fk));
      }
    };
    auto put = [&](auto set_c, float* rbuf) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      *reinterpret_cast<f32x4*>(rbuf + rdst0) = ra[SET];
      *reinterpret_cast<f32x4*>(rbuf + rdst1) = rb[SET];
    };
    // ---- prologue: F(0..3), W(0), F(4) | T(0), W(1), F(5) | T(1), W(2), F(6)
    issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
    put(I0{}, raw);
    issue(I0{});
    __syncthreads();
    input_transform(raw, V, ptid);
    put(I1{}, raw + RAW);
    issue(I1{});
    __syncthreads();
    input_transform(raw + RAW, V + VSZ, ptid);
    put(I2{}, raw);
    issue(I2{});
    __syncthreads();
    // ---- steady state, phase s: W(s+3), F(s+7), T(s+2) into ring slot (s+2)%3, and the previous item's tile store.
    //      Everything past the last step writes buffers nobody reads (branch-free on purpose: the waits stay static).
    int k = 0, c = 0, vt = 2;             // vt = (s+2) % 3
#if W6_TRACE
    unsigned long long pt_put = 0, pt_tr = 0, pt_st = 0, pt_bar = 0;
#endif
    auto phase = [&](auto j_c) __attribute__((always_inline)) {
      constexpr int J = decltype(j_c)::value;
      W6_T(t0)
      put(std::integral_constant<int, (J + 3) & 3>{}, raw + ((J + 1) & 1) * RAW);
      issue(std::integral_constant<int, (J + 3) & 3>{});
      W6_T(t1)
      if (EXP_NOT == 0) input_transform(raw + (J & 1) * RAW, V + vt * VSZ, ptid);
      vt = vt == 2 ? 0 : vt + 1;
      W6_T(t2)
      if (c == 0 && k > 0) store_tile<POOL>(p, sc.item(k - 1), Ot, ptid);
      W6_T(t3)
      __syncthreads();
#if W6_TRACE
      { const unsigned long long t4 = __builtin_readcyclecounter(); pt_put += t1 - t0; pt_tr += t2 - t1; pt_st += t3 - t2; pt_bar += t4 - t3; }
#endif
      if (++c == nchunk) { c = 0; ++k; }
    };
    for (int s = 0; s < S; s += 4) {
      phase(I0{}); phase(I1{}); phase(I2{}); phase(I3{});
    }
#if W6_TRACE
    if (ptid == 0) { atomicAdd(&w6_trace[4], pt_put); atomicAdd(&w6_trace[5], pt_tr); atomicAdd(&w6_trace[6], pt_st); atomicAdd(&w6_trace[7], pt_bar); }
#endif
    store_tile<POOL>(p, sc.item(sc.count - 1), Ot, ptid);
  } else {
    // ---- FIRST: the raw patches are computed, not fetched: conv1a (+ folded BN + ReLU) of 8 channels over the 10x18
    //      halo patch straight into a raw buffer.  thread = (channel tc, run): 30 runs of 6 pixels (10 rows x 3);
    //      taps are float2 reads of the image patch, weights come from LDS (staged once per workgroup).
    float* w1s = img + 2 * IMG;          // [9][64] + bias [64]
    for (int e = ptid; e < 9 * 64; e += 256) w1s[e] = p.w1[e];
    if (ptid < 64) w1s[9 * 64 + ptid] = p.b1[ptid];
    // Packed over channel PAIRS: thread = (pair cp of the chunk's 8 channels, run of 3 pixels; 60 runs cover the 10x18
    // patch, threads 240..255 repeat run 59).  Both channels of a pair share every tap, so each multiply-add is one
    // v_pk_fma_f32 with the tap broadcast: half the FMA instructions per channel -- the producers' instruction count is
    // what bounds a phase beside the consumers' MFMAs.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int c1_cp = ptid & 3, c1_run = min(ptid >> 2, 59), c1_py = c1_run / 6, c1_xr = (c1_run % 6) * 3;
    struct C1 { f32x2 w[9], bias; float tap[3][5]; };
    auto conv1a_load = [&](int cch, const float* im, C1& r) __attribute__((always_inline)) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) r.w[tp] = *reinterpret_cast<const f32x2*>(w1s + tp * 64 + cch * CK + 2 * c1_cp);
      r.bias = *reinterpret_cast<const f32x2*>(w1s + 9 * 64 + cch * CK + 2 * c1_cp);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int j = 0; j < 5; ++j) r.tap[dy][j] = im[(c1_py + dy) * IMG_W + c1_xr + j];
    };
    auto conv1a_finish = [&](const Item& it, const C1& r, float* rbuf) __attribute__((always_inline)) {
      const int gy = it.y0 + c1_py - 1;
      const float rowmask = (gy >= 0 && gy < H) ? 1.f : 0.f;
      const f32x2 zero2 = {0.f, 0.f};
#pragma unroll
      for (int px = 0; px < 3; ++px) {
        f32x2 v = r.bias;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float t = r.tap[dy][px + dx];
            v = __builtin_elementwise_fma((f32x2){t, t}, r.w[dy * 3 + dx], v);
          }
        const int gx = it.x0 + c1_xr + px - 1;
        const float mask = (gx >= 0 && gx < W) ? rowmask : 0.f;
        *reinterpret_cast<f32x2*>(rbuf + (c1_py * RW + c1_xr + px) * RS + 2 * c1_cp) = __builtin_elementwise_max(v, zero2) * mask;
      }
    };
    auto img_load = [&](const Item& it) -> float {      // one pixel of the 12x20 image patch per thread (ptid < 240)
      if (ptid >= IMG) return 0.f;
      const float* im = (it.b < p.split) ? p.in + (size_t)it.b * H * W : p.in2 + (size_t)(it.b - p.split) * H * W;
      const int gy = it.y0 + ptid / IMG_W - 2, gx = it.x0 + ptid % IMG_W - 2;
      return (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.f;
    };
    int ck = 0, cc = 0;                    // compute cursor: (item, chunk) of the next conv1a; past the end it repeats the last
    Item ccit = sc.item(0);                // item's chunks into buffers nobody reads (branch-free)
    auto advance = [&]() __attribute__((always_inline)) {
      if (++cc == nchunk) { cc = 0; if (ck + 1 < sc.count) ccit = sc.item(++ck); }
    };
    auto compute_next = [&](float* rbuf) __attribute__((always_inline)) {
      C1 r;
      conv1a_load(cc, img + (ck & 1) * IMG, r);
      conv1a_finish(ccit, r, rbuf);
      advance();
    };
    // ---- prologue: image patch + weights | C(0) | T(0), C(1) | T(1), C(2)
    float ipix = img_load(ccit);
    if (ptid < IMG) img[ptid] = ipix;
    __syncthreads();
    compute_next(raw);
    __syncthreads();
    input_transform(raw, V, ptid);
    compute_next(raw + RAW);
    __syncthreads();
    input_transform(raw + RAW, V + VSZ, ptid);
    compute_next(raw);
    __syncthreads();
    // ---- steady state, phase s: T(s+2) into ring slot (s+2)%3, C(s+3); the next item's image patch is fetched at
    //      chunk 2 and stored at chunk 4 (its first conv1a runs at chunk nchunk-3)
    int k = 0, c = 0, vt = 2;
    auto phase = [&](auto j_c) __attribute__((always_inline)) {
      constexpr int J = decltype(j_c)::value;     // = s & 1
      // T(s+2) and C(s+3) interleaved: both LDS read batches first, then the two compute + write halves
      float d[16];
      C1 r;
      input_transform_load(raw + J * RAW, ptid, d);
      conv1a_load(cc, img + (ck & 1) * IMG, r);
      input_transform_finish(d, V + vt * VSZ, ptid);
      conv1a_finish(ccit, r, raw + (J ^ 1) * RAW);
      advance();
      vt = vt == 2 ? 0 : vt + 1;
      if (c == 2 && k + 1 < sc.count) ipix = img_load(sc.item(k + 1));
      if (c == 4 && k + 1 < sc.count && ptid < IMG) img[((k + 1) & 1) * IMG + ptid] = ipix;
      if (c == 0 && k > 0) store_tile<POOL>(p, sc.item(k - 1), Ot, ptid);
      __syncthreads();
      if (++c == nchunk) { c = 0; ++k; }
    };
    for (int s = 0; s < S; s += 2) { phase(I0{}); phase(I1{}); }
    store_tile<POOL>(p, sc.item(sc.count - 1), Ot, ptid);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Sixteen input channels per phase (two 8-channel chunks back to back, ONE barrier per 128 MFMAs).  The producers are
// bound by dependent LDS round trips (read patch -> add -> write V), not by issue slots: with two independent
// transforms per thread and phase those round trips overlap, and the barrier count halves.  Same arithmetic and U / V
// layouts as conv3x3_wino6; V is double-buffered ([2 buffers][2 halves]), raw patches hold 16 channels per pixel
// (stride 20: conflict-free reads).  LDS 144 KB.  Not for the fused first layer.
constexpr int RS2 = 20, RAW2 = RH * RW * RS2;      // 3600
template <bool POOL, bool RELU>
__global__ __launch_bounds__(512) void conv3x3_wino6x2(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;                    // [2][2][VSZ]
  float* raw = V + 4 * VSZ;           // [2][RAW2]
  float* Ot = raw + 2 * RAW2;         // [OTSZ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Sched sc;
  sc.init(p, blockIdx.x, gridDim.x);
  if (sc.count == 0) return;
  const int nchunk = p.Cin / CK, nphase = nchunk / 2;
  const int S = sc.count * nphase;
  const int H = p.H, W = p.W, Cin = p.Cin;

  if (wave < 4) {
    // ======================================================================================== consumers
    const int cb = wave;
    f32x4 acc[16][2];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][j][r] = 0.f;
    f32x4 bf[8];
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wu6, 0, (p.Cout / NT) * nchunk * USZ * 4, 0x00020000);
    const int voff = (cb * 64 + lane) * 16;
    auto ustep = [&](int k, int c) __attribute__((always_inline)) -> int {      // byte offset of a chunk's U block (scalar)
      if (c >= nchunk) { c -= nchunk; ++k; }
      if (k >= sc.count) { k = sc.count - 1; c = nchunk - 1; }
      return __builtin_amdgcn_readfirstlane((sc.item(k).cob * nchunk + c) * (USZ * 4));
    };
    load_b_panel(bf, ur, ustep(0, 0), voff);
    __syncthreads();
    __syncthreads();
    float af[2][4][2];
    const int vlane = (lane >> 5) * VK + (lane & 15) * 2 + ((lane >> 4) & 1);
    int buf = 0;
    for (int k = 0; k < sc.count; ++k) {
      const Item it = sc.item(k);
      for (int c = 0; c < nphase; ++c) {
        const float* v0 = V + buf * 2 * VSZ;
#pragma unroll
        for (int i = 0; i < 4; ++i) {               // this phase's first A operands (its V became visible at the barrier)
          af[0][i][0] = v0[vlane + i * QS];
          af[0][i][1] = v0[vlane + i * QS + 32];
        }
        mfma_chunk(acc, v0, v0 + VSZ, af, bf, ur, ustep(k, 2 * c + 1), voff, lane);
        mfma_chunk(acc, v0 + VSZ, v0 + VSZ, af, bf, ur, ustep(k, 2 * c + 2), voff, lane);
        if (c + 1 == nphase) output_transform<POOL, RELU>(acc, Ot, p.bias, it.cob * NT, cb, lane);
        __syncthreads();
        buf ^= 1;
      }
    }
    return;
  }

  // ========================================================================================== producers
  const int ptid = tid - 256;
  __builtin_amdgcn_s_setprio(3);
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // this thread's three raw-patch items: e < 720: pixel e>>2, channel quarter e&3 (threads without a third repeat their first)
  int rdst[3], rpy[3], rpx[3], rq[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = ptid + 256 * i < RH * RW * 4 ? ptid + 256 * i : ptid;
    rpy[i] = (e >> 2) / RW; rpx[i] = (e >> 2) % RW; rq[i] = e & 3;
    rdst[i] = (rpy[i] * RW + rpx[i]) * RS2 + 4 * rq[i];
  }
  int fk = 0, fc = 0;
  __amdgpu_buffer_rsrc_t frc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
  int fvoff[3] = {0, 0, 0};
  const int img_bytes = H * W * Cin * 4;
  auto raw_sources = [&](const Item& it) __attribute__((always_inline)) {
    frc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)it.b * H * W * Cin), 0, img_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int gy = it.y0 + rpy[i] - 1, gx = it.x0 + rpx[i] - 1;
      fvoff[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? ((gy * W + gx) * Cin + 4 * rq[i]) * 4 : img_bytes;
    }
  };
  raw_sources(sc.item(0));
  f32x4 ra[4][3];
  auto issue = [&](auto set_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int soff = __builtin_amdgcn_readfirstlane(fc * 2 * CK * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) ra[SET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frc, fvoff[i], soff, 2));
    if (++fc == nphase) {
      fc = 0;
      if (fk + 1 < sc.count) raw_sources(sc.item(++fk));
    }
  };
  auto put = [&](auto set_c, float* rbuf) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(rbuf + rdst[i]) = ra[SET][i];
  };
  auto transform2 = [&](const float* rbuf, float* vbuf) __attribute__((always_inline)) {
    float d0[16], d1[16];                          // two independent transforms: their LDS round trips overlap
    input_transform_load<RS2>(rbuf, ptid, d0);
    input_transform_load<RS2>(rbuf + CK, ptid, d1);
    input_transform_finish(d0, vbuf, ptid);
    input_transform_finish(d1, vbuf + VSZ, ptid);
  };
  // ---- prologue: F(0..3), W(0), F(4) | T(0), W(1), F(5)
  issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
  put(I0{}, raw);
  issue(I0{});
  __syncthreads();
  transform2(raw, V);
  put(I1{}, raw + RAW2);
  issue(I1{});
  __syncthreads();
  // ---- steady state, phase s: W(s+2), F(s+6), T(s+1), and the previous item's tile store
  int k = 0, c = 0;
  auto phase = [&](auto j_c) __attribute__((always_inline)) {
    constexpr int J = decltype(j_c)::value;
    put(std::integral_constant<int, (J + 2) & 3>{}, raw + (J & 1) * RAW2);
    issue(std::integral_constant<int, (J + 2) & 3>{});
    transform2(raw + ((J + 1) & 1) * RAW2, V + ((J + 1) & 1) * 2 * VSZ);
    if (c == 0 && k > 0) store_tile<POOL>(p, sc.item(k - 1), Ot, ptid);
    __syncthreads();
    if (++c == nphase) { c = 0; ++k; }
  };
  for (int s = 0; s < S; s += 4) {
    phase(I0{}); phase(I1{}); phase(I2{}); phase(I3{});
  }
  store_tile<POOL>(p, sc.item(sc.count - 1), Ot, ptid);
}

template <bool POOL, bool RELU>
hipError_t launch_x2(const ConvArgs& a, hipStream_t s) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, dev);
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount / NXCD * NXCD : 256;
  }
  const size_t lds = (size_t)(4 * VSZ + 2 * RAW2 + OTSZ) * sizeof(float);
  auto k = conv3x3_wino6x2<POOL, RELU>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)ncu), dim3(512), lds, s, a);
  return hipGetLastError();
}

template <bool POOL, bool RELU, bool FIRST>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, dev);
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount / NXCD * NXCD : 256;
  }
  const size_t lds = (size_t)(3 * VSZ + 2 * RAW + OTSZ + (FIRST ? 2 * IMG + 10 * 64 : 0)) * sizeof(float);
  auto k = conv3x3_wino6<POOL, RELU, FIRST>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
#if W6_TRACE
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(w6_trace), z, sizeof(z));
#endif
  hipLaunchKernelGGL(k, dim3((unsigned)ncu), dim3(512), lds, s, a);
#if W6_TRACE
  (void)hipStreamSynchronize(s);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(w6_trace), sizeof(z));
  const double n = (double)z[3];
  fprintf(stderr, "[w6 trace] H=%d Cin=%d Cout=%d first=%d | consumer/phase: mfma %.0f barrier %.0f epilogue %.0f | producer/phase: put+issue %.0f transform %.0f store %.0f barrier %.0f\n",
          a.H, a.Cin, a.Cout, (int)FIRST, z[0] / n, z[1] / n, z[2] / n, z[4] / n, z[5] / n, z[6] / n, z[7] / n);
#endif
  return hipGetLastError();
}
}  // namespace

hipError_t launch_conv3x3_wino6(const ConvArgs& a, hipStream_t s) {
  if (a.Cin % (2 * CK) || a.Cout % NT || (a.first && a.Cin != 64) || !a.wu6) return hipErrorInvalidValue;
  if (a.first) return a.pool ? launch_t<true, true, true>(a, s) : launch_t<false, true, true>(a, s);
  const bool x1 = getenv("IMX_WINO6_X1") != nullptr;       // A/B: 8 channels per phase (read per launch: tests switch it)
  if (!x1 && a.Cin % 64 == 0) {
    if (a.pool) return a.relu ? launch_x2<true, true>(a, s) : launch_x2<true, false>(a, s);
    return a.relu ? launch_x2<false, true>(a, s) : launch_x2<false, false>(a, s);
  }
  if (a.pool) return a.relu ? launch_t<true, true, false>(a, s) : launch_t<true, false, false>(a, s);
  return a.relu ? launch_t<false, true, false>(a, s) : launch_t<false, false, false>(a, s);
}

}  // namespace imx
