// conv3x3_wino.hip — 3x3 / pad 1 convolution by Winograd F(2x2,3x3) on the fp32 matrix cores.
//
// Same contract as conv3x3.hip (replaces F.conv2d + folded BatchNorm + ReLU (+ MaxPool2d(2)) of
// superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-123), 2.25x fewer multiplies:
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile ("wtile") and 4x4 input patch d,
// summed over input channels as 16 independent GEMMs  M_p[wtile][co] = sum_ci V_p[wtile][ci] U_p[ci][co]
// (p = 4x4 transform position).  U = G g G^T is computed once at weight load (imx_api.cpp).
// All arithmetic is fp32 (v_mfma_f32_16x16x4_f32); the transforms only use +,- and (in U) 1/2,
// so the result differs from the direct form by fp32 rounding only (measured in the parity tests).
//
// Workgroup = 256 threads (4 waves) -> 4x8 wtiles (8x16 output pixels) x 64 output channels.
//   wave: i-block ib (16 wtiles) x co-group cg (32 channels = 2 MFMA column blocks) x all 16
//   positions -> 32 accumulators of 4 VGPRs.  The 16x16 MFMA's D layout gives each lane one output
//   channel and 4 wtiles with all 16 positions: the output transform, bias, ReLU and the 2x2
//   max-pool (one wtile = one pooled pixel) are in-lane, no LDS exchange.
// K loop: 8 input channels per chunk:
//   regs (prefetched behind the previous chunk's MFMAs) -> raw patch [10][20][9] and U [16][4][4][2][16]
//   -> input transform (one (wtile, channel) per thread, 32 adds) -> V [16][4][32][2]
//   -> 64 MFMAs per wave.  LDS 57 KB -> 2 workgroups per CU so one group's staging/transform
//   overlaps the other's matrix work.  V/U layouts make every MFMA operand read 32 consecutive floats.
// FIRST mode fuses conv1a (1->64, K=9) into the raw-patch staging, as in conv3x3.hip.
#include "imx_kernels.h"

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TR = 4, TC = 8, OH = 2 * TR, OW = 2 * TC;   // wtiles / output pixels per workgroup
constexpr int RH = OH + 2, RW = OW + 2;                   // raw patch (pad 1 halo)
constexpr int RPITCH = 20, RS = 9;                        // raw patch LDS pitch / pixel stride (2-way max)
constexpr int CK = 8, NT = 64;
constexpr int RAW = RH * RPITCH * RS;                     // 1800
constexpr int RAW_PAD = (RAW + 3) & ~3;
constexpr int VSZ = 16 * 4 * 32 * 2;                      // 4096
constexpr int USZ = 16 * 4 * 4 * 2 * 16;                  // 8192 = 16 positions x 8 ci x 64 co
constexpr int IMG_H = RH + 2, IMG_W = RW + 2;             // FIRST: image patch 12 x 20

template <bool POOL, bool RELU, bool FIRST>
__global__ __launch_bounds__(256, 2) void conv3x3_wino(ConvArgs p, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;
  float* V = smem + RAW_PAD;
  float* U = V + VSZ;
  float* img = U + USZ;   // FIRST only

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ib = wave & 1, cg = wave >> 1;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * OW, y0 = ty * OH;
  const int n0 = blockIdx.y * NT;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CK;
  const float* ublk = p.wu + (size_t)blockIdx.y * nchunk * USZ;

  if constexpr (FIRST) {
    const float* im = (b < p.split) ? p.in + (size_t)b * H * W : p.in2 + (size_t)(b - p.split) * H * W;
    for (int e = tid; e < IMG_H * IMG_W; e += 256) {
      const int py = e / IMG_W, px = e % IMG_W;
      const int gy = y0 + py - 2, gx = x0 + px - 2;
      img[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.f;
    }
  }

  f32x4 acc[16][2];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][j][r] = 0.f;

  // ---- global -> register prefetch (named registers; issued before the previous chunk's MFMAs)
  float4 r0, r1, u0, u1, u2, u3, u4, u5, u6, u7;
#define IMX_GR(reg_, it_)                                                                              \
  {                                                                                                    \
    const int e = tid + (it_) * 256;                                                                   \
    reg_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                            \
    if (e < RH * RW * 2) {                                                                             \
      const int pix = e >> 1, half = e & 1, py = pix / RW, px = pix % RW;                              \
      const int gy = y0 + py - 1, gx = x0 + px - 1;                                                    \
      if (gy >= 0 && gy < H && gx >= 0 && gx < W)                                                      \
        reg_ = *reinterpret_cast<const float4*>(p.in + ((size_t)(b * H + gy) * W + gx) * Cin + cc + 4 * half); \
    }                                                                                                  \
  }
#define IMX_SR(reg_, it_)                                                                              \
  {                                                                                                    \
    const int e = tid + (it_) * 256;                                                                   \
    if (e < RH * RW * 2) {                                                                             \
      const int pix = e >> 1, half = e & 1, py = pix / RW, px = pix % RW;                              \
      float* d = raw + (py * RPITCH + px) * RS + 4 * half;                                             \
      d[0] = reg_.x; d[1] = reg_.y; d[2] = reg_.z; d[3] = reg_.w;                                      \
    }                                                                                                  \
  }
#define IMX_GU(reg_, it_) reg_ = *reinterpret_cast<const float4*>(uc + (tid + (it_) * 256) * 4);
#define IMX_SU(reg_, it_) *reinterpret_cast<float4*>(U + (tid + (it_) * 256) * 4) = reg_;
#define IMX_GLOAD(chunk_)                                                                              \
  {                                                                                                    \
    const int cc = (chunk_) * CK;                                                                      \
    const float* uc = ublk + (size_t)(chunk_) * USZ;                                                   \
    (void)cc;                                                                                          \
    if constexpr (!FIRST) { IMX_GR(r0, 0) IMX_GR(r1, 1) }                                              \
    IMX_GU(u0, 0) IMX_GU(u1, 1) IMX_GU(u2, 2) IMX_GU(u3, 3) IMX_GU(u4, 4) IMX_GU(u5, 5) IMX_GU(u6, 6) IMX_GU(u7, 7) \
  }
#define IMX_LSTORE()                                                                                   \
  {                                                                                                    \
    if constexpr (!FIRST) { IMX_SR(r0, 0) IMX_SR(r1, 1) }                                              \
    IMX_SU(u0, 0) IMX_SU(u1, 1) IMX_SU(u2, 2) IMX_SU(u3, 3) IMX_SU(u4, 4) IMX_SU(u5, 5) IMX_SU(u6, 6) IMX_SU(u7, 7) \
  }

  IMX_GLOAD(0)
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();               // previous chunk's MFMA phase is done with raw / V / U
    if constexpr (FIRST) {
      // conv1a + folded BN + ReLU for channels [8ch, 8ch+8) at the 10x18 halo pixels; positions
      // outside the image are conv1b's zero padding.
      if (tid < RH * RW) {
        const int py = tid / RW, px = tid % RW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const int cb = ch * CK;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          float im9[9];
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) im9[tp] = img[(py + tp / 3) * IMG_W + px + tp % 3];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = p.b1[cb + j];
#pragma unroll
          for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(im9[tp], p.w1[tp * 64 + cb + j], v[j]);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        float* d = raw + (py * RPITCH + px) * RS;
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = v[j];
      }
    }
    IMX_LSTORE()
    __syncthreads();
    // ---- input transform  V = B^T d B : thread = (wtile w, channel c)
    {
      const int w = tid & 31, c = tid >> 5;
      const int wr = w >> 3, wc = w & 7;
      const float* rp = raw + ((2 * wr) * RPITCH + 2 * wc) * RS + c;
      float d[4][4], tt[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) d[a][bb] = rp[(a * RPITCH + bb) * RS];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {       // rows: B^T d
        tt[0][bb] = d[0][bb] - d[2][bb];
        tt[1][bb] = d[1][bb] + d[2][bb];
        tt[2][bb] = d[2][bb] - d[1][bb];
        tt[3][bb] = d[1][bb] - d[3][bb];
      }
      float* vp = V + ((c >> 1) * 32 + w) * 2 + (c & 1);
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {       // columns: (B^T d) B ; position p = xi*4 + nu
        vp[((xi * 4 + 0) * 4) * 64] = tt[xi][0] - tt[xi][2];
        vp[((xi * 4 + 1) * 4) * 64] = tt[xi][1] + tt[xi][2];
        vp[((xi * 4 + 2) * 4) * 64] = tt[xi][2] - tt[xi][1];
        vp[((xi * 4 + 3) * 4) * 64] = tt[xi][1] - tt[xi][3];
      }
    }
    __syncthreads();
    // prefetch the next chunk (branch-free: the last iteration re-fetches its own chunk, unused)
    IMX_GLOAD(ch + 1 < nchunk ? ch + 1 : ch)
    // ---- 16 positions x 2 k-steps x 2 column blocks of v_mfma_f32_16x16x4_f32, operands one step ahead
    {
      const float* va = V + (lane >> 5) * 64 + (ib * 16 + (lane & 15)) * 2 + ((lane >> 4) & 1);
      const float* ub = U + (lane >> 5) * 128 + cg * 64 + ((lane >> 4) & 1) * 16 + (lane & 15);
      float af[2], bf[2][2];
      af[0] = va[0];
      bf[0][0] = ub[0];
      bf[0][1] = ub[32];
#pragma unroll
      for (int st = 0; st < 32; ++st) {
        const int cur = st & 1, nxt = cur ^ 1;
        const int q = st & 15;                  // position
        if (st + 1 < 32) {
          const int q1 = (st + 1) & 15, s1 = (st + 1) >> 4;
          af[nxt] = va[(q1 * 4 + 2 * s1) * 64];
          bf[nxt][0] = ub[(q1 * 4 + 2 * s1) * 128];
          bf[nxt][1] = ub[(q1 * 4 + 2 * s1) * 128 + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur], bf[cur][0], acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur], bf[cur][1], acc[q][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#undef IMX_GR
#undef IMX_SR
#undef IMX_GU
#undef IMX_SU
#undef IMX_GLOAD
#undef IMX_LSTORE

  // ---- output transform Y = A^T M A, bias, ReLU, (2x2 max-pool), store.
  //      acc[p][jb][r]: wtile ib*16 + 4*(lane>>4) + r, channel n0 + cg*32 + jb*16 + (lane&15).
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int co = n0 + cg * 32 + jb * 16 + (lane & 15);
    const float bs = p.bias[co];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = ib * 16 + 4 * (lane >> 4) + r;
      const int wr = w >> 3, wc = w & 7;
      float m[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) m[q] = acc[q][jb][r];
      float t0[4], t1[4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        t0[nu] = m[0 * 4 + nu] + m[1 * 4 + nu] + m[2 * 4 + nu];
        t1[nu] = m[1 * 4 + nu] - m[2 * 4 + nu] - m[3 * 4 + nu];
      }
      float y00 = t0[0] + t0[1] + t0[2] + bs, y01 = t0[1] - t0[2] - t0[3] + bs;
      float y10 = t1[0] + t1[1] + t1[2] + bs, y11 = t1[1] - t1[2] - t1[3] + bs;
      if (RELU) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
      if constexpr (POOL) {
        const int Ho = H >> 1, Wo = W >> 1;
        const int oy = (y0 >> 1) + wr, ox = (x0 >> 1) + wc;
        if (oy < Ho && ox < Wo)
          p.out[((size_t)(b * Ho + oy) * Wo + ox) * Cout + co] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
      } else {
        const int oy = y0 + 2 * wr, ox = x0 + 2 * wc;
        float* o = p.out + ((size_t)(b * H + oy) * W + ox) * Cout + co;
        if (oy < H) {
          if (ox < W) o[0] = y00;
          if (ox + 1 < W) o[Cout] = y01;
        }
        if (oy + 1 < H) {
          if (ox < W) o[(size_t)W * Cout] = y10;
          if (ox + 1 < W) o[(size_t)W * Cout + Cout] = y11;
        }
      }
    }
  }
}

template <bool POOL, bool RELU, bool FIRST>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
  dim3 grid((unsigned)(tiles_x * tiles_y * a.B), (unsigned)(a.Cout / NT));
  size_t lds = (size_t)(RAW_PAD + VSZ + USZ + (FIRST ? IMG_H * IMG_W : 0)) * sizeof(float);
  auto k = conv3x3_wino<POOL, RELU, FIRST>;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, tiles_x, tiles_y);
  return hipGetLastError();
}
}  // namespace

hipError_t launch_conv3x3_wino(const ConvArgs& a, hipStream_t s) {
  if (a.Cin % CK || a.Cout % NT || (a.first && a.Cin != 64) || !a.wu) return hipErrorInvalidValue;
  if (a.first) return a.pool ? launch_t<true, true, true>(a, s) : launch_t<false, true, true>(a, s);
  if (a.pool) return a.relu ? launch_t<true, true, false>(a, s) : launch_t<true, false, false>(a, s);
  return a.relu ? launch_t<false, true, false>(a, s) : launch_t<false, false, false>(a, s);
}

}  // namespace imx
