// conv1ab_wino24.hip — the fused first layer (conv1a + conv1b + folded BN + ReLU + MaxPool2d(2);
// superpoint/models/unet_parts.py:10-48, superpoint_test.py:113-114) with conv1b as Winograd F(2x4, 3x3):
//   Y = A2^T [ (G2 g G4^T) (.) (B2^T d B4) ] A4     per 2-row x 4-column output tile ("wtile") and 4x6 input patch d
// (F(2,3) down the rows, F(4,3) along the columns: 24 multiplies per 8 outputs, 3x fewer than the direct form and
// 1.33x fewer than F(2x2,3x3); fp32 throughout, error measured in the parity tests).  Summed over input channels as 24
// independent GEMMs  M_p[wtile][co] = sum_ci V_p[wtile][ci] U_p[ci][co]  on v_mfma_f32_16x16x4_f32.
//
// Workgroup = 256 threads (4 waves) -> 8x16 output pixels = 4x4 wtiles x 64 output channels; wave = 16 channels x the 16
// wtiles x 24 positions = 24 accumulators of 4 VGPRs.  The 16x16 D layout gives a lane one channel and one ROW of four
// wtiles with all 24 positions: output transform, bias, ReLU and the 2x2 max-pool (one wtile = 1x2 pooled pixels) are
// in-lane.
// conv1a (1 -> 64, K = 9) is evaluated once per workgroup for the 10x18 halo patch into LDS (packed over channel pairs).
// K loop: 8 input channels per chunk, two barriers per chunk.
//   input transform: ONE wave per chunk (wave ch & 3; the others wait at the barrier and leave their SIMD to the
//     co-resident workgroup's MFMAs): lane = (wtile, channel PAIR), patch read as 24 ds_read_b64, every operation of both
//     passes is one packed (v_pk_*_f32) instruction over the pair, results leave as 12 ds_write_b128;
//   V layout [12 quads][4 k][18][4]: a quad = two positions x the two channels of a pair.  The chunk's two MFMA k-steps
//     take the even / odd channels of the four pairs, so a lane's A operands of four MFMAs are ONE ds_read_b128
//     (12 LDS reads per chunk for 48 MFMAs), and the B operands (U = G2 g G4^T, transformed at weight load, laid out
//     [chunk][quad][co-block][lane][4]) ONE buffer_load_dwordx4, issued before the chunk barrier: U never touches LDS.
// LDS 62 KB, 2 workgroups per CU.
#include "imx_kernels.h"

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int OH = 8, OW = 16;                 // output pixels per workgroup (4 x 4 wtiles of 2 x 4)
constexpr int RH = OH + 2, RW = OW + 2;        // conv1a patch (pad-1 halo)
constexpr int IMG_H = RH + 2, IMG_W = RW + 2;  // image patch 12 x 20
constexpr int RS = 66;                         // conv1a patch pixel stride: wtile columns 4 px apart land 8 banks apart
constexpr int RAWSZ = RH * RW * RS;            // 11880
constexpr int KS = 18;                         // V: 16-byte slots per k index (16 wtiles + 2: conflict-free b128 writes)
constexpr int QSL = 4 * KS;                    // slots per quad
constexpr int NQ = 12;                         // quads per chunk: 24 positions x 2 k-steps / 4
constexpr int VSZ = NQ * QSL * 4;              // 3456
constexpr int UCH = NQ * 4 * 64 * 4;           // 12288 floats of U per (64 co, 8 ci)
constexpr int CK = 8, NT = 64, OS = NT + 4;

__global__ __launch_bounds__(256, 2) void conv1ab_wino24(ConvArgs p, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;
  float* raw = V + VSZ;
  float* img = raw + RAWSZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * OW, y0 = ty * OH;
  const int H = p.H, W = p.W, Cout = p.Cout;
  constexpr int nchunk = 64 / CK;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wu24, 0, nchunk * UCH * 4, 0x00020000);
  const int voff = (cb * 64 + lane) * 16;

  // ---- conv1a + folded BN + ReLU for all 64 channels of the 10x18 halo patch, packed over channel pairs: lane = (pair,
  //      one of 8 pixel groups), a group takes runs g, g+8, .. of the 30 six-pixel runs; every multiply-add is one
  //      v_pk_fma_f32 with the tap broadcast.  Positions outside the image are conv1b's zero padding (mask multiply).
  {
    const float* im = (b < p.split) ? p.in + (size_t)b * H * W : p.in2 + (size_t)(b - p.split) * H * W;
    const int c1_cp = tid & 31, c1_g = tid >> 5;
    f32x2 wr[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wr[tp] = *reinterpret_cast<const f32x2*>(p.w1 + tp * 64 + 2 * c1_cp);
    const f32x2 bias = *reinterpret_cast<const f32x2*>(p.b1 + 2 * c1_cp);
    for (int e = tid; e < IMG_H * IMG_W; e += 256) {
      const int py = e / IMG_W, px = e % IMG_W;
      const int gy = y0 + py - 2, gx = x0 + px - 2;
      img[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    const f32x2 zero2 = {0.f, 0.f};
#pragma unroll 1
    for (int run = c1_g; run < 30; run += 8) {
      const int py = run / 3, xr = (run % 3) * 6;
      float tap[3][8];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float2 tv = *reinterpret_cast<const float2*>(img + (py + dy) * IMG_W + xr + 2 * j);
          tap[dy][2 * j] = tv.x;
          tap[dy][2 * j + 1] = tv.y;
        }
      const int gy = y0 + py - 1;
      const float rowmask = (gy >= 0 && gy < H) ? 1.f : 0.f;
#pragma unroll
      for (int px = 0; px < 6; ++px) {
        f32x2 v = bias;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float tt = tap[dy][px + dx];
            v = __builtin_elementwise_fma((f32x2){tt, tt}, wr[dy * 3 + dx], v);
          }
        const int gx = x0 + xr + px - 1;
        const float mask = (gx >= 0 && gx < W) ? rowmask : 0.f;
        *reinterpret_cast<f32x2*>(raw + (py * RW + xr + px) * RS + 2 * c1_cp) = __builtin_elementwise_max(v, zero2) * mask;
      }
    }
  }

  f32x4 acc[24];
#pragma unroll
  for (int q = 0; q < 24; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transform lane = (channel pair tk, wtile tw): a quarter-wave (16 lanes) = 4 pairs x the 4 wtiles of one wtile row
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const float* rbase = raw + ((2 * twr) * RW + 4 * twc) * RS + 2 * tk;
  float* vwr = V + (tk * KS + tw) * 4;
  int aoff = ((lane >> 4) * KS + (lane & 15)) * 4;
  asm volatile("" : "+v"(aoff));               // opaque: the 12 quad reads are immediate offsets from one base
  const float* vrd = V + aoff;

  for (int ch = 0; ch < nchunk; ++ch) {
    // ---- this chunk's B operands: 12 buffer_load_dwordx4 (SGPR descriptor + offset), landing while the wave waits at
    //      the barriers and the transform runs
    f32x4 bf[NQ];
    {
      const int uoff = __builtin_amdgcn_readfirstlane(ch * (UCH * 4));
#pragma unroll
      for (int g = 0; g < NQ; ++g)
        bf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, voff, uoff + g * 4096, 0));
    }
    __syncthreads();               // previous chunk's MFMA phase is done with V (first chunk: conv1a patch complete)
    if (wave == (ch & 3)) {
      // ---- input transform  V = B2^T d B4  for (wtile tw, channels ch*8 + 2 tk, +1)
      const float* rp = rbase + ch * CK;
      f32x2 T[4][6];
#pragma unroll
      for (int a = 0; a < 4; ++a) {            // along the columns: F(4,3)  B4^T
        f32x2 d[6];
#pragma unroll
        for (int bb = 0; bb < 6; ++bb) d[bb] = *reinterpret_cast<const f32x2*>(rp + (a * RW + bb) * RS);
        const f32x2 e42 = d[4] - 4.f * d[2], e31 = d[3] - 4.f * d[1];
        const f32x2 f42 = d[4] - d[2], f31 = d[3] - d[1];
        T[a][0] = 4.f * d[0] - 5.f * d[2] + d[4];
        T[a][1] = e42 + e31;
        T[a][2] = e42 - e31;
        T[a][3] = f42 + 2.f * f31;
        T[a][4] = f42 - 2.f * f31;
        T[a][5] = 4.f * d[1] - 5.f * d[3] + d[5];
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {            // down the rows: F(2,3)  B2^T ; position p = j*4 + i
        const f32x2 o0 = T[0][j] - T[2][j], o1 = T[1][j] + T[2][j];
        const f32x2 o2 = T[2][j] - T[1][j], o3 = T[1][j] - T[3][j];
        *reinterpret_cast<f32x4*>(vwr + (2 * j) * QSL * 4) = (f32x4){o0.x, o0.y, o1.x, o1.y};
        *reinterpret_cast<f32x4*>(vwr + (2 * j + 1) * QSL * 4) = (f32x4){o2.x, o2.y, o3.x, o3.y};
      }
    }
    __syncthreads();
    // ---- 24 positions x 2 k-steps of v_mfma_f32_16x16x4_f32 in 12 quads: A operands one ds_read_b128 per quad, one
    //      quad ahead; B operands already in registers
    {
      f32x4 af[2];
      af[0] = *reinterpret_cast<const f32x4*>(vrd);
#pragma unroll
      for (int g = 0; g < NQ; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < NQ) af[nxt] = *reinterpret_cast<const f32x4*>(vrd + (g + 1) * QSL * 4);
        acc[2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][0], bf[g][0], acc[2 * g], 0, 0, 0);
        acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][2], bf[g][2], acc[2 * g + 1], 0, 0, 0);
        acc[2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][1], bf[g][1], acc[2 * g], 0, 0, 0);
        acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][3], bf[g][3], acc[2 * g + 1], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read (next quad's A operands)
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- output transform Y = A2^T M A4, 2x2 max-pool, bias, ReLU (max-pool commutes with both), LDS-staged float4 stores.
  //      acc[j*4 + i][r]: wtile (row lane>>4, column r), channel cb*16 + (lane&15).
  float* Ot = smem;
  __syncthreads();          // every wave is done with V (the staging tile aliases it)
  {
    const int col = cb * 16 + (lane & 15);
    const float bs = p.bias[col];
    f32x4 s0[6], s1[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      s0[j] = acc[j * 4 + 0] + acc[j * 4 + 1] + acc[j * 4 + 2];
      s1[j] = acc[j * 4 + 1] - acc[j * 4 + 2] - acc[j * 4 + 3];
    }
    f32x4 pooled[2];
    {
      f32x4 y[2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x4* m = r ? s1 : s0;
        const f32x4 a12 = m[1] + m[2], b12 = m[1] - m[2], c34 = m[3] + m[4], d34 = m[3] - m[4];
        y[r][0] = m[0] + a12 + c34;
        y[r][1] = b12 + 2.f * d34;
        y[r][2] = a12 + 4.f * c34;
        y[r][3] = b12 + 8.f * d34 + m[5];
      }
      pooled[0] = __builtin_elementwise_max(__builtin_elementwise_max(y[0][0], y[0][1]), __builtin_elementwise_max(y[1][0], y[1][1]));
      pooled[1] = __builtin_elementwise_max(__builtin_elementwise_max(y[0][2], y[0][3]), __builtin_elementwise_max(y[1][2], y[1][3]));
    }
    const f32x4 bs4 = {bs, bs, bs, bs}, zero4 = {0.f, 0.f, 0.f, 0.f};
    const int wr = lane >> 4;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4 v = __builtin_elementwise_max(pooled[hh] + bs4, zero4);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ot[(wr * (OW / 2) + 2 * r + hh) * OS + col] = v[r];
    }
  }
  __syncthreads();
  const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
  for (int it = 0; it < (OH / 2) * (OW / 2) * (NT / 4) / 256; ++it) {
    const int e = tid + it * 256;
    const int pix = e / (NT / 4), v4 = e % (NT / 4);
    const int oy = (y0 >> 1) + pix / (OW / 2), ox = (x0 >> 1) + pix % (OW / 2);
    if (oy < Ho && ox < Wo)
      *reinterpret_cast<float4*>(p.out + ((size_t)(b * Ho + oy) * Wo + ox) * Cout + 4 * v4) =
          *reinterpret_cast<const float4*>(Ot + pix * OS + 4 * v4);
  }
}
}  // namespace

hipError_t launch_conv1ab_wino24(const ConvArgs& a, hipStream_t s) {
  if (!a.first || !a.pool || a.Cin != 64 || a.Cout != 64 || !a.wu24) return hipErrorInvalidValue;
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
  const size_t lds = (size_t)(VSZ + RAWSZ + IMG_H * IMG_W) * sizeof(float);
  hipLaunchKernelGGL(conv1ab_wino24, dim3((unsigned)(tiles_x * tiles_y * a.B)), dim3(256), lds, s, a, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace imx
