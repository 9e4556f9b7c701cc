// Image ingest and the warp post-step of the registration CLIs, as HBM-bound byte kernels for gfx950.
//
//   resize_u8_unit  : cv2.resize(uint8 gray, INTER_LINEAR) followed by `/255`          datasets/SSHIDataset.py:19-27
//   warp_affine_u8  : cv2.warpAffine(source_original*255, Matrix, (w,h)) -> imwrite    superpoint_glue_test.py:101-113
//
// OpenCV is a third-party dependency absent from the reference tree (README.md:24 pins 4.5.1.48): both kernels follow
// its published fixed-point algorithms (restated in oracle/ingest_ref.py, parity vs cv2 itself unpinned) and are
// bit-exact against that restatement: the arithmetic is integer / exactly-rounded double.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "imx_kernels.h"

namespace imx {
namespace {

// 11-bit bilinear coefficients of one output coordinate (OpenCV resize.cpp, INTER_LINEAR, 8U fixed-point path):
// f = float((d+0.5)*scale - 0.5); s = floor(f); f -= s; edges clamp to weight 0; short(rint(w*2048)).
__device__ __forceinline__ void lin_coef(int d, double scale, int n, bool clamp_weight, int& s, int& a0, int& a1) {
#pragma clang fp contract(off)
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
  if (clamp_weight) {                       // horizontal pass: out-of-range taps collapse onto the border pixel
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
  }
  a0 = __float2int_rn((1.f - f) * 2048.f);
  a1 = __float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(256) void resize_u8_unit_kernel(const uint8_t* __restrict__ src, long sstride, int Hs, int Ws,
                                                             float* __restrict__ dst, int H, int W, double sx, double sy) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const uint8_t* im = src + (size_t)blockIdx.z * sstride;
  int cx, a0, a1, cy, b0, b1;
  lin_coef(x, sx, Ws, true, cx, a0, a1);
  lin_coef(y, sy, Hs, false, cy, b0, b1);
  const int y0 = min(max(cy, 0), Hs - 1), y1 = min(max(cy + 1, 0), Hs - 1);   // vertical taps clamp rows, keep weights
  const int x1 = min(cx + 1, Ws - 1);
  const int r0 = im[(size_t)y0 * Ws + cx] * a0 + im[(size_t)y0 * Ws + x1] * a1;
  const int r1 = im[(size_t)y1 * Ws + cx] * a0 + im[(size_t)y1 * Ws + x1] * a1;
  const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
  const int u = min(max(v, 0), 255);
  dst[((size_t)blockIdx.z * H + y) * W + x] = (float)((double)u / 255.0);      // uint8/255 in float64, then .float()
}

// cv2.warpAffine, INTER_LINEAR, BORDER_CONSTANT 0, on the float64 image `u8/255*255`: fixed-point source coordinates
// with 10+5 bits (imgwarp.cpp: AB_BITS 10, INTER_BITS 5), float table weights, double accumulation; the file write
// converts with round-half-even + saturation.  Minv is the inverted 2x3 matrix (host, double).
struct WarpM { double m[6]; };

__device__ __forceinline__ int sat_int(double v) { return (int)rint(fmin(fmax(v, -2147483648.0), 2147483647.0)); }

__global__ __launch_bounds__(256) void warp_affine_u8_kernel(const uint8_t* __restrict__ src, int Hs, int Ws,
                                                             uint8_t* __restrict__ dst, int H, int W, WarpM M) {
#pragma clang fp contract(off)      // OpenCV's host arithmetic has no fused multiply-add: keep every rounding
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int X0 = sat_int((M.m[1] * y + M.m[2]) * 1024.0) + 16, Y0 = sat_int((M.m[4] * y + M.m[5]) * 1024.0) + 16;
  const int X = (X0 + sat_int(M.m[0] * x * 1024.0)) >> 5, Y = (Y0 + sat_int(M.m[3] * x * 1024.0)) >> 5;
  const int ix = X >> 5, iy = Y >> 5;                       // arithmetic shifts: floor
  const float fx = (float)(X & 31) * (1.f / 32.f), fy = (float)(Y & 31) * (1.f / 32.f);
  const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
  auto px = [&](int yy, int xx) -> double {
    if ((unsigned)yy >= (unsigned)Hs || (unsigned)xx >= (unsigned)Ws) return 0.0;
    return (double)src[(size_t)yy * Ws + xx] / 255.0 * 255.0;              // `source_original/255` then `*255` in float64
  };
  const double v = px(iy, ix) * (double)w00 + px(iy, ix + 1) * (double)w01 + px(iy + 1, ix) * (double)w10 + px(iy + 1, ix + 1) * (double)w11;
  dst[(size_t)y * W + x] = (uint8_t)min(max((int)rint(v), 0), 255);
}

}  // namespace

hipError_t launch_resize_u8_unit(const uint8_t* src, long sstride, int B, int Hs, int Ws, float* dst, int H, int W, hipStream_t s) {
  const double sx = 1.0 / ((double)W / Ws), sy = 1.0 / ((double)H / Hs);   // scale = 1/inv_scale as OpenCV derives it
  hipLaunchKernelGGL(resize_u8_unit_kernel, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(256), 0, s, src, sstride, Hs, Ws, dst, H, W, sx, sy);
  return hipGetLastError();
}

hipError_t launch_warp_affine_u8(const uint8_t* src, int Hs, int Ws, uint8_t* dst, int H, int W, const double* Minv, hipStream_t s) {
  WarpM M;
  for (int i = 0; i < 6; ++i) M.m[i] = Minv[i];
  hipLaunchKernelGGL(warp_affine_u8_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, src, Hs, Ws, dst, H, W, M);
  return hipGetLastError();
}

}  // namespace imx
