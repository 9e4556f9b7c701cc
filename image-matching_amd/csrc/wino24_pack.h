// wino24_pack.h -- host side of the Winograd F(2x4, 3x3) kernels: the transformed weights U = G2 g G4^T in the layouts the kernels read.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace imx {

inline double wino24_u(const double (&g)[3][3], int i, int j) {
  static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  static const double G4[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  double acc = 0.0;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) acc += G2[i][ky] * g[ky][kx] * G4[j][kx];
  return acc;
}

// U = G2 g G4^T (4 x 6 per (co, ci); F(2,3) down the rows, F(4,3) along the columns), laid out as conv1ab_wino24.hip's MFMA
// B fragments read it: [chunk of 8 ci][quad = position pair][co-block][lane = (ci pair)*16 + co%16][(position parity)*2 + ci%2],
// position p = j*4 + i: a lane's four B registers of a quad are one buffer_load_dwordx4.
inline std::vector<float> wino24_transform(const std::vector<float>& w, int cin, int cout) {
  static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  static const double G4[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  const int nchunk = cin / 8;
  std::vector<float> u((size_t)(cout / 64) * nchunk * 12288, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      double g[3][3];
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
      const int chunk = ci / 8, kk = ci % 8, k = kk >> 1, sstep = kk & 1;
      const int cog = co / 64, col = co % 64;
      float* blk = u.data() + ((size_t)cog * nchunk + chunk) * 12288;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j) {
          double acc = 0.0;
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) acc += G2[i][ky] * g[ky][kx] * G4[j][kx];
          const int pos = j * 4 + i, quad = pos >> 1, e = (pos & 1) * 2 + sstep;
          blk[(size_t)(((quad * 4 + (col >> 4)) * 64 + k * 16 + (col & 15)) * 4) + e] = (float)acc;
        }
    }
  return u;
}


// fp32 -> fp16 bit pattern, round to nearest even (subnormals and overflow to infinity included)
inline uint16_t f16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                     // rounds to 65536 or more
  if (u < 0x38800000u) {                                                       // below 2^-14: subnormal result
    if (u < 0x33000000u) return (uint16_t)sign;                                // below 2^-25
    const int e = (int)(u >> 23);
    const uint32_t m = (u & 0x7fffffu) | 0x800000u;
    const int sh = 126 - e;                                                    // 14 .. 24
    const uint32_t q = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t r = u + 0xfffu + ((u >> 13) & 1u);
  return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = sign;
    else { float f = (float)m * 5.9604644775390625e-8f; memcpy(&u, &f, 4); u |= sign; }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// conv3x3_wino24h.hip: U s_u as two fp16 planes (U s_u = h + m, round to nearest), s_u = the power of two that brings max |U| to
// [2^13, 2^14); layout [Cout/64][Cin/32][position p = j*4 + i][plane][wave = co/16 % 4][lane = (ci%32 / 8)*16 + co%16][ci % 8].
// w: [9][cin][cout].  Returns the bit patterns; *scale_inv = 1 / s_u.
// *spread (optional): max |U| over the median over the output channels of their own max |U| -- how far below the layer's one scale a
// typical channel's weights sit (imx_api.cpp's guard: beyond 2^14 the scaled values of a typical channel lose their low plane)
inline std::vector<uint16_t> wino24h_pack(const std::vector<float>& w, int cin, int cout, float* scale_inv, float* spread = nullptr) {
  const int nchunk = cin / 32;
  std::vector<double> U((size_t)cout * cin * 24);
  std::vector<double> comax(cout, 0.0);
  double umax = 0.0;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      double g[3][3];
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j) {
          const double u = wino24_u(g, i, j);
          U[((size_t)co * cin + ci) * 24 + j * 4 + i] = u;
          if (std::fabs(u) > umax) umax = std::fabs(u);
          if (std::fabs(u) > comax[co]) comax[co] = std::fabs(u);
        }
    }
  if (spread) {
    std::vector<double> m = comax;
    std::nth_element(m.begin(), m.begin() + m.size() / 2, m.end());
    const double med = m[m.size() / 2];
    *spread = med > 0 ? (float)(umax / med) : (umax > 0 ? 3.0e38f : 1.f);
  }
  int e = 0;
  if (umax > 0) std::frexp(umax, &e);                     // umax = f 2^e, f in [0.5, 1)
  const double su = std::ldexp(1.0, 14 - e);              // umax su in [2^13, 2^14)
  *scale_inv = (float)(1.0 / su);
  std::vector<uint16_t> out((size_t)(cout / 64) * nchunk * 24 * 2 * 4 * 64 * 8, 0);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int p = 0; p < 24; ++p) {
        const float x = (float)(U[((size_t)co * cin + ci) * 24 + p] * su);
        const uint16_t h = f16_rne(x), m = f16_rne(x - f16_to_f32(h));
        const int cob = co / 64, wave = (co % 64) / 16, lane = ((ci % 32) / 8) * 16 + co % 16;
        const size_t base = (((size_t)cob * nchunk + ci / 32) * 24 + p) * 2;
        out[(((base + 0) * 4 + wave) * 64 + lane) * 8 + ci % 8] = h;
        out[(((base + 1) * 4 + wave) * 64 + lane) * 8 + ci % 8] = m;
      }
  return out;
}

}  // namespace imx
