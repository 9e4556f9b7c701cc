// mfma_bf16x3.hip -- can the bf16 MFMA (16x the fp32 MFMA rate on gfx950) carry an fp32 product?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_bf16x3.hip -o tools/ubench/mfma_bf16x3 && tools/ubench/mfma_bf16x3
// An fp32 value splits EXACTLY into three bf16 terms x = h + m + l (8 significant bits each, round-to-nearest residuals).
// a.b = sum of nine term products; the six largest (hh, hm, mh, hl, lh, mm) leave |error| <= ~2^-25 |a b|, below half an
// fp32 ulp.  Each bf16 x bf16 product is exact in fp32; what is NOT documented is how v_mfma_f32_32x32x16_bf16 adds its
// sixteen products and the accumulator.  This program measures:
//   1. the operand layout assumed by the kernels (integer data, exact compare);
//   2. deterministic probes of the internal sum (is a small addend lost against a large product of the SAME instruction?);
//   3. error against a double-precision product, rms / max / mean signed, for: the fp32 MFMA chain the library uses today,
//      the 6-product split into ONE accumulator (small terms first), the same into three accumulators by magnitude class,
//      and a 3-product split (hh, hm, mh: "bf16x2"-grade) for scale;
//   4. issue rate: cycles per MFMA for a stream of independent bf16 MFMAs, alone and with 2/4/6 VALU fillers per MFMA
//      (the fp32 MFMA shares the VALU datapath -- tools/ubench/mfma_valu.hip -- the bf16 one should not).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// D[32][32] = A[32][K] * B[K][32]; lane (i = l & 31, kb = l >> 5) feeds k = 16 s + 8 kb + j, j = 0..7, to both operands
template <int MODE>     // 0: 6 products one accumulator, 1: 6 products three accumulators, 2: 3 products, 3: hh only
__global__ void k_split(const float* A, const float* B, float* D, int K) {
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  f32x16 acc = {0}, acc1 = {0}, acc2 = {0};
  for (int s = 0; s < K / 16; ++s) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * kb + j;
      __bf16 h, m, lo;
      split3(A[i * K + k], h, m, lo); ah[j] = h; am[j] = m; al[j] = lo;
      split3(B[k * 32 + i], h, m, lo); bh[j] = h; bm[j] = m; bl[j] = lo;
    }
    if (MODE == 0) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    } else if (MODE == 1) {
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    } else if (MODE == 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  if (MODE == 1) acc = acc + (acc1 + acc2);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = acc[r];
}

__global__ void k_f32(const float* A, const float* B, float* D, int K) {
  const int l = threadIdx.x, i = l & 31, hi = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + hi], B[(k + hi) * 32 + i], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + i] = acc[r];
}

// probes of the internal sum.  All values are bf16-exact.  Row 0 / col 0 of D is read.
//  p0: acc = 0; products {2^24 (k=0), 1, 1, ... (15 ones)}       exact: 2^24+15.  fp32 RNE of the exact sum: 2^24+16.
//      sequential fp32 adds from the big one: 2^24.
//  p1: acc = 2^24, sixteen products of 1                          exact 2^24+16 (representable)
//  p2: acc = 1, products {2^-24 x 16}  (sum 2^-20)                exact 1 + 2^-20 (representable: ulp(1) = 2^-23)
//  p3: acc = 1, one product 0.75 * 2^-23                          RNE: 1 + 2^-23, truncation: 1
//  p4: acc = -1, one product (1 + 2^-7)*(1 + 2^-7) = 1 + 2^-6 + 2^-14   exact result 2^-6 + 2^-14 (is the product kept exact?)
__global__ void k_probe(float* out) {
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  auto run = [&](auto fa, auto fb, float c0) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { const int k = 8 * kb + j; a[j] = (__bf16)(i == 0 ? fa(k) : 0.f); b[j] = (__bf16)(i == 0 ? fb(k) : 0.f); }
    f32x16 c; for (int r = 0; r < 16; ++r) c[r] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    return c[0];
  };
  const float r0 = run([](int k) { return k == 0 ? 4096.f : 1.f; }, [](int k) { return k == 0 ? 4096.f : 1.f; }, 0.f);
  const float r1 = run([](int) { return 1.f; }, [](int) { return 1.f; }, 16777216.f);
  const float r2 = run([](int) { return 0.000244140625f; }, [](int) { return 0.000244140625f; }, 1.f);   // 2^-12 squared
  const float r3 = run([](int k) { return k == 0 ? 0.75f : 0.f; }, [](int k) { return k == 0 ? 1.1920929e-07f : 0.f; }, 1.f);
  const float r4 = run([](int k) { return k == 0 ? 1.0078125f : 0.f; }, [](int k) { return k == 0 ? 1.0078125f : 0.f; }, -1.f);
  // p5: the big product in the LAST k slot instead of the first (order dependence inside the instruction)
  const float r5 = run([](int k) { return k == 15 ? 4096.f : 1.f; }, [](int k) { return k == 15 ? 4096.f : 1.f; }, 0.f);
  // p6: acc = 2^24, fifteen products of 1 and one of 0 -> exact 2^24 + 15 -> RNE 2^24+16, truncation 2^24+14
  const float r6 = run([](int k) { return k == 3 ? 0.f : 1.f; }, [](int) { return 1.f; }, 16777216.f);
  if (l == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; out[4] = r4; out[5] = r5; out[6] = r6; }
}

// layout check: A[i][k] = small integers, B[k][n] likewise: exact in bf16 and in the fp32 sum
__global__ void k_layout(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[i * 16 + 8 * kb + j]; b[j] = (__bf16)B[(8 * kb + j) * 32 + i]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = c[r];
}

// issue rate: NACC independent accumulators, FILL VALU instructions (v_fma_f32 on private registers) per MFMA
template <int FILL, int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
  f32x16 c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  float f[8];
  for (int q = 0; q < 8; ++q) f[q] = (float)threadIdx.x * 0.001f + q;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[q], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < FILL; ++v) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[v]) : "v"(f[(v + 1) & 7]));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[v]));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[v]) : "v"(f[(v + 1) & 7]));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  for (int q = 0; q < 8; ++q) s += f[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *dA, *dB, *dD; long long* dC;
  hipMalloc(&dA, 32 * 1024 * 4); hipMalloc(&dB, 32 * 1024 * 4); hipMalloc(&dD, 8 * 1024 * 4 + 256 * 1024 * 4); hipMalloc(&dC, 64);
  {  // 1. layout
    std::vector<float> A(32 * 16), B(16 * 32), D(1024);
    srand(3);
    for (auto& v : A) v = (float)(rand() % 15 - 7);
    for (auto& v : B) v = (float)(rand() % 15 - 7);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
      float ref = 0; for (int k = 0; k < 16; ++k) ref += A[i * 16 + k] * B[k * 32 + n];
      bad += D[i * 32 + n] != ref;
    }
    printf("layout: %d of 1024 outputs differ from the exact integer product (operand lane (i, kb) holds k = 8 kb .. 8 kb + 7)\n", bad);
  }
  {  // 2. probes
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dD);
    float o[7]; hipMemcpy(o, dD, 28, hipMemcpyDeviceToHost);
    printf("probe p0 (2^24 first + 15 ones, acc 0): got 2^24%+.0f   (exact +15; RNE of exact +16; sequential fp32 +0)\n", o[0] - 16777216.f);
    printf("probe p5 (2^24 LAST  + 15 ones, acc 0): got 2^24%+.0f\n", o[5] - 16777216.f);
    printf("probe p1 (acc 2^24 + 16 ones)         : got 2^24%+.0f   (exact +16)\n", o[1] - 16777216.f);
    printf("probe p6 (acc 2^24 + 15 ones)         : got 2^24%+.0f   (exact +15; RNE +16; truncation +14)\n", o[6] - 16777216.f);
    printf("probe p2 (acc 1 + 16 x 2^-24)         : got 1%+.3e   (exact +9.537e-07 = 2^-20)\n", (double)o[2] - 1.0);
    printf("probe p3 (acc 1 + 0.75 ulp)           : got 1%+.3e   (RNE +1.192e-07; truncation +0)\n", (double)o[3] - 1.0);
    printf("probe p4 (acc -1 + (1+2^-7)^2)        : got %.10e   (exact 1.5686035156e-02 = 2^-6 + 2^-14)\n", (double)o[4]);
  }
  for (int K : {32, 128, 256, 1024}) {  // 3. accuracy
    for (int dist = 0; dist < 2; ++dist) {     // 0: uniform(-1,1) both; 1: positive operands (sum grows: relative test)
      std::vector<float> A(32 * K), B(K * 32);
      srand(11 + K + dist);
      for (auto& v : A) v = dist ? rand() / (float)RAND_MAX : (rand() / (float)RAND_MAX) * 2.f - 1.f;
      for (auto& v : B) v = dist ? rand() / (float)RAND_MAX : (rand() / (float)RAND_MAX) * 2.f - 1.f;
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      hipMemset(dD, 0, 5 * 1024 * 4);
      hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
      hipLaunchKernelGGL(k_split<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD + 1024, K);
      hipLaunchKernelGGL(k_split<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD + 2048, K);
      hipLaunchKernelGGL(k_split<2>, dim3(1), dim3(64), 0, 0, dA, dB, dD + 3072, K);
      hipLaunchKernelGGL(k_split<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD + 4096, K);
      std::vector<float> D(5 * 1024);
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      const char* names[5] = {"fp32 mfma 32x32x2", "bf16 x3, 6 prod, 1 acc", "bf16 x3, 6 prod, 3 acc", "bf16 x3, 3 prod", "bf16 x1 (hh)"};
      for (int v = 0; v < 5; ++v) {
        double se = 0, me = 0, mx = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
          double ref = 0;
          for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[k * 32 + j];
          const double e = (double)D[v * 1024 + i * 32 + j] - ref;
          se += e * e; me += e; mx = fmax(mx, fabs(e));
        }
        printf("K=%4d %-8s %-24s rms %.3e  max %.3e  mean signed %+.3e\n", K, dist ? "positive" : "signed", names[v], sqrt(se / 1024), mx, me / 1024);
      }
    }
  }
  {  // 4. issue rate (1 wave per SIMD: 256 threads, 1 workgroup per CU)
    const int iters = 2000;
    auto rate = [&](auto kern, const char* what) {
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, dD + 8192, dC, iters);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, dD + 8192, dC, iters);
      long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
      printf("rate: %-34s %.1f s_memtime ticks per MFMA (100 MHz ticks? see ratio to the bare stream)\n", what, (double)c / (iters * 4.0));
    };
    rate(k_rate<0, 0>, "bare bf16 MFMA stream");
    rate(k_rate<2, 0>, "+2 v_fma_f32 per MFMA");
    rate(k_rate<4, 0>, "+4 v_fma_f32 per MFMA");
    rate(k_rate<6, 0>, "+6 v_fma_f32 per MFMA");
    rate(k_rate<8, 0>, "+8 v_fma_f32 per MFMA");
    rate(k_rate<1, 1>, "+1 v_exp_f32 per MFMA");
    rate(k_rate<2, 1>, "+2 v_exp_f32 per MFMA");
    rate(k_rate<4, 2>, "+4 v_cvt_pk_bf16_f32 per MFMA");
  }
  return 0;
}
