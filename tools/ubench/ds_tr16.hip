// ds_tr16.hip -- what does ds_read_b64_tr_b16 return?  LDS holds lds[e] = e (bf16-exact for e < 256); every lane passes the
// address of 4 consecutive elements.  Test A: lane-linear addresses (lane l -> elements 4l..4l+3).  Test B: the image the
// attention kernel uses: a row-major [key][32 dims] tile, lane l = (hi, g, i) -> &V[key0(hi) + (i >> 2)][16 g + 4 (i & 3)].
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/ds_tr16.hip -o tools/ubench/ds_tr16 && tools/ubench/ds_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds_p;
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[512];
  for (int e = threadIdx.x; e < 512; e += 64) sm[e] = (__bf16)(float)(e & 255);
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = (l >> 4) & 1, hi = l >> 5;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sm + 4 * l));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sm + (4 * hi + (i >> 2)) * 32 + 16 * g + 4 * (i & 3)));
  // (extracting the elements of the v4i16 result one by one miscompiles under hipcc 7.2 -- every element reads as element 0;
  //  the whole vector is reinterpreted as two dwords instead, which is also how the kernels pass it on to the MFMA)
  const u32x2 aw = __builtin_bit_cast(u32x2, a), bw = __builtin_bit_cast(u32x2, b);
  for (int j = 0; j < 4; ++j) {
    out[l * 4 + j] = __uint_as_float(((aw[j >> 1] >> (16 * (j & 1))) & 0xffffu) << 16);
    out[256 + l * 4 + j] = __uint_as_float(((bw[j >> 1] >> (16 * (j & 1))) & 0xffffu) << 16);
  }
}
int main() {
  float* d; hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("A: lane-linear addresses (lane l passes &lds[4 l]); lane: its four results\n");
  for (int l = 0; l < 64; ++l) printf("%2d: %3.0f %3.0f %3.0f %3.0f%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   ");
  printf("B: row-major [key][32] tile, lane (hi, g, i) passes &V[4 hi + (i >> 2)][16 g + 4 (i & 3)]; wanted: lane gets V[4 hi + j][16 g + i] = 32 (4 hi + j) + 16 g + i\n");
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += h[256 + l * 4 + j] != (float)(32 * (4 * (l >> 5) + j) + 16 * ((l >> 4) & 1) + (l & 15));
  printf("B: %d of 256 differ from the wanted transpose\n", bad);
  return 0;
}
