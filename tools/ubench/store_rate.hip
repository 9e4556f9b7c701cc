// store_rate.hip -- what a CU's burst of 16-byte buffer stores costs in ISSUE time (round 5; conv3x3_wino24p's epilogue: the waves that
// store second spent 5.4 k cycles issuing eight buffer_store_dwordx4).  Every wave of a 512-thread workgroup issues NST stores of
// 1 KB back to back, then idles ~20 k cycles; cycles from the first store to the end of the last one's issue, slowest wave, by
//   pattern  0: a wave-instruction writes 1 KB contiguous        1: 32-byte pieces 128 bytes apart (the blocked activation layout)
//            2: 64-byte pieces 256 bytes apart (NHWC, 64 channels)
//            3: conv3x3_wino24p's own addresses (240 x 320 x 64 channels blocked by 8: per instruction 32-byte pieces in 4 rows of 2 planes)
//            4, 5: the same bytes after a register <-> lane transpose (4: 16-byte pieces, 4 rows of one plane; 5: 512 contiguous bytes of 2 planes)
//   grid     1 ... 256 workgroups (is the limit the CU's or the chip's?)
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/store_rate.hip -o tools/ubench/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned su32x4 __attribute__((__vector_size__(16)));

template <int NST>
__global__ __launch_bounds__(512) void burst(float* out, long long* cyc, int pattern, int iters, int idle, size_t wg_bytes, unsigned WP, unsigned PLANE) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = pattern >= 3 ? __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffff0, 0x00020000)
                                                      : __builtin_amdgcn_make_buffer_rsrc((void*)((char*)out + (size_t)blockIdx.x * wg_bytes), 0, (int)wg_bytes, 0x00020000);
  f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
  long long worst = 0, sum = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned offs[NST];            // (addresses first: the timed region is the stores alone)
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      unsigned off;
      const unsigned base = (unsigned)(((it & 7) * 8 + wave) * NST + k);     // a fresh 1 KB (or its footprint) per store
      if (pattern == 0) off = base * 1024u + lane * 16u;
      else if (pattern == 1) off = (base >> 2) * 4096u + (base & 3) * 32u + (lane >> 1) * 128u + (lane & 1) * 16u;
      else if (pattern == 2) off = (base >> 2) * 4096u + (base & 3) * 64u + (lane >> 2) * 256u + (lane & 3) * 16u;
      else {
        // image = 240 x 320, 8 planes of 8 channels (2457600 B each); tile pair index tp -> (image, tile row, tile column pair)
        const unsigned tp = (unsigned)(blockIdx.x + it * gridDim.x), tile = tp * 2 + (wave >> 2);
        const unsigned tx = tile % 20u, ty = (tile / 20u) % 30u;           // (one image's worth of addresses is enough: 19.7 MB)
        const unsigned lwc = lane & 3, lwr = (lane >> 2) & 3, half = (lane >> 4) & 1, pl = (wave & 3) * 2 + (lane >> 5);
        const unsigned r = k >> 2, x = k & 3;
        if (pattern == 3) off = pl * PLANE + ((ty * 8 + 2 * lwr + r) * WP + tx * 16 + 4 * lwc + x) * 32u + half * 16u;
        else if (pattern == 4) {   // x <-> (half, plane) by v_permlane16/32_swap: lane bits 4-5 = x, registers = (r, half, plane)
          const unsigned xl = lane >> 4, hf = k & 1, p2 = (wave & 3) * 2 + ((k >> 1) & 1);
          off = p2 * PLANE + ((ty * 8 + 2 * lwr + r) * WP + tx * 16 + 4 * lwc + xl) * 32u + hf * 16u;
        } else if (pattern == 6) { // as 5, every wave in the same two planes (waves -> different tile rows)
          const unsigned xl = (lane >> 2) & 3, wr = k & 3, tyw = (ty + 3 * wave) % 30u;
          off = (lane >> 5) * PLANE + ((tyw * 8 + 2 * wr + r) * WP + tx * 16 + 4 * lwc + xl) * 32u + half * 16u;
        } else if (pattern == 7) { // as 5 with ONE plane per instruction: lanes = 32 pixels x half (1 KB contiguous), registers = (r, wtile row, plane)
          const unsigned px = lane >> 1, hf = lane & 1, wr = k & 3;
          off = ((wave & 3) * 2 + (wave >> 2)) * PLANE + ((ty * 8 + 2 * wr + r) * WP + (tx & ~1u) * 16 + px) * 32u + hf * 16u;
        } else if (pattern == 8) { // the staged blocked form: lanes in address order, two rows of 16 pixels x 32 bytes (2 x 512 contiguous bytes)
          const unsigned row = 2 * (k & 3) + (lane >> 5), px = (lane >> 1) & 15, hf = lane & 1;
          off = ((wave & 3) * 2 + (k >> 2)) * PLANE + ((ty * 8 + row) * WP + tx * 16 + px) * 32u + hf * 16u;
        } else if (pattern == 9) { // the staged NHWC form: 16 pixels x 64 bytes, a quad of lanes per pixel, pixels 2 KB apart (512 channels)
          off = ((ty * 8 + k) * WP + tx * 16 + (lane >> 2)) * 2048u + (wave & 3) * 64u + (lane & 3) * 16u;
        } else {                   // x <-> wtile row (a 4 x 4 transpose by DPP): lanes = (wtile column, x, half, plane), registers = (r, wtile row)
          const unsigned xl = (lane >> 2) & 3, wr = k & 3;
          off = pl * PLANE + ((ty * 8 + 2 * wr + r) * WP + tx * 16 + 4 * lwc + xl) * 32u + half * 16u;
        }
      }
      offs[k] = off;
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) asm volatile("" : "+v"(offs[k]));
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), rs, (int)offs[k], 0, 0);
      v[1] += 1.f;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    sum += t1 - t0;
    worst = t1 - t0 > worst ? t1 - t0 : worst;
    for (int i = 0; i < idle; ++i) __builtin_amdgcn_s_sleep(16);
  }
  if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = sum / iters; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = worst; }
}

int main(int argc, char** argv) {
  const int iters = 32, idle = argc > 1 ? atoi(argv[1]) : 20;
  const size_t wg_bytes = 8 * 8 * 8 * 4096;      // (pattern 3 addresses one 19.7-MB image from every workgroup: descriptor below)      // 8 generations x 8 waves x 8 stores x (1 KB footprint up to 4 KB)
  float* out = nullptr; long long* cyc = nullptr;
  if (hipMalloc(&out, (size_t)2200 << 20) != hipSuccess || hipMalloc(&cyc, 256 * 8 * 2 * sizeof(long long)) != hipSuccess || !out || !cyc) {   // (pattern 9 spans 157 MB per image; the descriptor covers 2 GB)
    printf("allocation failed\n");
    return 1;
  }
  struct Geo { unsigned wp, plane; };
  const Geo geos[] = {{320, 2457600}};
  for (int pattern : {0, 3, 7, 8, 9})
    for (const Geo& ge : geos) {
      for (int grid : {1, 256}) {
        hipMemset(cyc, 0, 256 * 8 * 2 * sizeof(long long));
        hipLaunchKernelGGL(burst<8>, dim3(grid), dim3(512), 0, 0, out, cyc, pattern, iters, idle, wg_bytes, ge.wp, ge.plane);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 8 * 2);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double first = 0, last = 0;
        for (int g = 0; g < grid; ++g) {
          long long lo = 1LL << 60, hi = 0;
          for (int w = 0; w < 8; ++w) { lo = h[(g * 8 + w) * 2] < lo ? h[(g * 8 + w) * 2] : lo; hi = h[(g * 8 + w) * 2] > hi ? h[(g * 8 + w) * 2] : hi; }
          first += (double)lo / grid; last += (double)hi / grid;
        }
        printf("pattern %d row pitch %3u px plane %8u B grid %3d: fastest wave %6.0f cycles, slowest %6.0f  (%.1f cycles per store instruction of the CU's 64; %.1f B/clk/CU)\n", pattern, ge.wp, ge.plane, grid,
               first, last, last / 64, 65536.0 / last);
      }
    }
  return 0;
}
