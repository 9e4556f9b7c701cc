// wg_dispatch.hip -- how does the chip place workgroups of a given shape?  (round 6's first question about the C5 Sinkhorn launch:
// profiles/r05_v6_c5_pmc_limiter.json shows 2.9 of 8 waves resident per SIMD over a 42-us launch of 520 workgroups x 16 waves x 74 KB of LDS
// whose registers / LDS / wave count allow 512 of them at once.)
//
// Every workgroup stamps the 100-MHz wall clock when it starts, reads where it runs (XCC, SE, CU), keeps its CU slot for `hold_us`
// (touching its LDS so the allocation is real), stamps again.  The host prints, per shape:
//   * the launch's span, the time at which the first / median / last workgroup STARTED (a dispatch ramp shows here),
//   * the largest number of workgroups alive at once on one CU and on the chip (do two 74-KB workgroups share a CU?),
//   * workgroups per XCC (round-robin remainder = who runs a second round).
// build:  hipcc -O2 --offload-arch=gfx950 tools/ubench/wg_dispatch.hip -o /tmp/wg_dispatch
// run:    /tmp/wg_dispatch                    (the shapes below)   or   /tmp/wg_dispatch <nwg> <waves> <lds_bytes> <hold_us> [vgprs: 64|128]
// NOT part of the library (outside its build id).  Round 5's run: profiles/r05_wg_dispatch.txt -- two 74-KB 16-wave workgroups share a CU, all 512
// start within 1.7 us, 520 are dispatched in 2.7 us: dispatch is not what a Sinkhorn iteration waits for; 520 against 512 costs one more
// workgroup life (33.0 against 18.1 us at 15 us of hold).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x)                                                                                     \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }       \
  } while (0)

struct Stamp {
  unsigned long long t0, t1;
  unsigned xcc, hw;
};

// VG = 1: <= 64 registers (eight waves per SIMD, sinkhorn_slab's budget); VG = 2: a body that keeps ~100 values live (four per SIMD)
template <int VG>
__global__ void hold_kernel(Stamp* out, unsigned long long hold_ticks, int lds_words) {
  extern __shared__ float lds[];
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  float acc[VG == 2 ? 96 : 8];
#pragma unroll
  for (int k = 0; k < (VG == 2 ? 96 : 8); ++k) acc[k] = (float)(threadIdx.x + k);
  for (int j = threadIdx.x; j < lds_words; j += blockDim.x) lds[j] = (float)j;
  __syncthreads();
  unsigned it = 0;
  while ((unsigned long long)wall_clock64() - t0 < hold_ticks) {
    const int j = (threadIdx.x * 33 + it * 64) % lds_words;
#pragma unroll
    for (int k = 0; k < (VG == 2 ? 96 : 8); ++k) acc[k] = fmaf(acc[k], 0.999f, lds[(j + k) % lds_words]);
    ++it;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < (VG == 2 ? 96 : 8); ++k) s += acc[k];
  if (s == 12345.678f) lds[0] = s;      // (keeps acc alive)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = Stamp{t0, (unsigned long long)wall_clock64(), xcc & 0xfu, hw};
  }
}

static void run(int nwg, int waves, int lds_bytes, double hold_us, int vg) {
  Stamp* d;
  CHECK(hipMalloc(&d, sizeof(Stamp) * nwg));
  auto kern = vg == 2 ? hold_kernel<2> : hold_kernel<1>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int per_cu = -1;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * waves, (size_t)lds_bytes));
  const unsigned long long ticks = (unsigned long long)(hold_us * 100.0);      // wall_clock64: 100 MHz
  for (int rep = 0; rep < 2; ++rep) {                                          // the second launch is the one reported (code resident)
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * waves), (size_t)lds_bytes, 0, d, ticks, lds_bytes / 4);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<Stamp> h(nwg);
  CHECK(hipMemcpy(h.data(), d, sizeof(Stamp) * nwg, hipMemcpyDeviceToHost));
  CHECK(hipFree(d));
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& s : h) { tmin = std::min(tmin, s.t0); tmax = std::max(tmax, s.t1); }
  std::vector<double> starts;
  for (auto& s : h) starts.push_back((s.t0 - tmin) / 100.0);
  std::sort(starts.begin(), starts.end());
  // alive at once: sweep over start / end events, per CU (xcc, se, sh, cu) and chip-wide
  auto cu_key = [](const Stamp& s) { return (s.xcc << 16) | (s.hw & 0xff00u); };      // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
  std::vector<std::pair<unsigned long long, int>> all;
  std::map<unsigned, int> per_xcc;
  for (auto& s : h) {
    ev[cu_key(s)].push_back({s.t0, +1});
    ev[cu_key(s)].push_back({s.t1, -1});
    all.push_back({s.t0, +1});
    all.push_back({s.t1, -1});
    per_xcc[s.xcc]++;
  }
  auto peak = [](std::vector<std::pair<unsigned long long, int>>& v) {
    std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first < b.first : a.second < b.second; });
    int cur = 0, best = 0;
    for (auto& e : v) { cur += e.second; best = std::max(best, cur); }
    return best;
  };
  int peak_cu = 0;
  for (auto& kv : ev) peak_cu = std::max(peak_cu, peak(kv.second));
  const int peak_chip = peak(all);
  printf("nwg %5d x %2d waves, LDS %6d B, hold %5.1f us, %s: occupancy API %d per CU | span %7.1f us | started: first 0, median %6.1f, last %6.1f us | "
         "alive at once: %d per CU (max), %d on the chip, %zu CUs used | per XCC:",
         nwg, waves, lds_bytes, hold_us, vg == 2 ? "~100 regs" : "<=64 regs", per_cu, (tmax - tmin) / 100.0, starts[starts.size() / 2], starts.back(),
         peak_cu, peak_chip, ev.size());
  for (auto& kv : per_xcc) printf(" %d", kv.second);
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc >= 5) {
    run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atof(argv[4]), argc > 5 && atoi(argv[5]) > 64 ? 2 : 1);
    return 0;
  }
  // the C5 Sinkhorn shape (520 x 16 waves x 73.9 KB, ~15 us of life), the same with 512, with half the LDS, as 8-wave workgroups;
  // the C3 shape (4160 x 8 waves x 36.9 KB, ~12 us); a zero-hold launch of each (pure dispatch)
  const int c5 = (8 * 2048 + 2049) * 4 + 192, c3 = (8 * 1024 + 1025) * 4 + 96;
  run(520, 16, c5, 15.0, 1);
  run(512, 16, c5, 15.0, 1);
  run(520, 16, c5 / 2, 15.0, 1);
  run(1040, 8, c5 / 2, 7.5, 1);
  run(520, 16, c5, 0.0, 1);
  run(4160, 8, c3, 12.0, 1);
  run(4160, 8, c3, 0.0, 1);
  run(520, 16, 1024, 15.0, 1);
  return 0;
}
