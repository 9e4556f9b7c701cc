// registration.hip — RANSAC partial-affine (4-DoF similarity) fit on matched keypoints, one
// workgroup per image pair.  GPU counterpart of the post-step inside the reference's timed region,
// cv2.estimateAffinePartial2D(mkpts0, mkpts1, method=cv2.RANSAC, ransacReprojThreshold=7)
// (superpoint_glue_test.py:86-92; superpoint_flann_test.py:84-90) — SURVEY §8(f) rank 1.
// OpenCV's arithmetic (its RNG, LMedS/LM refinement) is third-party and absent from the reference
// tree: parity is against oracle/ransac_ref.py, which uses the same counter-based hypothesis
// sequence, so inlier masks are comparable bit for bit; parity vs OpenCV itself is unpinned.
#include "imx_kernels.h"

namespace imx {
namespace {

__device__ __forceinline__ unsigned mix32(unsigned x) {     // murmur3 finaliser
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

// block-wide sum of a double over 256 threads (LDS scratch of 4 doubles)
__device__ double block_sum(double v, double* scratch) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ __launch_bounds__(256) void ransac_similarity_kernel(RansacArgs a) {
  extern __shared__ float sm[];
  // matched source / destination coordinates: in LDS up to 8192 slots (128 KB), in the caller's scratch above that
  float* sx = a.K <= 8192 ? sm : a.scratch + (size_t)blockIdx.x * 4 * a.K;
  float* sy = sx + a.K;
  float* dx = sy + a.K;
  float* dy = dx + a.K;
  __shared__ int s_n, s_best_cnt;
  __shared__ int wcnt[4], wh[4];
  __shared__ float s_model[4];
  __shared__ double dscr[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* k0 = a.kpts0 + (size_t)b * a.K * 2;
  const float* k1 = a.kpts1 + (size_t)b * a.K * 2;
  const long long* m0 = a.matches0 + (size_t)b * a.K;
  const int cnt0 = a.counts0 ? a.counts0[b] : a.K;

  // ---- 1. ordered compaction of the matched pairs (index order, as kpts0[valid] in the reference)
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int i0 = 0; i0 < a.K; i0 += 256) {
    const int i = i0 + tid;
    const long long j = (i < cnt0) ? m0[i] : -1;
    const bool v = j >= 0;
    const unsigned long long bal = __ballot(v);
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_n;
    for (int w = 0; w < wave; ++w) off += wcnt[w];
    if (v) {
      const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
      sx[pos] = k0[2 * i]; sy[pos] = k0[2 * i + 1];
      dx[pos] = k1[2 * j]; dy[pos] = k1[2 * j + 1];
    }
    __syncthreads();
    if (tid == 0) s_n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int n = s_n;
  float* M = a.M + (size_t)b * 6;
  if (n <= 3) {                 // the reference only fits when len(mkpts0) > 3 (:86)
    if (tid < 6) M[tid] = 0.f;
    if (tid == 0) a.n_inliers[b] = 0;
    for (int i = tid; i < a.K; i += 256) a.inlier[(size_t)b * a.K + i] = 0;
    return;
  }
  const float thr2 = a.threshold * a.threshold;

  // ---- 2. hypotheses: similarity through two matches, inlier count over all matches
  int best_cnt = -1, best_h = 0x7fffffff;
  for (int h = tid; h < a.hypotheses; h += 256) {
    const unsigned r = mix32(a.seed ^ (0x9E3779B9u * (unsigned)(b + 1)) ^ (0x85EBCA6Bu * (unsigned)(h + 1)));
    const int i = (int)(r % (unsigned)n);
    int j = (int)(mix32(r + 0x27D4EB2Fu) % (unsigned)(n - 1));
    if (j >= i) ++j;
    const float px = sx[j] - sx[i], py = sy[j] - sy[i];
    const float qx = dx[j] - dx[i], qy = dy[j] - dy[i];
    const float den = px * px + py * py;
    int cnt = -1;
    if (den > 1e-12f) {
      const float ca = (qx * px + qy * py) / den, cb = (qy * px - qx * py) / den;   // (q2-q1)/(p2-p1) as complex
      const float tx = dx[i] - (ca * sx[i] - cb * sy[i]), ty = dy[i] - (cb * sx[i] + ca * sy[i]);
      cnt = 0;
      for (int k = 0; k < n; ++k) {
        const float ex = ca * sx[k] - cb * sy[k] + tx - dx[k], ey = cb * sx[k] + ca * sy[k] + ty - dy[k];
        cnt += (ex * ex + ey * ey < thr2) ? 1 : 0;
      }
    }
    if (cnt > best_cnt) { best_cnt = cnt; best_h = h; }      // ascending h per thread: first best kept
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int oc = __shfl_xor(best_cnt, o), oh = __shfl_xor(best_h, o);
    if (oc > best_cnt || (oc == best_cnt && oh < best_h)) { best_cnt = oc; best_h = oh; }
  }
  if (lane == 0) { wcnt[wave] = best_cnt; wh[wave] = best_h; }
  __syncthreads();
  if (tid == 0) {
    int bc = wcnt[0], bh = wh[0];
    for (int w = 1; w < 4; ++w)
      if (wcnt[w] > bc || (wcnt[w] == bc && wh[w] < bh)) { bc = wcnt[w]; bh = wh[w]; }
    s_best_cnt = bc;
    // recompute the winning model
    const unsigned r = mix32(a.seed ^ (0x9E3779B9u * (unsigned)(b + 1)) ^ (0x85EBCA6Bu * (unsigned)(bh + 1)));
    const int i = (int)(r % (unsigned)n);
    int j = (int)(mix32(r + 0x27D4EB2Fu) % (unsigned)(n - 1));
    if (j >= i) ++j;
    const float px = sx[j] - sx[i], py = sy[j] - sy[i], qx = dx[j] - dx[i], qy = dy[j] - dy[i];
    const float den = px * px + py * py;
    const float ca = (qx * px + qy * py) / den, cb = (qy * px - qx * py) / den;
    s_model[0] = ca; s_model[1] = cb;
    s_model[2] = dx[i] - (ca * sx[i] - cb * sy[i]);
    s_model[3] = dy[i] - (cb * sx[i] + ca * sy[i]);
  }
  __syncthreads();
  if (s_best_cnt < 2) {
    if (tid < 6) M[tid] = 0.f;
    if (tid == 0) a.n_inliers[b] = 0;
    for (int i = tid; i < a.K; i += 256) a.inlier[(size_t)b * a.K + i] = 0;
    return;
  }
  // ---- 3. least-squares similarity on the inliers of the best hypothesis (double accumulation)
  const float ca = s_model[0], cb = s_model[1], tx = s_model[2], ty = s_model[3];
  double c = 0, spx = 0, spy = 0, sqx = 0, sqy = 0;
  for (int k = tid; k < n; k += 256) {
    const float ex = ca * sx[k] - cb * sy[k] + tx - dx[k], ey = cb * sx[k] + ca * sy[k] + ty - dy[k];
    if (ex * ex + ey * ey < thr2) { c += 1; spx += sx[k]; spy += sy[k]; sqx += dx[k]; sqy += dy[k]; }
  }
  c = block_sum(c, dscr); spx = block_sum(spx, dscr); spy = block_sum(spy, dscr);
  sqx = block_sum(sqx, dscr); sqy = block_sum(sqy, dscr);
  const double mpx = spx / c, mpy = spy / c, mqx = sqx / c, mqy = sqy / c;
  double den = 0, dot = 0, crs = 0;
  for (int k = tid; k < n; k += 256) {
    const float ex = ca * sx[k] - cb * sy[k] + tx - dx[k], ey = cb * sx[k] + ca * sy[k] + ty - dy[k];
    if (ex * ex + ey * ey < thr2) {
      const double ux = sx[k] - mpx, uy = sy[k] - mpy, vx = dx[k] - mqx, vy = dy[k] - mqy;
      den += ux * ux + uy * uy; dot += ux * vx + uy * vy; crs += ux * vy - uy * vx;
    }
  }
  den = block_sum(den, dscr); dot = block_sum(dot, dscr); crs = block_sum(crs, dscr);
  if (tid == 0) {
    const double fa = den > 1e-12 ? dot / den : ca, fb = den > 1e-12 ? crs / den : cb;
    M[0] = (float)fa; M[1] = (float)(-fb); M[2] = (float)(mqx - (fa * mpx - fb * mpy));
    M[3] = (float)fb; M[4] = (float)fa;    M[5] = (float)(mqy - (fb * mpx + fa * mpy));
    a.n_inliers[b] = (int)c;
  }
  // ---- 4. inlier mask in keypoint0 index space (mask of the RANSAC model, as OpenCV reports it)
  for (int i = tid; i < a.K; i += 256) {
    const long long j = (i < cnt0) ? m0[i] : -1;
    unsigned char v = 0;
    if (j >= 0) {
      const float x = k0[2 * i], y = k0[2 * i + 1];
      const float ex = ca * x - cb * y + tx - k1[2 * j], ey = cb * x + ca * y + ty - k1[2 * j + 1];
      v = (ex * ex + ey * ey < thr2) ? 1 : 0;
    }
    a.inlier[(size_t)b * a.K + i] = v;
  }
}

// ---- exact 2-nearest-neighbour search + ratio test on descriptor rows: GPU replacement of
//      cv2.FlannBasedMatcher(KDTree).knnMatch(Desc1, Desc2, k=2) + `m.distance < 0.7*n.distance`
//      (superpoint_flann_test.py:66-74).  FLANN is approximate: the exact search is
//      its ideal; parity vs FLANN is unpinned, vs the brute-force oracle exact.
//      dots (B,N0p,N1p) = D0 . D1^T from score_mfma; dist^2(i,j) = |a_i|^2 + |b_j|^2 - 2 dot.
__global__ __launch_bounds__(256) void rownorm2_kernel(const float* __restrict__ x, int d, long rows, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) { const float v = x[r * d + c]; s += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[r] = s;
}

__global__ __launch_bounds__(256) void knn2_kernel(KnnArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (i >= a.N0) return;
  const int m = a.n0 ? a.n0[b] : a.N0, n = a.n1 ? a.n1[b] : a.N1;
  const size_t o = (size_t)b * a.N0 + i;
  if (i >= m || n < 2) {            // knnMatch needs two neighbours; padded rows stay unmatched
    if (lane == 0) { a.matches[o] = -1; a.dist1[o] = 0.f; a.dist2[o] = 0.f; }
    return;
  }
  const float* dots = a.dots + ((size_t)b * a.N0p + i) * a.N1p;
  const float* nb2 = a.norm1 + (size_t)b * a.N1p;
  float k1 = INFINITY, k2 = INFINITY;
  int j1 = 0x7fffffff;
  for (int j = lane; j < n; j += 64) {
    const float k = nb2[j] - 2.0f * dots[j];
    if (k < k1) { k2 = k1; k1 = k; j1 = j; } else if (k < k2) { k2 = k; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ok1 = __shfl_xor(k1, off), ok2 = __shfl_xor(k2, off);
    const int oj1 = __shfl_xor(j1, off);
    if (ok1 < k1 || (ok1 == k1 && oj1 < j1)) { k2 = fminf(k1, ok2); k1 = ok1; j1 = oj1; }
    else { k2 = fminf(k2, ok1); }
  }
  if (lane == 0) {
    const float na2 = a.norm0[(size_t)b * a.N0p + i];
    const float d1 = sqrtf(fmaxf(na2 + k1, 0.f)), d2 = sqrtf(fmaxf(na2 + k2, 0.f));
    a.matches[o] = (d1 < a.ratio * d2) ? (long long)j1 : -1;
    a.dist1[o] = d1;
    a.dist2[o] = d2;
  }
}

}  // namespace

hipError_t launch_rownorm2(const float* x, int d, long rows, float* out, hipStream_t s) {
  hipLaunchKernelGGL(rownorm2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, d, rows, out);
  return hipGetLastError();
}

hipError_t launch_knn2(const KnnArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(knn2_kernel, dim3((unsigned)((a.N0 + 3) / 4), (unsigned)a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_ransac(const RansacArgs& a, hipStream_t s) {
  if (a.K <= 0 || a.B <= 0 || a.hypotheses <= 0) return hipErrorInvalidValue;
  if (a.K > 8192 && !a.scratch) return hipErrorInvalidValue;
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(ransac_similarity_kernel), 128 * 1024, attr);
  hipLaunchKernelGGL(ransac_similarity_kernel, dim3((unsigned)a.B), dim3(256), a.K <= 8192 ? (size_t)a.K * 4 * sizeof(float) : 0, s, a);
  return hipGetLastError();
}

}  // namespace imx
