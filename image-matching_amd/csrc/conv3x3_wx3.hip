// conv3x3_wx3.hip — 3x3 / pad 1 convolution + folded BatchNorm + ReLU (+ MaxPool2d(2)) of superpoint/models/unet_parts.py:10-48
// and superpoint_test.py:113-123 as Winograd F(2x4, 3x3) with every fp32 product of the 24 per-position GEMMs carried by the
// bf16 matrix pipe as SIX bf16 term products (the scheme of gemm_x3.hip / attention_x3.hip: x = h + m + l exactly, one rounding per
// sixteen products; against float64 more accurate than the fp32 MFMA chain, tools/ubench/mfma_bf16x3.hip).  Round 3: the fp32-MFMA
// form of the same arithmetic (conv3x3_wino24.hip) sits at 0.58-0.66 of a pipe that is 16x slower than the bf16 one.
//
//   Y = A2^T [ (G2 g G4^T) (.) (B2^T d B4) ] A4   per 2 x 4 output tile ("wtile") and 4 x 6 input patch d;
//   M_p[co][wtile] = sum_ci U_p[co][ci] V_p[ci][wtile]  for the 24 positions p = (i, j), on v_mfma_f32_32x32x16_bf16.
//
// Why the two earlier bf16-pipe attempts lost and what is different here.  On the bf16 pipe an fp32 product costs 3/8 of
// the fp32 MFMA's cycles, so everything AROUND the MFMAs has to shrink by the same factor: operand bytes per MFMA (each value is
// three bf16 planes) and the VALU work of producing V (input transform + the split: ~8.5 instructions per V value, and only ~5
// instructions issue beneath one 32-cycle MFMA).  The decomposition:
//   * a workgroup = 4 waves, ONE PER SIMD (up to 512 registers each), works on 32 wtiles x 64 output channels; wave w owns the
//     Winograd ROW i = w: six positions (w, j), 64 co x 32 wtiles each = 12 accumulator blocks of 32 x 32 (192 registers);
//   * V never touches LDS: row i of B2^T d is a +-1 combination of two raw rows, so the wave reads the two rows of its wtiles' raw
//     patch (fp32, staged per 16-channel chunk), applies the row combination and B4 along the columns and splits the six values
//     per channel into bf16 planes IN REGISTERS, directly in the MFMA's B-operand layout (lane = (wtile, k half)).  Each V value
//     feeds 64 output channels: ~5.7 VALU instructions per MFMA -- the kernel is VALU-issue bound by design, at ~0.7 of the bf16 pipe,
//     i.e. ~1.9x the fp32 MFMA PEAK in fp32-equivalent terms;
//   * U (weights, split on the host, MFMA A-fragment order) streams from L2 straight into registers, 36 KB per wave and chunk,
//     three j-steps ahead;
//   * the output transform is A4 along j in registers (6 -> 4 values), then the four waves' rows are combined (A2: 4 -> 2) through a
//     68 KB LDS exchange, and every thread finishes a share of the outputs: bias, ReLU, 2x2 max-pool, 16-byte stores.
// Persistent: one workgroup per CU walks (tile, 64-channel output block) items; the 16-channel chunks of all its items form ONE
// stream (raw patch of chunk s+3 in flight from HBM/L2, chunk s+2 going to LDS, chunk s+1 being transformed, chunk s multiplied).
#include "imx_kernels.h"
#include <cstdio>
#include <cstdlib>

namespace imx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int CK = 16, NT = 64;                 // input channels per chunk, output channels per item
constexpr unsigned OOB = 0x7ffffff0u;           // byte offset beyond any image: buffer loads return 0 (= the zero padding)
constexpr int UJ = 2 * 3 * 512;                 // bf16 elements of U per (chunk, row i, column j): [cob][plane][lane][8]
constexpr int UCH = 4 * 6 * UJ;                 // per (64 co, 16 ci) chunk: [i][j][cob][plane][lane][8]

template <int TYW, int TXW>
struct Geo {
  static_assert(TYW * TXW == 32, "32 wtiles per workgroup (the MFMA's 32 columns)");
  static constexpr int OH = 2 * TYW, OW = 4 * TXW, RH = OH + 2, RW = OW + 2;   // output pixels / raw patch (pad-1 halo) per item
  static constexpr int RWP = RW + 1;            // row pitch of the staged patch in 16-byte pixels
  static constexpr int QSZ = RH * RWP * 4;      // floats per channel-quad plane
  static constexpr int RAWC = 4 * QSZ;          // floats per 16-channel chunk: [quad q][row][pixel][4]
  static constexpr int NE = 4 * RH * RW;        // 16-byte entries per chunk
  static constexpr int LPT = (NE + 255) / 256;  // entries per thread
};
constexpr int XTS = 33;                          // exchange buffer: 16-byte slots per (row i, wtile): [x (4)][co quad (8)] + 1 pad
constexpr int XSZ = 4 * 32 * XTS * 4;            // floats

template <bool V>
struct BoolC { static constexpr bool value = V; };
struct Item { int b, y0, x0, cog; };
// position in a workgroup's stream of chunks.  A workgroup's items are `grid` apart, so (tile, cog) advance incrementally; the
// divisions that turn a tile into (image, y0, x0) run once per item, in the item-change paths only.
struct Cursor { int item, chunk, tile, cog; };

// An accumulator element, AGPR -> VGPR, pinned where it is written: left to the compiler, the copies of all 192 accumulators are
// placed in the MFMA block that precedes the (conditional) output transform and executed every chunk.
__device__ __forceinline__ float acc_read(float a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

// x = h + m + l, two values at a time (the conversions round to nearest even and pack)
__device__ __forceinline__ void split2(float x0, float x1, bf16x2& h, bf16x2& m, bf16x2& l) {
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  l[0] = (__bf16)(r0 - (float)m[0]); l[1] = (__bf16)(r1 - (float)m[1]);
}

// EXP (trace builds only, IMX_WX3_EXP=n): timing experiments that DROP one ingredient of the phase (results are garbage):
//   1 no U-fragment loads in the stream   2 no input transform / split   3 no raw-patch loads / stores
template <int TYW, int TXW, bool POOL, bool RELU, bool TRACE, int EXP = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_wx3(ConvArgs p, const __bf16* __restrict__ ux, int tiles_x, int tiles_y, int nitems, unsigned* trace) {
  using G = Geo<TYW, TXW>;
  // TRACE: s_memtime deltas summed over the stream (developer instrumentation, IMX_WINO_TRACE=1): barrier, first half, second half,
  // tail (loader), epilogue
  unsigned tph[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
  int nitem_done = 0;
#define IMX_TS(i_)                                                   \
  if constexpr (TRACE) {                                             \
    const unsigned long long now = __builtin_readcyclecounter();     \
    tph[i_] += (unsigned)(now - tprev);                              \
    tprev = now;                                                     \
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw0 = smem;                  // [RAWC]  even chunks of the stream
  float* raw1 = smem + G::RAWC;        // [RAWC]  odd chunks
  float* X = smem + 2 * G::RAWC;       // [XSZ]   output-transform exchange
  // per-thread loader tables [k][tid]: LDS store offsets (static) and global byte offsets of the current loader item.  They are
  // read back with ds_read right before use: kept in registers the compiler spills them to scratch, and a scratch reload waits
  // with vmcnt(0) -- i.e. for every U fragment in flight (measured: 1-2.4 k cycles per chunk at the end of the phase)
  int* tab_dst = reinterpret_cast<int*>(X + XSZ);            // [LPT][256]
  unsigned* tab_src = reinterpret_cast<unsigned*>(tab_dst + G::LPT * 256);   // [LPT][256]
  int* tab_pos = reinterpret_cast<int*>(tab_src + G::LPT * 256);             // [LPT][256]  (row << 16 | column) inside the patch
  int* tab_blk = tab_pos + G::LPT * 256;                                     // [LPT][256]  byte offset of the entry's channel block / half
  float* bias_s = reinterpret_cast<float*>(tab_blk + G::LPT * 256);          // [Cout] (<= 512)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = Winograd row i of this wave
  const int n = lane & 31, kg = lane >> 5, ty = n / TXW, tx = n % TXW;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CK, ncog = Cout / NT;
  const int grid = (int)gridDim.x;
  // XCD-aware start index: workgroups are dispatched round-robin over the 8 XCDs; consecutive items (the output blocks of one tile,
  // neighbouring tiles) land on ONE XCD in adjacent dispatch slots, so a tile's input patch is fetched from HBM once
  const int vb = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (vb >= nitems) return;
  const int nmine = (nitems - vb + grid - 1) / grid, Q = nmine * nchunk;

  const int gq = grid / ncog, gstep = grid % ncog;      // an item step of `grid` = gq tiles and gstep output blocks
  auto origin = [&](const Cursor& c) -> Item {          // (rare paths only: two integer divisions)
    Item r;
    r.cog = c.cog;
    r.x0 = (c.tile % tiles_x) * G::OW;
    r.y0 = ((c.tile / tiles_x) % tiles_y) * G::OH;
    r.b = c.tile / (tiles_x * tiles_y);
    return r;
  };
  auto advance = [&](Cursor& c) {          // block-uniform; past the end the cursor stays on the last chunk (harmless re-fetch)
    if (c.chunk + 1 < nchunk) { ++c.chunk; }
    else if (c.item + grid < nitems) {
      c.item += grid; c.chunk = 0; c.tile += gq; c.cog += gstep;
      if (c.cog >= ncog) { c.cog -= ncog; ++c.tile; }
    }
  };

  // ---- rows of B2^T: i0 = d0 - d2, i1 = d1 + d2, i2 = d2 - d1, i3 = d1 - d3  ->  o = d[ra] + sg * d[rb]
  const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1, rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const float sg = wave == 1 ? 1.f : -1.f;
  // float offsets of this lane's (wtile, k half) inside a staged chunk: channel quads 2 kg, 2 kg + 1; patch rows 2 ty + ra / rb
  const int offA = ((2 * kg * G::RH + 2 * ty + ra) * G::RWP + 4 * tx) * 4;
  const int offB = ((2 * kg * G::RH + 2 * ty + rb) * G::RWP + 4 * tx) * 4;

  // ---- loader: thread -> LPT 16-byte entries of the chunk's patch.  Entry e -> channel block blk (8 channels) = e / (2 NPIX),
  //      pixel = (e % (2 NPIX)) >> 1, half = e & 1: lane pairs read a pixel's 32 contiguous bytes of the channel-blocked input
  //      (B, C/8, H, W, 8), consecutive lanes consecutive pixels of a patch row (dense lines)
  constexpr int NPIX = G::RH * G::RW;
  f32x4 Gr[G::LPT];
#pragma unroll
  for (int k = 0; k < G::LPT; ++k) {
    const int e = (tid + 256 * k < G::NE) ? tid + 256 * k : tid;      // the tail repeats the thread's first entry (same source, same destination)
    const int blk = e / (2 * NPIX), rem = e % (2 * NPIX), pix = rem >> 1, half = rem & 1;
    const int py = pix / G::RW, px = pix % G::RW;
    tab_dst[k * 256 + tid] = (((2 * blk + half) * G::RH + py) * G::RWP + px) * 4;
    tab_pos[k * 256 + tid] = (py << 16) | px;                        // where in the patch, and
    tab_blk[k * 256 + tid] = blk * H * W * 32 + half * 16;           // which channel block / half (byte offset inside the image's chunk)
  }
  for (int c = tid; c < Cout; c += 256) bias_s[c] = p.bias[c];
  __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
  const int img_bytes = H * W * Cin * 4;
  auto bind_item = [&](const Item& it) {       // per-image descriptor + this thread's byte offsets inside the image (chunk 0)
    const unsigned long long base = (unsigned long long)(p.in + (size_t)it.b * H * W * Cin);
    const unsigned long long base_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)base);
    lrs = __builtin_amdgcn_make_buffer_rsrc((void*)base_u, 0, img_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < G::LPT; ++k) {
      const int pos = tab_pos[k * 256 + tid];
      const int gy = it.y0 - 1 + (pos >> 16), gx = it.x0 - 1 + (pos & 0xffff);
      tab_src[k * 256 + tid] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)(tab_blk[k * 256 + tid] + (gy * W + gx) * 32) : OOB;
    }
  };
  auto gload = [&](const Cursor& c) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane(c.chunk * 2 * H * W * 32);      // two channel blocks per chunk
#pragma unroll
    for (int k = 0; k < G::LPT; ++k)
      Gr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)tab_src[k * 256 + tid], so, 0));
  };
  auto lstore = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < G::LPT; ++k) *reinterpret_cast<f32x4*>(dst + tab_dst[k * 256 + tid]) = Gr[k];
  };

  // ---- input transform + split, skewed by half a phase so that ONE set of B operands is live (two sets + the ring of U fragments
  //      + 192 accumulators do not fit 512 registers): while the MFMAs of positions j = 3..5 of chunk s run, chunk s+1 is transformed --
  //      v[0..2] are split straight into B[0..2] (their MFMAs of chunk s were issued in the first half), v[3..5] are kept as fp32
  //      (`vh`, 24 registers) and split into B[3..5] during the first half of phase s+1, beneath the MFMAs of j = 0..2.
  //      B[j][plane] = eight channels (k = 8 kg + e) of V_(wave, j) for the lane's wtile.
  auto split_into = [&](const f32x4& v, bf16x8 (&Bj)[3], int qq) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
      bf16x2 h, m, l;
      split2(v[c], v[c + 1], h, m, l);
      const int e = qq * 4 + c;
      Bj[0][e] = h[0]; Bj[0][e + 1] = h[1];
      Bj[1][e] = m[0]; Bj[1][e + 1] = m[1];
      Bj[2][e] = l[0]; Bj[2][e + 1] = l[1];
    }
  };
  auto xload = [&](const float* rbuf, int qq, f32x4 (&da)[6], f32x4 (&db)[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      da[c] = *reinterpret_cast<const f32x4*>(rbuf + offA + qq * G::QSZ + c * 4);
      db[c] = *reinterpret_cast<const f32x4*>(rbuf + offB + qq * G::QSZ + c * 4);
    }
  };
  auto xcalc = [&](int qq, const f32x4 (&da)[6], const f32x4 (&db)[6], bf16x8 (&B)[6][3], f32x4 (&vh)[3][2]) __attribute__((always_inline)) {
    const f32x4 sg4 = {sg, sg, sg, sg};
    f32x4 o[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) o[c] = __builtin_elementwise_fma(db[c], sg4, da[c]);
    // B4^T of F(4,3) along the columns (the same operation order as conv3x3_wino24.hip's packed form):
    //   v0 = 4 o0 - 5 o2 + o4      v1 = (o4 - 4 o2) + (o3 - 4 o1)     v2 = (o4 - 4 o2) - (o3 - 4 o1)
    //   v5 = 4 o1 - 5 o3 + o5      v3 = (o4 - o2) + 2 (o3 - o1)       v4 = (o4 - o2) - 2 (o3 - o1)
    const f32x4 k4 = {4.f, 4.f, 4.f, 4.f}, k5 = {5.f, 5.f, 5.f, 5.f}, k2 = {2.f, 2.f, 2.f, 2.f};
    const f32x4 e42 = __builtin_elementwise_fma(-k4, o[2], o[4]), e31 = __builtin_elementwise_fma(-k4, o[1], o[3]);
    const f32x4 f42 = o[4] - o[2], f31 = o[3] - o[1];
    split_into(__builtin_elementwise_fma(k4, o[0], __builtin_elementwise_fma(-k5, o[2], o[4])), B[0], qq);
    split_into(e42 + e31, B[1], qq);
    split_into(e42 - e31, B[2], qq);
    vh[0][qq] = __builtin_elementwise_fma(k2, f31, f42);
    vh[1][qq] = __builtin_elementwise_fma(-k2, f31, f42);
    vh[2][qq] = __builtin_elementwise_fma(k4, o[1], __builtin_elementwise_fma(-k5, o[3], o[5]));
  };
  // The values computed for the NEXT phase are used by no instruction of this basic block: without this the compiler sinks their
  // whole computation out of the MFMA block into the loop latch, where nothing overlaps it (measured in the ISA: 211 VALU
  // instructions per phase in a block without a single MFMA).
  auto pin = [&](bf16x8 (&B)[6][3], f32x4 (&vh)[3][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(B[j][pl]));
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) asm volatile("" : "+v"(vh[j][qq]));
    }
  };
  auto split_held = [&](const f32x4 (&vh)[3][2], bf16x8 (&B)[6][3]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) split_into(vh[j][qq], B[3 + j], qq);
  };

  // ---- U fragments: ring of TWO j-steps (cob x plane = six 1 KB loads each; a third does not fit the register file next to the
  //      B operands and a raw chunk in flight), straight from L2 into registers
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)ux, 0, ncog * nchunk * UCH * 2, 0x00020000);
  constexpr int NA = (EXP >= 4) ? 3 : 2;       // EXP 4 / 5: a ring of three j-steps (5: and no raw-patch loads)
  bf16x8 Ar[NA][2][3];
  auto ubase = [&](const Cursor& c) -> int {                // byte offset of (cog, chunk, row i = wave)
    return __builtin_amdgcn_readfirstlane((((c.cog * nchunk + c.chunk) * 4 + wave) * 6 * UJ) * 2);
  };
  auto aload = [&](int slot, int ub, int j) __attribute__((always_inline)) {
#pragma unroll
    for (int cob = 0; cob < 2; ++cob)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        Ar[slot][cob][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane * 16, ub + ((j * 2 + cob) * 3 + pl) * 1024, 0));
  };

  // The accumulators live in AGPRs for the whole stream: an item's first chunk starts every accumulator from a LITERAL zero (the
  // `first` variants of the phase) instead of clearing registers -- any VALU write to them makes the register allocator keep them in
  // VGPRs across the loop and copy all 192 into AGPRs and back around every block of MFMAs (measured in the ISA: 96 v_accvgpr_write
  // per half phase).
  f32x16 acc[6][2];

  // ---- output transform + exchange + stores of one finished item
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int out_img_bytes = Ho * Wo * Cout * 4;
  auto epilogue = [&](const Item& it) __attribute__((always_inline)) {
    const unsigned long long ob = (unsigned long long)(p.out + (size_t)it.b * Ho * Wo * Cout);
    const unsigned long long ob_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(ob >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ob);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)ob_u, 0, out_img_bytes, 0x00020000);
#pragma unroll
    for (int cob = 0; cob < 2; ++cob) {
      // A4 along j (in registers), four accumulator registers (= four consecutive channels) at a time:
      //   t0 = m0 + (m1+m2) + (m3+m4), t1 = (m1-m2) + 2 (m3-m4), t2 = (m1+m2) + 4 (m3+m4), t3 = (m1-m2) + 8 (m3-m4) + m5
      // lane (wtile n, k half kg) holds channels 8 g + 4 kg + (r & 3), g = r >> 2: one 16-byte slot per (x, g)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 m[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) m[j][e] = acc_read(acc[j][cob][4 * g + e]);
        const f32x4 a12 = m[1] + m[2], b12 = m[1] - m[2], c34 = m[3] + m[4], d34 = m[3] - m[4];
        const f32x4 t[4] = {(m[0] + a12) + c34, b12 + 2.f * d34, a12 + 4.f * c34, (b12 + 8.f * d34) + m[5]};
#pragma unroll
        for (int x = 0; x < 4; ++x) *reinterpret_cast<f32x4*>(X + ((wave * 32 + n) * XTS + x * 8 + 2 * g + kg) * 4) = t[x];
        __builtin_amdgcn_sched_barrier(0);      // one slice at a time: the stream is short of registers here (B, the U ring and a raw chunk stay live)
      }
      __syncthreads();
      const int cbase = it.cog * NT + cob * 32;
      // A2 across the four waves' rows: y0 = t(0) + t(1) + t(2), y1 = t(1) - t(2) - t(3); a thread finishes (wtile, channel quad, x)
      auto rows = [&](int tile, int x, int co4, f32x4& y0, f32x4& y1) __attribute__((always_inline)) {
        const float* xb = X + (tile * XTS + x * 8 + co4) * 4;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(xb), t1 = *reinterpret_cast<const f32x4*>(xb + 32 * XTS * 4);
        const f32x4 t2 = *reinterpret_cast<const f32x4*>(xb + 2 * 32 * XTS * 4), t3 = *reinterpret_cast<const f32x4*>(xb + 3 * 32 * XTS * 4);
        y0 = (t0 + t1) + t2;
        y1 = (t1 - t2) - t3;
      };
      // stores through a per-image buffer descriptor: 32-bit offsets, pixels outside the map get an out-of-range offset (dropped)
      auto store = [&](int oy, int ox, int c0, f32x4 v) __attribute__((always_inline)) {
        v += *reinterpret_cast<const f32x4*>(bias_s + c0);
        if (RELU) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
        const int off = p.out_blocked ? ((((c0 >> 3) * Ho + oy) * Wo + ox) * 8 + (c0 & 7)) * 4 : ((oy * Wo + ox) * Cout + c0) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), ors,
                                               (oy < Ho && ox < Wo) ? off : (int)OOB, 0, 0);
      };
      if (!POOL) {
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
          const int u = tid + 256 * k, co4 = u & 7, x = (u >> 3) & 3, tile = u >> 5;
          f32x4 y0, y1;
          rows(tile, x, co4, y0, y1);
          const int oy = it.y0 + 2 * (tile / TXW), ox = it.x0 + 4 * (tile % TXW) + x;
          store(oy, ox, cbase + co4 * 4, y0);
          store(oy + 1, ox, cbase + co4 * 4, y1);
        }
      } else {
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
          const int u = tid + 256 * k, co4 = u & 7, xp = (u >> 3) & 1, tile = u >> 4;
          f32x4 a0, a1, b0, b1;
          rows(tile, 2 * xp, co4, a0, a1);
          rows(tile, 2 * xp + 1, co4, b0, b1);
          // relu(max(.) + bias) == max(relu(. + bias)): the bias is per channel and ReLU is monotone (unet_parts.py:44-48)
          const f32x4 mx = __builtin_elementwise_max(__builtin_elementwise_max(a0, a1), __builtin_elementwise_max(b0, b1));
          store((it.y0 >> 1) + tile / TXW, (it.x0 >> 1) + 2 * (tile % TXW) + xp, cbase + co4 * 4, mx);
        }
      }
      __syncthreads();
    }
  };

  // ---- the stream
  Cursor cc{vb, 0, vb / ncog, vb % ncog}, lc = cc;
  Item ci = origin(cc), li = ci;
  bind_item(li);
  auto lnext = [&]() {                          // loader cursor one chunk on; a new item re-binds the descriptor and offsets
    const int was = lc.item;
    advance(lc);
    if (lc.item != was) { li = origin(lc); bind_item(li); }      // block-uniform, once per item
  };
  gload(lc); lstore(raw0);                      // chunk 0
  lnext(); gload(lc); lstore(raw1);             // chunk 1
  lnext(); gload(lc);                           // chunk 2 stays in registers
  int ub = ubase(cc);
  aload(0, ub, 0); aload(1, ub, 1);
  if (NA == 3) aload(2, ub, 2);
  bf16x8 B[6][3];
  f32x4 vh[3][2];
  __syncthreads();
  {
    f32x4 da[6], db[6];
    xload(raw0, 0, da, db);
    xcalc(0, da, db, B, vh);
    xload(raw0, 1, da, db);
    xcalc(1, da, db, B, vh);
  }

  // six term products, smallest first: planes (U, V) = (m,m) (h,l) (l,h) (h,m) (m,h) (h,h); the two output-channel blocks interleave
  auto mfma3 = [&](int j0, int ubc, int ubn, auto first) __attribute__((always_inline)) {
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int j = j0 + jj;
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int cob = 0; cob < 2; ++cob)
          acc[j][cob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ar[j % NA][cob][PA[t]], B[j][PB[t]],
                                                                (decltype(first)::value && t == 0) ? zero16 : acc[j][cob], 0, 0, 0);
      if (EXP != 1) aload(j % NA, j + NA < 6 ? ubc : ubn, (j + NA) % 6);      // refill the slot just consumed, NA j-steps ahead
    }
  };
  auto phase = [&](float* rstore, const float* rnext, auto first) __attribute__((always_inline)) {
    if constexpr (TRACE) tprev = __builtin_readcyclecounter();
    __syncthreads();                            // raw of chunk s+1 is complete; every wave is done with the buffer chunk s lived in
    IMX_TS(0)
    Cursor nc = cc;
    advance(nc);
    const int ubn = ubase(nc);
    // ---- ONE basic block from here to the end of the MFMAs (any branch in between splits the scheduling region and the VALU work
    //      stops overlapping the matrix pipe)
    // first half: positions 0..2 of chunk s; beneath them the held v[3..5] of chunk s are split into B[3..5]
    if (EXP != 2) split_held(vh, B);
    mfma3(0, ub, ubn, first);
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU beneath it
    }
    if constexpr (TRACE) { __builtin_amdgcn_sched_barrier(0); IMX_TS(5) }
    if (EXP != 3 && EXP != 5) lstore(rstore);               // chunk s+2 (requested at the end of the previous phase) -> the buffer chunk s lived in
    __builtin_amdgcn_sched_group_barrier(0x200, G::LPT, 0);   // ... AFTER the first half's MFMAs: the loads need the time
    __builtin_amdgcn_sched_barrier(0);          // the raw-chunk registers die here: the transform below needs the room
    IMX_TS(1)
    // second half: positions 3..5 of chunk s; beneath them chunk s+1 is transformed (B[0..2] are free: their MFMAs are issued)
    f32x4 da[6], db[6];
    if (EXP != 2) {
      xload(rnext, 0, da, db);
      xcalc(0, da, db, B, vh);
      xload(rnext, 1, da, db);
      xcalc(1, da, db, B, vh);
    }
    mfma3(3, ub, ubn, first);
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      if (g < 6 || (g >= 14 && g < 20)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // a quad's twelve LDS reads, two per MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // eight VALU
    }
    pin(B, vh);
    __builtin_amdgcn_sched_barrier(0);
    IMX_TS(2)
    ub = ubn;
    lnext();                                    // (branches from here on: a new item re-binds the loader)
    if (EXP != 3 && EXP != 5) gload(lc);                    // chunk s+3: in flight across the barrier and the next first half
    IMX_TS(3)
    if (cc.chunk == nchunk - 1) {               // block-uniform
      epilogue(ci);
      ++nitem_done;
      IMX_TS(4)
    }
    const int was = cc.item;
    advance(cc);
    if (cc.item != was) ci = origin(cc);        // block-uniform, once per item
  };
  // nchunk is even (Cin % 32 == 0), so an item's first chunk always falls on an even stream position
  for (int s = 0; s < Q; s += 2) {
    if (cc.chunk == 0) phase(raw0, raw1, BoolC<true>{}); else phase(raw0, raw1, BoolC<false>{});      // block-uniform
    phase(raw1, raw0, BoolC<false>{});
  }
  if constexpr (TRACE) {
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned* o = trace + ((size_t)blockIdx.x * 4 + wave) * 8;
      for (int i = 0; i < 5; ++i) o[i] = tph[i];
      o[5] = (unsigned)nitem_done;
      o[6] = tph[5];
    }
  }
#undef IMX_TS
}

template <int TYW, int TXW, bool POOL, bool RELU>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
  using G = Geo<TYW, TXW>;
  const int tiles_x = (a.W + G::OW - 1) / G::OW, tiles_y = (a.H + G::OH - 1) / G::OH;
  const int nitems = tiles_x * tiles_y * a.B * (a.Cout / NT);
  const size_t lds = (size_t)(2 * G::RAWC + XSZ + 4 * G::LPT * 256 + a.Cout) * sizeof(float);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  auto k = conv3x3_wx3<TYW, TXW, POOL, RELU, false>;
  static unsigned long long attr = 0;
  raise_lds_limit(reinterpret_cast<const void*>(k), (int)lds, attr);
  const dim3 grid((unsigned)(nitems < ncu ? nitems : ncu));       // persistent: one workgroup (four 512-register waves) per CU
  last_form = "conv3x3_wx3:bf16x3";
  static const bool trace = getenv("IMX_WINO_TRACE") != nullptr;     // developer instrumentation, read once per process
  if (trace) {
    auto kt = conv3x3_wx3<TYW, TXW, POOL, RELU, true>;
    static const int exp_id = getenv("IMX_WX3_EXP") ? atoi(getenv("IMX_WX3_EXP")) : 0;
    if (exp_id == 1) kt = conv3x3_wx3<TYW, TXW, POOL, RELU, true, 1>;
    if (exp_id == 2) kt = conv3x3_wx3<TYW, TXW, POOL, RELU, true, 2>;
    if (exp_id == 3) kt = conv3x3_wx3<TYW, TXW, POOL, RELU, true, 3>;
    if (exp_id == 4) kt = conv3x3_wx3<TYW, TXW, POOL, RELU, true, 4>;
    if (exp_id == 5) kt = conv3x3_wx3<TYW, TXW, POOL, RELU, true, 5>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static unsigned* dbuf = nullptr;
    constexpr int NREC = 1024 * 4 * 8;
    if (!dbuf) (void)hipMalloc(&dbuf, NREC * sizeof(unsigned));
    (void)hipMemsetAsync(dbuf, 0, NREC * sizeof(unsigned), s);
    hipLaunchKernelGGL(kt, grid, dim3(256), lds, s, a, static_cast<const __bf16*>(a.wux3), tiles_x, tiles_y, nitems, dbuf);
    (void)hipStreamSynchronize(s);
    static unsigned host[NREC];
    (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
    const int n = grid.x < 1024 ? (int)grid.x : 1024;
    double sum[7] = {0}, ni = 0;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < 5; ++j) sum[j] += host[(i * 4 + 1) * 8 + j];
      sum[5] += host[(i * 4 + 1) * 8 + 6];
      ni += host[(i * 4 + 1) * 8 + 5];
    }
    const int nchunk = a.Cin / CK;
    fprintf(stderr, "[wx3 trace] H=%d W=%d Cin=%d Cout=%d pool=%d | per chunk: barrier %.0f  first half %.0f + raw store %.0f  second half %.0f  tail %.0f | epilogue %.0f per item  "
                    "(%.0f items per workgroup; 72 MFMAs = 2304 cycles per chunk)\n", a.H, a.W, a.Cin, a.Cout, (int)POOL, sum[0] / ni / nchunk,
            sum[5] / ni / nchunk, sum[1] / ni / nchunk, sum[2] / ni / nchunk, sum[3] / ni / nchunk, sum[4] / ni, ni / n);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, static_cast<const __bf16*>(a.wux3), tiles_x, tiles_y, nitems, (unsigned*)nullptr);
  return hipGetLastError();
}
}  // namespace

bool conv3x3_wx3_supported(const ConvArgs& a) {
  if (a.first || !a.in_blocked || a.Cin % (2 * CK) || a.Cout % NT || a.Cout > 512 || !a.wux3 || a.H < 2 || a.W < 4) return false;
  // per-image byte offsets are 31-bit, on the input and on the output side
  const size_t Ho = a.pool ? a.H / 2 : a.H, Wo = a.pool ? a.W / 2 : a.W;
  return (size_t)a.H * a.W * a.Cin * 4 < (size_t)OOB && Ho * Wo * (size_t)a.Cout * 4 < (size_t)OOB;
}

hipError_t launch_conv3x3_wx3(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_wx3_supported(a)) return hipErrorInvalidValue;
  const bool wide = a.W >= 128;       // 8 x 32-pixel items; small maps (the 60 x 80 stage) take 16 x 16-pixel items (less ragged-edge waste)
#define IMX_WX3(P_, R_) (wide ? launch_t<4, 8, P_, R_>(a, s) : launch_t<8, 4, P_, R_>(a, s))
  if (a.pool) return a.relu ? IMX_WX3(true, true) : IMX_WX3(true, false);
  return a.relu ? IMX_WX3(false, true) : IMX_WX3(false, false);
#undef IMX_WX3
}

}  // namespace imx
