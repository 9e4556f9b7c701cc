// conv3x3_wino24.hip — 3x3 / pad 1 convolution + folded BatchNorm + ReLU (+ MaxPool2d(2)) of superpoint/models/unet_parts.py:10-48
// and superpoint_test.py:113-123 as Winograd F(2x4, 3x3) on the fp32 matrix cores, for every layer after the fused first one
// (that one is conv1ab_wino24.hip; the arithmetic, the operand-quad layout and the in-stream transform are the same):
//   Y = A2^T [ (G2 g G4^T) (.) (B2^T d B4) ] A4   per 2-row x 4-column output tile ("wtile") and 4x6 input patch d:
//   24 multiplies per 8 outputs (3x fewer than direct, 1.33x fewer than F(2x2,3x3)), summed over input channels as 24
//   independent GEMMs  M_p[wtile][co] = sum_ci V_p[wtile][ci] U_p[ci][co]  on v_mfma_f32_16x16x4_f32.
//
// Persistent: 2 workgroups per CU walk work items (8x16-pixel tile, 64-channel output block), item i -> workgroup i % grid, so
// that at any moment the chip works on neighbouring tiles and every output block of a tile (input patches and the current U
// panels stay in L2).  Workgroup = 4 waves; wave = 16 output channels x the 16 wtiles x 24 positions = 24 accumulators of 4
// VGPRs; with U as the MFMA's A operand the 16x16 D layout gives a lane one wtile and four consecutive channels with all 24
// positions, so the output transform, bias, ReLU and the 2x2 max-pool are in-lane and every result leaves as a 16-byte store.
// The K loop is ONE continuous stream of 8-channel chunks across items, one barrier per chunk.  During the MFMA phase of
// stream position s (48 MFMAs per wave, A operands one ds_read_b128 per quad of four, B operands registers refilled in place
// with position s+1's U panel right behind the MFMAs that consumed them), the same instruction stream carries:
//   * the input transform of position s+1 (raw[(s+1)&1] -> V[(s+1)&1]), split over the four waves by transformed row, as LDS
//     reads + three dense batches of six packed operations + LDS writes;
//   * the raw 10x18x8 patch of position s+2 going from registers to raw[s&1] (it was fetched during phase s-1);
//   * the fetch of position s+3's patch: two buffer_load_dwordx4 per thread through a per-image descriptor; pixels outside
//     the image get an out-of-range offset and come back as the zero padding.
// (Why in-stream: beside a saturated MFMA stream another wave's VALU instruction issues once per ~41 cycles; inside the MFMA
// wave a batch of n costs ~11 + 4.5 n cycles of matrix-pipe time -- tools/ubench/mfma_valu.hip.)
// LDS: V 2 x 13.5 KB, raw 2 x 7.5 KB = 42 KB (no output staging: see the epilogue).
#include "imx_kernels.h"
#include "wino24_pk.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace imx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int OH = 8, OW = 16;                 // output pixels per item (4 x 4 wtiles of 2 x 4)
constexpr int RH = OH + 2, RW = OW + 2;        // input patch (pad-1 halo)
constexpr int RSC = 10;                        // raw chunk: pixel stride (8 channels + 2: wtile columns 4 px apart land 8 banks apart)
constexpr int RAWC = 192 * RSC;                // 180 pixels + pad
// V: [12 quads][4 k rows of 16 slots, 4 slots of padding between rows 1 and 2][4 floats].  The LDS services a ds_read_b128 in four
// 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): a group mixes eight lanes of k row 2m
// with the complementary eight of row 2m+1, so rows 2m and 2m+1 must sit a multiple of 16 slots (one 256-byte bank row)
// apart for the sixteen 16-byte slots of a group to cover all 64 banks.  (Round 1 used a row stride of 18: 2-way conflicts
// on every A-operand read -- SQ_LDS_BANK_CONFLICT was 39 % of SQ_LDS_IDX_ACTIVE.)  The pad between rows 1 and 2 puts the
// transform's ds_write_b64 of k rows {0,1} and {2,3} on different bank halves (2-way instead of 4-way; a wave writes one
// half of every slot, so 2-way is the floor).
constexpr int NQ = 12, QSL = 68;               // slots per quad
__host__ __device__ constexpr int vslot(int k, int n) { return 16 * k + 4 * (k >> 1) + n; }
constexpr int VSZ = NQ * QSL * 4;              // 3264
constexpr int UCH = NQ * 4 * 64 * 4;           // 12288 floats of U per (64 co, 8 ci)
constexpr int CK = 8, NT = 64;
constexpr unsigned OOB = 0x7ffffff0u;          // byte offset beyond any image: buffer loads return 0

struct Item { int b, y0, x0, cob; };
template <bool V>
struct BoolC { static constexpr bool value = V; };

template <bool POOL, bool RELU, bool TRACE>
__global__ __launch_bounds__(256, 2) void conv3x3_wino24(ConvArgs p, int tiles_x, int tiles_y, int nitems, unsigned* trace) {
  // TRACE: s_memtime deltas summed over the stream (bring-up instrumentation, IMX_WINO_TRACE=1)
  unsigned tph[4] = {0, 0, 0, 0};
  unsigned long long tprev = 0;
  int nitem_done = 0;
#define IMX_TS(i_)                                                   \
  if constexpr (TRACE) {                                             \
    const unsigned long long now = __builtin_readcyclecounter();     \
    tph[i_] += (unsigned)(now - tprev);                              \
    tprev = now;                                                     \
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* V = smem;                  // [2][VSZ]
  float* raw = V + 2 * VSZ;         // [2][RAWC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave;
  const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  const int nchunk = Cin / CK, ncob = Cout / NT;
  const int grid = (int)gridDim.x;
  // XCD-aware start index: workgroups are dispatched round-robin over the 8 XCDs; virtual index vb makes consecutive items
  // (the output blocks of one tile, neighbouring tiles) land on ONE XCD in adjacent dispatch slots, so a tile's input patch is
  // fetched from HBM once and re-read from that L2 by its other output blocks
  const int vb = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (vb >= nitems) return;
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.wu24, 0, ncob * nchunk * UCH * 4, 0x00020000);
  const int voff = (cb * 64 + lane) * 16;
  const int img_bytes = H * W * Cin * 4;

  // ---- per-lane constants of the input transform: lane = (channel pair tk, wtile tw); transformed row i = wave:
  //      i0 = d0 - d2, i1 = d1 + d2, i2 = d2 - d1, i3 = d1 - d3  (rows of B2^T)
  const int tk = lane & 3, tw = lane >> 2, twr = tw >> 2, twc = tw & 3;
  const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1, rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const float sg = wave == 1 ? 1.f : -1.f;
  const f32x2 sg2 = {sg, sg};
  const f32x2 m5 = {-5.f, -5.f};
  const float* rpa = raw + ((2 * twr + ra) * RW + 4 * twc) * RSC + 2 * tk;
  const float* rpb = raw + ((2 * twr + rb) * RW + 4 * twc) * RSC + 2 * tk;
  float* vwr = V + vslot(tk, tw) * 4 + (wave >> 1) * QSL * 4 + (wave & 1) * 2;
  int aoff = vslot(lane >> 4, lane & 15) * 4;
  asm volatile("" : "+v"(aoff));               // opaque: the 12 quad reads are immediate offsets from one base
  const float* vrd = V + aoff;

  // ---- loader: thread -> two (pixel, channel half) float4 of the 10x18x8 patch (360 of them; threads >= 104 repeat their
  //      first one: same source, same destination).  Entry e = (half = e / 180, pixel = e % 180): sixteen consecutive lanes
  //      store sixteen consecutive pixels' halves, 10 floats apart = 16 distinct bank pairs of the 32-bank write path (the
  //      pixel-pair-major order of round 1 put two of them on the same pair: 2-way conflicts on every raw store)
  int lpy[2], lpx[2], ldst[2], lhalf[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = (k == 1 && tid + 256 < RH * RW * 2) ? tid + 256 : tid;
    // blocked input: lane pairs read a pixel's 32 contiguous bytes, consecutive lanes consecutive pixels (dense lines; the LDS
    // stores are then 2-way conflicted, 16 extra LDS cycles per phase); NHWC input: pixel-major (conflict-free stores)
    const int px = p.in_blocked ? e >> 1 : e % (RH * RW), half = p.in_blocked ? e & 1 : e / (RH * RW);
    lpy[k] = px / RW - 1;
    lpx[k] = px % RW - 1;
    ldst[k] = px * RSC + half * 4;
    lhalf[k] = half;
  }
  auto decode = [&](int it) -> Item {
    Item r;
    r.cob = it % ncob;
    const int tile = it / ncob;
    r.x0 = (tile % tiles_x) * OW;
    r.y0 = ((tile / tiles_x) % tiles_y) * OH;
    r.b = tile / (tiles_x * tiles_y);
    return r;
  };
  Item cur = decode(vb), nxt = cur;
  int item_c = vb;
  // loader cursor
  int litem = item_c, lchunk = 0;
  __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
  unsigned goff[2];
  // Input layout.  NHWC: a pixel's 8-channel chunk is 32 bytes of a Cin*4-byte record, so a wave's patch load touches 64
  // different 128-byte lines for 1 KB of payload and the 2 x 56 KB per phase of U + patch traffic evicts them from the 32 KB
  // L1 before the next chunk reuses them (measured: dense loads of the same volume shorten a phase from 3450 to 3050 cycles,
  // a round-2 timing experiment).  Channel-blocked (B, Cin/8, H, W, 8): the chunk's plane holds consecutive pixels' 32 bytes back to back --
  // a patch row of 18 pixels is 576 contiguous bytes.  The chunk's plane offset rides in the SGPR operand.
  const bool inb = p.in_blocked != 0;
  const int pxb = inb ? CK * 4 : Cin * 4;                       // bytes from one pixel to the next
  const int chunk_step = inb ? H * W * CK * 4 : CK * 4;         // bytes from one chunk to the next
  auto loader_item = [&](const Item& it, bool live) {
    lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)(live ? it.b : 0) * H * W * Cin), 0, live ? img_bytes : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int gy = it.y0 + lpy[k], gx = it.x0 + lpx[k];
      goff[k] = (live && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)((gy * W + gx) * pxb + lhalf[k] * 16) : OOB;
    }
  };
  // two register sets, one per stream-position parity: a patch is requested FOUR positions ahead and has two full phases
  // to arrive (under this kernel's L2 load a request takes 3-4 k cycles, about one phase: with one set -- requested one phase
  // before its use -- removing the raw path altogether shortened a phase by 670 of 3500 cycles: a round-2 timing experiment)
  f32x4 rr[2][2];
  auto issue_load = [&](int set) {
    const int so = __builtin_amdgcn_readfirstlane(lchunk * chunk_step);
    rr[set][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)goff[0], so, 0));
    rr[set][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrs, (int)goff[1], so, 0));
  };
  auto advance_loader = [&]() {      // kept out of the MFMA stream: a branch there ends the scheduling region
    // a REAL (uniform) branch: if-converted, the item decode below -- three divisions by run-time values, ~60 scalar
    // instructions -- was executed in every phase
    if (__builtin_expect(++lchunk == nchunk, 0)) {   // the loader moves on to this workgroup's next item
      lchunk = 0;
      litem += grid;
      const bool live = litem < nitems;
      if (live) nxt = decode(litem);
      loader_item(nxt, live);
      asm volatile("" ::: "memory");                 // keeps the block a block (no select conversion across it)
    }
  };
  auto store_raw = [&](int buf, int set) {
    float* d0 = raw + buf * RAWC + ldst[0];
    float* d1 = raw + buf * RAWC + ldst[1];
    *reinterpret_cast<f32x2*>(d0) = (f32x2){rr[set][0][0], rr[set][0][1]};
    *reinterpret_cast<f32x2*>(d0 + 2) = (f32x2){rr[set][0][2], rr[set][0][3]};
    *reinterpret_cast<f32x2*>(d1) = (f32x2){rr[set][1][0], rr[set][1][1]};
    *reinterpret_cast<f32x2*>(d1 + 2) = (f32x2){rr[set][1][2], rr[set][1][3]};
  };

  // ---- pipeline fill: positions 0 and 1 into raw[0] / raw[1], positions 2 and 3 in flight (sets 0 / 1), B panel of position 0, V[0]
  loader_item(cur, true);
  issue_load(0); advance_loader();
  store_raw(0, 0);
  issue_load(1); advance_loader();
  store_raw(1, 1);
  issue_load(0); advance_loader();
  issue_load(1); advance_loader();
  f32x4 bf[NQ];
  {
    const int uoff = __builtin_amdgcn_readfirstlane(cur.cob * nchunk * (UCH * 4));
#pragma unroll
    for (int g = 0; g < NQ; ++g) bf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, voff, uoff + g * 4096, 0));
  }
  // drain every load once: the waitcnt bookkeeping at the stream-loop header then sees only the loop's own in-order loads
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  {
    f32x2 o[6];
#pragma unroll
    for (int bb = 0; bb < 6; ++bb)
      o[bb] = __builtin_elementwise_fma(sg2, *reinterpret_cast<const f32x2*>(rpb + bb * RSC), *reinterpret_cast<const f32x2*>(rpa + bb * RSC));
    const W24Half hb = w24_batch_a(o, m5);
    f32x2 T[6];
    w24_batch_b(o, hb, T);
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) *reinterpret_cast<f32x2*>(vwr + (2 * jj) * QSL * 4) = T[jj];
  }

  f32x4 acc[24];          // never cleared: an item's first chunk starts every accumulator from a literal-zero C operand
  const f32x2 k8 = {8.f, 8.f};
  // ---- store offsets.  FASTW (the width is a whole number of tiles: every SuperPoint layer at 640x480 and
  //      1280x960): a lane's byte offsets relative to its item's first pixel never change.  The item part (tile origin,
  //      output block) goes into the BASE of a per-item buffer descriptor whose range is what is left of the image from
  //      there, so rows below the image are out of range and dropped by the hardware (the SGPR offset operand of a buffer
  //      access is NOT bounds-checked, so the item part must not ride there); no per-store address arithmetic.
  //      Output layout: NHWC, or channel-blocked (B, Cout/8, Ho, Wo, 8) for the next Winograd layer: the lane's four channels
  //      are half of chunk cob*8 + cb*2 + (lane>>5).  In the blocked layout a row below the image would land inside the next
  //      chunk's plane instead of past the end, so the fast path there also needs the height to be a whole number of tiles.
  const int Ho_k = POOL ? H >> 1 : H, Wo_k = POOL ? W >> 1 : W;
  const bool outb = p.out_blocked != 0;
  // The conditions are on the INPUT size: with floor pooling an odd-sized input (H = 8k+1, W = 16k+1, ...) has a partial last
  // tile row / column whose pooled outputs do not exist although Ho / Wo are whole numbers of pooled tiles (round 3, found by
  // the shape fuzz: 113x96 -> 56x48 wrote pooled rows 56..59 into the next channel plane, 48x49 -> 24x24 wrapped columns 24..31
  // into the next row).
  const bool fastw = (W % OW) == 0 && (!outb || (H % OH) == 0);
  const int lwr = (lane & 15) >> 2, lwc = lane & 3;
  const int opx = outb ? CK * 4 : Cout * 4;                                  // bytes from one output pixel to the next
  const int chl = outb ? (cb * 2 + (lane >> 5)) * (Ho_k * Wo_k * CK * 4) + ((lane >> 4) & 1) * 16 : (cb * 16 + 4 * (lane >> 4)) * 4;
  int soff[POOL ? 2 : 8];
#pragma unroll
  for (int e = 0; e < (POOL ? 2 : 8); ++e) {
    const int oy = POOL ? lwr : 2 * lwr + (e >> 2), ox = POOL ? 2 * lwc + e : 4 * lwc + (e & 3);
    soff[e] = (oy * Wo_k + ox) * opx + chl;
  }
  int chunk = 0;
  f32x4 bs4 = {0.f, 0.f, 0.f, 0.f};   // bias of this lane's four output channels (current item)

  if constexpr (TRACE) tprev = __builtin_readcyclecounter();
  // One phase = one 8-channel chunk of the stream.  The loop below is unrolled by TWO phases with the buffer parity as a
  // compile-time constant: every LDS address of a phase is then a per-thread base + an immediate offset (the run-time parity
  // cost 8 v_add_u32 + 3 v_lshl_add per phase, each ~4.5 cycles of matrix-pipe time).  An item has an even number of chunks
  // (Cin % 32 == 0), so the parity of a stream position is the parity of its chunk index and items end after an odd phase.
  // FIRST (an item's chunk 0): the MFMAs' C operand is the literal 0 instead of the accumulator -- clearing 24 accumulators was 48
  // v_mov_b64 in the epilogue, where a VALU instruction costs ~25 cycles beside the co-resident workgroup's MFMA stream (round 3:
  // the epilogue's store phase 3.0 k -> cycles per item in the cycle trace)
  const f32x4 zero4c = {0.f, 0.f, 0.f, 0.f};
  auto phase = [&](auto parc, auto firstc, int c) __attribute__((always_inline)) {
    constexpr int par = decltype(parc)::value ? 1 : 0;
    constexpr bool FIRST = decltype(firstc)::value;
    __syncthreads();               // V[par] and raw[par ^ 1] (position s+1) complete; the buffers written below are free
    IMX_TS(0)
    {
      // B panel of position s+1: next chunk of this item, or chunk 0 of the next item's output block
      const bool last = par == 1 && c + 1 == nchunk;
      const int ncb = last ? nxt.cob : cur.cob, nch = last ? 0 : c + 1;
      const int uoff = __builtin_amdgcn_readfirstlane((ncb * nchunk + nch) * (UCH * 4));
      bs4 = *reinterpret_cast<const f32x4*>(p.bias + cur.cob * NT + cb * 16 + 4 * (lane >> 4));      // every phase: an unconditional
      const float* vr = vrd + par * VSZ;                       // load keeps the vmcnt bookkeeping exact; the epilogue never waits for it
      float* vw = vwr + (par ^ 1) * VSZ;
      const float* pa = rpa + (par ^ 1) * RAWC;
      const float* pb = rpb + (par ^ 1) * RAWC;
      f32x2 va[6], vb[6], o[6], T[6];
        W24Half hb;
      f32x4 af[2];
      af[0] = *reinterpret_cast<const f32x4*>(vr);
#pragma unroll
      for (int g = 0; g < NQ; ++g) {
        const int cu = g & 1, nx = cu ^ 1;
        if (g + 1 < NQ) af[nx] = *reinterpret_cast<const f32x4*>(vr + (g + 1) * QSL * 4);
        acc[2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][0], af[cu][0], FIRST ? zero4c : acc[2 * g], 0, 0, 0);
        acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][2], af[cu][2], FIRST ? zero4c : acc[2 * g + 1], 0, 0, 0);
        acc[2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][1], af[cu][1], acc[2 * g], 0, 0, 0);
        acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g][3], af[cu][3], acc[2 * g + 1], 0, 0, 0);
        bf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, voff, uoff + g * 4096, 0));
        if (g == 0) {
#pragma unroll
          for (int bb = 0; bb < 6; ++bb) { va[bb] = *reinterpret_cast<const f32x2*>(pa + bb * RSC); vb[bb] = *reinterpret_cast<const f32x2*>(pb + bb * RSC); }
        }
        if (g == 1) {                // position s+2's patch (requested in phase s-2): registers -> raw[par];
          store_raw(par, par);                               // then request position s+4 into the same register set
          issue_load(par);
        }
        if (g == 3) {
#pragma unroll
          for (int bb = 0; bb < 6; ++bb) o[bb] = pk_fma(sg2, vb[bb], va[bb]);   // down the rows: F(2,3), row i
        }
        if (g == 5) {                                                                          // along the columns: F(4,3)
          hb = w24_batch_a(o, m5);
        }
        if (g == 7) {
          w24_batch_b(o, hb, T);
        }
        if (g == 9) {
#pragma unroll
          for (int jj = 0; jj < 6; ++jj) *reinterpret_cast<f32x2*>(vw + (2 * jj) * QSL * 4) = T[jj];   // position p = j*4 + i
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    advance_loader();
    IMX_TS(1)
  };
#pragma unroll 1
  for (;;) {
    phase(BoolC<false>{}, BoolC<true>{}, 0);
    phase(BoolC<true>{}, BoolC<false>{}, 1);
#pragma unroll 1
    for (chunk = 2; chunk < nchunk; chunk += 2) {
      phase(BoolC<false>{}, BoolC<false>{}, chunk);
      phase(BoolC<true>{}, BoolC<false>{}, chunk + 1);
    }

    // ---- item done: output transform Y = A2^T M A4, bias, ReLU, (2x2 max-pool), stores straight from registers.
    //      The MFMAs take U as their A operand and V as B, so D is [channel][wtile]: acc[j*4 + i][r] belongs to wtile
    //      n = lane&15 (row n>>2, column n&3) and channel cob*64 + cb*16 + 4*(lane>>4) + r -- a lane's four registers are
    //      four CONSECUTIVE CHANNELS of one pixel, i.e. one 16-byte store each (four lanes cover 64 contiguous bytes, the
    //      four waves the pixel's 256), with no LDS staging tile and no barrier.  Buffer stores through a per-image
    //      descriptor: pixels outside the image get an out-of-range offset and are dropped by the hardware.
    chunk = 0;
    {
      f32x4 y[2][4];
      w24_output_transform(acc, k8, y);
      IMX_TS(2)
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int Ho = Ho_k, Wo = Wo_k;
      typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
      // item part of every store offset: first output pixel of the tile + output block (uniform)
      const int ibase = __builtin_amdgcn_readfirstlane(((POOL ? cur.y0 >> 1 : cur.y0) * Wo + (POOL ? cur.x0 >> 1 : cur.x0)) * opx +
                                                       (outb ? cur.cob * (NT / CK) * (Ho * Wo * CK * 4) : cur.cob * NT * 4));
      const int fbase = fastw ? ibase : 0;      // FASTW: descriptor starts at the item; else at the image (offsets masked per store)
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (size_t)cur.b * Ho * Wo * Cout + (fbase >> 2)), 0,
                                                                           Ho * Wo * Cout * 4 - fbase, 0x00020000);
      if constexpr (POOL) {
        const int oy = (cur.y0 >> 1) + lwr;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4 v = __builtin_elementwise_max(__builtin_elementwise_max(y[0][2 * hh], y[0][2 * hh + 1]), __builtin_elementwise_max(y[1][2 * hh], y[1][2 * hh + 1])) + bs4;
          if (RELU) v = __builtin_elementwise_max(v, zero4);
          if (fastw) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, soff[hh], 0, 0);
          } else {
            const int ox = (cur.x0 >> 1) + 2 * lwc + hh;
            const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)(soff[hh] + ibase) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (int)off, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            f32x4 v = y[r][x] + bs4;
            if (RELU) v = __builtin_elementwise_max(v, zero4);
            if (fastw) {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, soff[r * 4 + x], 0, 0);
            } else {
              const int oy = cur.y0 + 2 * lwr + r, ox = cur.x0 + 4 * lwc + x;
              const unsigned off = (oy < Ho && ox < Wo) ? (unsigned)(soff[r * 4 + x] + ibase) : OOB;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, (int)off, 0, 0);
            }
          }
      }
    }
    IMX_TS(3)
    ++nitem_done;
    item_c += grid;
    if (item_c >= nitems) break;
    cur = nxt;
  }
#undef IMX_TS
  if constexpr (TRACE) {
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned* o = trace + (blockIdx.x * 4 + wave) * 8;
      for (int i = 0; i < 4; ++i) o[i] = tph[i];
      o[4] = (unsigned)nitem_done;
    }
  }
}

template <bool POOL, bool RELU>
hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.W + OW - 1) / OW, tiles_y = (a.H + OH - 1) / OH;
  const int nitems = tiles_x * tiles_y * a.B * (a.Cout / NT);
  const size_t lds = (size_t)(2 * VSZ + 2 * RAWC) * sizeof(float);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  auto k = conv3x3_wino24<POOL, RELU, false>;
  auto kt = conv3x3_wino24<POOL, RELU, true>;
  static unsigned long long attr[2] = {0, 0};
  raise_lds_limit(reinterpret_cast<const void*>(k), (int)lds, attr[0]);
  raise_lds_limit(reinterpret_cast<const void*>(kt), (int)lds, attr[1]);
  const dim3 grid((unsigned)(nitems < 2 * ncu ? nitems : 2 * ncu));     // persistent: two workgroups per CU
  last_form = "conv3x3_wino24:f32";
  static const bool trace = getenv("IMX_WINO_TRACE") != nullptr;     // developer instrumentation, read once per process
  if (trace) {      // bring-up instrumentation: per-phase cycle counts, averaged over items and workgroups
    static unsigned* dbuf = nullptr;
    constexpr int NREC = 1024 * 4 * 8;
    if (!dbuf) (void)hipMalloc(&dbuf, NREC * sizeof(unsigned));
    (void)hipMemsetAsync(dbuf, 0, NREC * sizeof(unsigned), s);
    hipLaunchKernelGGL(kt, grid, dim3(256), lds, s, a, tiles_x, tiles_y, nitems, dbuf);
    (void)hipStreamSynchronize(s);
    static unsigned host[NREC];
    (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
    const int n = grid.x < 1024 ? (int)grid.x : 1024;
    double sum[4] = {0}, ni = 0;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < 4; ++j) sum[j] += host[(i * 4 + 2) * 8 + j];
      ni += host[(i * 4 + 2) * 8 + 4];
    }
    const int nchunk = a.Cin / CK;
    fprintf(stderr, "[wino24n trace] H=%d W=%d Cin=%d Cout=%d pool=%d | per item: %d chunks x (barrier %.0f  phase %.0f)  epilogue transform %.0f  "
                    "stores %.0f cycles  (%.0f items per workgroup)\n", a.H, a.W, a.Cin, a.Cout, (int)POOL, nchunk, sum[0] / ni / nchunk,
            sum[1] / ni / nchunk, sum[2] / ni, sum[3] / ni, ni / n);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a, tiles_x, tiles_y, nitems, (unsigned*)nullptr);
  return hipGetLastError();
}
}  // namespace

bool conv3x3_wino24_supported(const ConvArgs& a) {
  if (a.first || a.Cin % 64 || a.Cout % NT || !a.wu24) return false;      // >= 8 chunks: the patch loader runs 4 positions ahead and must stay within the next item
  // per-image byte offsets are 31-bit, on the input AND on the output side (the heads layer writes 4x its input: ADVICE r2)
  const size_t Ho = a.pool ? a.H / 2 : a.H, Wo = a.pool ? a.W / 2 : a.W;
  return (size_t)a.H * a.W * a.Cin * 4 < (size_t)OOB && Ho * Wo * (size_t)a.Cout * 4 < (size_t)OOB;
}

hipError_t launch_conv3x3_wino24(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_wino24_supported(a)) return hipErrorInvalidValue;
  if (a.pool) return a.relu ? launch_t<true, true>(a, s) : launch_t<true, false>(a, s);
  return a.relu ? launch_t<false, true>(a, s) : launch_t<false, false>(a, s);
}

}  // namespace imx
