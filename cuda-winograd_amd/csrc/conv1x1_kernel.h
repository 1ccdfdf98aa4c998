// The 1x1-conv GEMM kernel as a header, so that the library (conv1x1.hip) and the ablation tool
// (tools/ablate_1x1.hip) compile the same source.  ABLATE (0 = product): 1 skip the A LDS-DMA,
// 2 skip the B LDS-DMA, 4 skip the MFMAs, 8 skip the per-stage wait+barrier, 512 skip the stores.
//
// SK = true is the stream-K launch form.  A launch lasts as long as its busiest CU: a CU gives its
// workgroups a fixed throughput (one alone walks 32 k-steps of the 1024->256 layer in 57 us, two
// that share it take 115 us each), so the plain form costs ceil(tiles / CUs) tile-times (measured:
// 384, 448 and 512 workgroups 116-117 us, every further 256 add 57 us) and the reference's 448-tile
// layers pay for 512; with fewer tiles than CUs it leaves CUs idle while the busy ones walk whole K
// loops.  Stream-K cuts the (row tile, k-step) space into equal contiguous ranges instead; a range
// is a run of whole tiles with at most one cut tile at either end.  A segment that is not a
// whole tile stores its raw accumulators as a write-through slab and the workgroup draws one
// ticket on the tile; whoever learns that the tile's other segments are all there adds them in k
// order (bitwise reproducible) and runs the epilogue.
// Nobody waits for anybody.  Same slab / ticket rules as the fused 3x3 kernel (wino_f2_fused_kernel.h).
#pragma once
#include "wino_common.h"

#include <type_traits>

namespace wino {
namespace gemm1x1 {

#ifndef WINO_1X1_DMA0
#define WINO_1X1_DMA0 4   // first step of a stage (of 14) that issues an LDS-DMA piece of the next one;
                          // tools/ablate_1x1: 0 / 2 / 4 / 6 within 1 % on all four reference shapes, 8 up to +9 %
#endif
#ifndef WINO_1X1_SK_PRIO
#define WINO_1X1_SK_PRIO 0   // experiment, measured slower; tools may build with 1
#endif
#ifndef WINO_1X1_PROLOGUE_PRIO
#define WINO_1X1_PROLOGUE_PRIO 1
#endif
constexpr int BM = 112;
constexpr int WINO_INTERNAL_NO_BN = 1 << 16;   // not part of the public flag set
constexpr int RB = BM / 16;  // 7 row blocks

__device__ __forceinline__ void wait_lds1(int n) {
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0xC07F); break;
    case 1: __builtin_amdgcn_s_waitcnt(0xC17F); break;
    case 2: __builtin_amdgcn_s_waitcnt(0xC27F); break;
    case 3: __builtin_amdgcn_s_waitcnt(0xC37F); break;
    case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
    case 5: __builtin_amdgcn_s_waitcnt(0xC57F); break;
    case 6: __builtin_amdgcn_s_waitcnt(0xC67F); break;
    case 7: __builtin_amdgcn_s_waitcnt(0xC77F); break;
    case 8: __builtin_amdgcn_s_waitcnt(0xC87F); break;
    case 9: __builtin_amdgcn_s_waitcnt(0xC97F); break;
    case 10: __builtin_amdgcn_s_waitcnt(0xCA7F); break;
    case 11: __builtin_amdgcn_s_waitcnt(0xCB7F); break;
    case 12: __builtin_amdgcn_s_waitcnt(0xCC7F); break;
    default: __builtin_amdgcn_s_waitcnt(0xCF7F); break;
  }
}

// Geometry of the padded tensors a chained layer reads (WINO_A_PADDED) or writes (WINO_C_PADDED):
// H x W feature maps inside [N][H+2][W+2][.] -- the 3x3 layer's input / output layout.  The
// reference's stage is 14 x 14 in 16 x 16; any other size travels here.
struct PadGeo {
  unsigned hw, w;        // pixels per image, per row
  unsigned Hp, Wp;       // padded extents
  FastDiv d_hw, d_w;
};
__host__ inline PadGeo make_padgeo(int H, int W) {
  PadGeo g;
  g.hw = (unsigned)H * (unsigned)W;
  g.w = (unsigned)W;
  g.Hp = (unsigned)H + 2;
  g.Wp = (unsigned)W + 2;
  g.d_hw = make_fastdiv(g.hw);
  g.d_w = make_fastdiv(g.w);
  return g;
}
// logical pixel row m = n*H*W + y*W + x  ->  row of the padded [N][H+2][W+2][.] tensor
__device__ __forceinline__ long padded_row(long m, const PadGeo& g) {
  const unsigned mu = (unsigned)m;   // M < 2^31 (checked on the host)
  const unsigned n = fastdiv(mu, g.d_hw);
  const unsigned rem = mu - n * g.hw;
  const unsigned y = fastdiv(rem, g.d_w);
  const unsigned x = rem - y * g.w;
  return (long)n * (g.Hp * g.Wp) + (long)((y + 1) * g.Wp + x + 1);
}

template <int BK, int NW>
struct Cfg {
  static constexpr int NT = 64 * NW;                // threads per workgroup
  static constexpr int BN = 16 * NW;                // output columns per workgroup (one 16-col block per wave)
  static constexpr int S = BK / 16;                 // 16-wide k sub-chunks per stage
  static constexpr int T = S * RB;                  // pinned steps per stage (4 MFMAs each)
  static constexpr int UNITS = BK / 4;              // 16-byte units per A row
  static constexpr int A_BYTES = BM * BK * 4;
  static constexpr int B_BYTES = BK * BN * 4;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE;
  static constexpr int A_PIECES = A_BYTES / 1024;   // LDS-DMA wave-instructions per stage
  static constexpr int B_PIECES = B_BYTES / 1024;
  static constexpr int A_PER_WAVE = (A_PIECES + NW - 1) / NW;
  static constexpr int B_PER_WAVE = B_PIECES / NW;
  static constexpr int ROWS_PER_PIECE = 1024 / (BK * 4);
  static constexpr int B_UNITS = BN / 4;            // 16-byte units per B row
  static constexpr int B_ROWS_PER_PIECE = 1024 / (BN * 4);
  static __device__ __forceinline__ int fa(int row) { return BK == 64 ? (row & 15) : ((row >> 1) & 7); }
  // LDS requests at the top of step q: the A fragment of step q+2, then (on row block 2) the
  // four B values of the next sub-chunk
  static constexpr int nA(int q) { return q + 2 < T ? 1 : 0; }
  static constexpr int nB(int q) { return (q % RB == 2 && q / RB + 1 < S) ? 4 : 0; }
  // requests younger than the A fragment step t consumes
  static constexpr int wait_count(int t) {
    int after = 0;
    if (t < 2) {
      after = 1 - t;                                  // pre-loop block: B(0)x4, A(0), A(1)
      for (int q = 0; q <= t; q++) after += nA(q) + nB(q);
    } else {
      after = nB(t - 2);
      for (int q = t - 1; q <= t; q++) after += nA(q) + nB(q);
    }
    return after;
  }
};

// scratch of the stream-K form (library-owned, per stream): 2 slab slots of NW*RB KiB per
// workgroup, one ticket counter per tile (zero between launches)
struct SkArgs {
  float* slabs;
  unsigned* tickets;
  unsigned long long* dbg;   // timeline build only (ABLATE & 32768, tools/ablate_1x1 t): 8 uint64 per workgroup
  unsigned* err;             // host-visible word, set when a ticket is drawn on a counter that was not zero at launch
};

// In-kernel clock of the most recent launch (see wino_clk_slot_3x3 in wino_f2_fused_kernel.h): workgroup 0
// stores {s_memtime, s_memrealtime} at its entry and at its exit; wino_diag_last_clock(1, ...) copies them out.
__device__ unsigned long long wino_clk_slot_1x1[4];

// Resident waves per SIMD the LDS footprint allows -- two 8-wave workgroups (60 KB each) or three
// 4-wave ones (44 KB) per CU -- stated to the register allocator, which otherwise takes the
// freedom of 256 VGPRs and halves the residency (tests/test_build_budget.py).
// RES = the launch adds a residual (WINO_ADD_RESIDUAL): a compile-time property, because the two epilogues in one
// kernel cost the one without residual 2-5 % (256->1024 99.6 -> 101.3 us, 64->256 14.1 -> 14.9) through nothing but
// their presence -- register allocation and code layout of the rest.
template <int BK, int NW, int ABLATE = 0, bool SK = false, bool RES = false>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 4 : 3)
conv1x1_bn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                  const float* __restrict__ bnBias, const float* __restrict__ bnScale,
                  const float* __restrict__ R, float* __restrict__ Cout, long M, int Cin, int Kout,
                  int flags, int nMB, long batchA, long batchB, long batchC, SkArgs sk, PadGeo pg) {
  using G = Cfg<BK, NW>;
  // batched GEMMs (the 36 Winograd points of the F(4x4) compatibility path): blockIdx.y selects
  // the problem, the three operands advance by their batch strides (in floats)
  A += (size_t)blockIdx.y * batchA;
  B += (size_t)blockIdx.y * batchB;
  Cout += (size_t)blockIdx.y * batchC;
  constexpr int BN = G::BN;
  const bool relu = flags & WINO_RELU, a_padded = flags & WINO_A_PADDED;
  const bool c_padded = flags & WINO_C_PADDED;
  constexpr bool add_res = RES;   // (the host picks the instantiation from flags & WINO_ADD_RESIDUAL)
  // Output stores: non-temporal when the output is written once and the layer is MFMA-bound (a K
  // loop of at least 4 steps) -- the L2 then stays with the A / B lines other workgroups re-read.
  // Old and new libraries interleaved: 512->128 31.95 -> 31.1 us, 128->512 36.25 -> 35.6, 1024->256
  // 100.95 -> 100.4, 256->1024 102.9 -> 102.6, the 14x14 and 28x28 bottleneck blocks -0.6 %.
  // Cached when the output is the padded input of a 3x3 layer (read again at once) and for short
  // K loops, which are bound by the stores themselves and lose the L2's write combining (with
  // streaming stores everywhere the 56x56 block, whose last layer is 64->256, went 522 -> 533 us).
#ifndef WINO_1X1_NT
#define WINO_1X1_NT 1
#endif
  const bool stream_out = WINO_1X1_NT && !c_padded && Cin >= 4 * BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NBLK = Kout / BN;
  const int bid = blockIdx.x;
  const int nk = Cin / BK;
  // A new workgroup's waves are the youngest on their SIMDs, and the arbiter serves the oldest first: beside two
  // resident workgroups in their MFMA loops, the address set-up below took 2.5 us (median; 6.7 us at the 90th
  // percentile) from entry to the first LDS-DMA on the 128->512 layer (tools/ablate_1x1 t) -- time in which the
  // slot holds LDS and registers and feeds nothing.  High priority until the first stage is requested.
  if (WINO_1X1_PROLOGUE_PRIO) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63;
  if (ABLATE == 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    wino_clk_slot_1x1[0] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_1x1[1] = __builtin_amdgcn_s_memrealtime();
  }
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (ABLATE & 32768) {   // timeline: chip-wide 100 MHz stamps at entry / first MFMA / start of the last epilogue / exit
    if (threadIdx.x == 0) {
      sk.dbg[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
      sk.dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime();
      // which CU: HW_REG_HW_ID (cu bits 8-11, se bits 13-15) and HW_REG_XCC_ID
      sk.dbg[(size_t)blockIdx.x * 8 + 4] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
  }
  if (c_padded && !(ABLATE & 512)) {
    // ring pass: the padded output's zero ring (the 3x3 layer's padding) as a flat list of
    // 16-byte units -- images x ring pixels x Kout/4 units -- split over the grid
    const unsigned upp = (unsigned)Kout >> 2;
    const unsigned rpx = 2 * pg.Wp + 2 * (pg.Hp - 2);   // ring pixels per image
    const unsigned imgs = fastdiv((unsigned)M, pg.d_hw);
    const unsigned long long U = (unsigned long long)imgs * rpx * upp;
    const unsigned u_begin = (unsigned)(U * bid / gridDim.x), u_end = (unsigned)(U * (bid + 1ull) / gridDim.x);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (unsigned u = u_begin + threadIdx.x; u < u_end; u += 64 * NW) {
      const unsigned pid = u / upp, unit = u - pid * upp;
      const unsigned n = pid / rpx, q = pid - n * rpx;
      // q: [0, Wp) row 0, [Wp, 2Wp) the last row, then column 0 and the last column of rows 1..Hp-2
      const unsigned y = q < pg.Wp ? 0u : q < 2 * pg.Wp ? pg.Hp - 1 : q < 2 * pg.Wp + pg.Hp - 2 ? q - 2 * pg.Wp + 1 : q - 2 * pg.Wp - (pg.Hp - 2) + 1;
      const unsigned x = q < pg.Wp ? q : q < 2 * pg.Wp ? q - pg.Wp : q < 2 * pg.Wp + pg.Hp - 2 ? 0u : pg.Wp - 1;
      *(f32x4*)(Cout + ((size_t)(n * pg.Hp + y) * pg.Wp + x) * Kout + unit * 4) = zero4;
    }
  }
  // The work of this workgroup: [u, uend) in the space (row tile mb) * nk + k-step, for ONE column
  // block nb (tile = mb * NBLK + nb).
  //   plain form: one whole tile; blocks that share a row tile are adjacent in `slot` on one XCD
  //   stream-K  : the (mb, k) space is cut into Gr = G / NBLK equal ranges and every range is run
  //               by NBLK workgroups, one per column block: logical workgroup lg = range * NBLK + nb
  //               with lg = (bid % 8) * (G / 8) + bid / 8, so that the workgroups of a range are
  //               neighbours on one XCD and read the same A k-slices at the same time (cutting
  //               tile * nk + k instead let them drift 4 steps apart: L2 hit rate 0.71 -> 0.49,
  //               HBM fetch 115 -> 214 MB on the 1024->256 layer).  G is a multiple of 8 and of NBLK.
  const int Gsk = (int)gridDim.x;
  const int Gr = SK ? Gsk / NBLK : 1;
  const long long Usk = (long long)nMB * nk;
  auto sk_u0 = [&](int r) -> long long { return Usk * r / Gr; };
  int lg = 0, rg = 0, nb_sk = 0;
  long long u, uend;
  if (SK) {
    lg = (bid & 7) * (Gsk >> 3) + (bid >> 3);
    rg = lg / NBLK;
    nb_sk = lg - rg * NBLK;
    u = sk_u0(rg);
    uend = sk_u0(rg + 1);
  } else {
    const int xcd = bid & 7, slot = bid >> 3;
    const int mb_plain = (slot / NBLK) * 8 + xcd;
    if (mb_plain >= nMB) return;
    nb_sk = slot % NBLK;
    u = (long long)mb_plain * nk;
    uend = u + nk;
  }
  const int r16 = lane & 15, h = lane >> 4;
#if WINO_1X1_SK_PRIO
  const int sk_range_len = SK ? (int)(uend - u) : 0;
  int sk_prog = 0;
#endif
  bool first_seg = true;
  bool first_seg_stamp = true;
  unsigned long long stamp_first = 0;
#pragma unroll 1
  while (u < uend) {
  const int mb = (int)(u / nk);
  const int k0 = (int)(u - (long long)mb * nk);
  const int len = (int)((uend - u) < (long long)(nk - k0) ? (uend - u) : (long long)(nk - k0));
  const int nb = nb_sk, tile = mb * NBLK + nb;
  const long m0 = (long)mb * BM;
  const int n0 = nb * BN;
  u += len;
  // the previous segment's epilogue has read its LDS image before stage 0 is refilled
  if (SK && !first_seg) __syncthreads();
  first_seg = false;

  // ---- DMA sources --------------------------------------------------------------
  // LDS-DMA through buffer descriptors, as in the fused 3x3 kernel: per-lane byte offsets that
  // are computed once per tile plus ONE scalar k offset per operand and iteration, instead of a
  // 64-bit per-lane address add, a scalar multiply and a branch around every piece (those were
  // ~9 instructions per piece, 72-108 of the ~300 in a pair of stages).
  //   A: the descriptor starts at the tile's first row, so A itself may exceed 4 GiB.  Piece q
  //      covers rows q*RPP .. (RPP = 1 KiB / row bytes); lane -> (row, unit'), source unit =
  //      unit' ^ f(row).  Wave w issues pieces w, w+NW, ...; the two pieces past the tile's 14
  //      repeat piece 13 (same bytes to the same LDS address) rather than cost a branch.
  //      Rows past M read row M-1 (never stored).  a_padded: rows map into the padded tensor,
  //      which only grows the window by the ring pixels in between.
  //   B: one descriptor over all of B (< 4 GiB, checked on the host); piece q covers
  //      B_ROWS_PER_PIECE k rows; lane -> (k, unit'), source unit = unit' ^ 4*bit2(k).
  const long a_row0 = a_padded ? padded_row(m0 < M ? m0 : M - 1, pg) : (m0 < M ? m0 : M - 1);
  const long m_last = m0 + BM - 1 < M ? m0 + BM - 1 : M - 1;
  const long a_rows = (a_padded ? padded_row(m_last, pg) : m_last) - a_row0 + 1;
  const auto rsrc_a = make_rsrc(A + a_row0 * Cin, (unsigned)(a_rows * Cin * (long)sizeof(float)));
  const auto rsrc_b = make_rsrc(B, (unsigned)((size_t)Cin * Kout * sizeof(float)));
  unsigned a_voff[G::A_PER_WAVE];
  int a_q[G::A_PER_WAVE];
#pragma unroll
  for (int j = 0; j < G::A_PER_WAVE; j++) {
    int q = w + NW * j;
    q = q < G::A_PIECES ? q : G::A_PIECES - 1;
    a_q[j] = q;
    const int row = q * G::ROWS_PER_PIECE + lane / G::UNITS;
    const int unit = (lane % G::UNITS) ^ G::fa(row);
    long gr = m0 + row;
    gr = gr < M ? gr : M - 1;  // clamp: rows past the end read a valid row
    if (a_padded) gr = padded_row(gr, pg);
    a_voff[j] = (unsigned)((gr - a_row0) * Cin + unit * 4) * (unsigned)sizeof(float);
  }
  unsigned b_voff[G::B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < G::B_PER_WAVE; j++) {
    const int q = w + NW * j;
    const int k = G::B_ROWS_PER_PIECE * q + lane / G::B_UNITS;
    const int unit = (lane % G::B_UNITS) ^ (((k >> 2) & 1) << 2);
    b_voff[j] = (unsigned)(k * Kout + n0 + unit * 4) * (unsigned)sizeof(float);
  }
  const unsigned a_kstep = (unsigned)(BK * sizeof(float)), b_kstep = (unsigned)(BK * sizeof(float)) * (unsigned)Kout;
  auto issue_piece = [&](int stage, unsigned a_soff, unsigned b_soff, int p) {  // p = 0 .. A_PER_WAVE + B_PER_WAVE - 1
    char* sb = smem + stage * G::STAGE;
    if (p < G::A_PER_WAVE) {
      if (!(ABLATE & 1)) dma16_buf(rsrc_a, a_voff[p], a_soff, sb + a_q[p] * 1024);
    } else {
      const int j = p - G::A_PER_WAVE, q = w + NW * j;
      if (!(ABLATE & 2)) dma16_buf(rsrc_b, b_voff[j], b_soff, sb + G::A_BYTES + q * 1024);
    }
  };
  constexpr int PIECES = G::A_PER_WAVE + G::B_PER_WAVE;

  // ---- fragment addresses ---------------------------------------------------------
  // A row rb*16 + r16, sub-chunk s: unit 4s + h, stored at unit' = (4s + h) ^ f(row); f only
  // depends on r16 for both BK (16 rows = a whole number of f periods)
  int a_off[G::S];
#pragma unroll
  for (int s = 0; s < G::S; s++) a_off[s] = r16 * (BK * 4) + (((4 * s + h) ^ G::fa(r16)) << 4);
  // B element (k = 16s + 4h + j, col = 16w + r16): float index k*128 + (col ^ 16*(h&1))
  const int b_off = G::A_BYTES + ((4 * h) * BN + ((16 * w + r16) ^ ((h & 1) << 4))) * 4;

  f32x4 acc[RB];
#pragma unroll
  for (int i = 0; i < RB; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < PIECES; p++) issue_piece(0, (unsigned)k0 * a_kstep, (unsigned)k0 * b_kstep, p);
  if (WINO_1X1_PROLOGUE_PRIO) __builtin_amdgcn_s_setprio(0);

  // `more` (is there a k-step after this one to fetch) is a compile-time property of the body: the
  // last iteration of a segment is peeled below, so no piece is issued behind a branch
  auto body = [&](auto par, auto more_c, int it) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool more = decltype(more_c)::value;
    if (!(ABLATE & 8)) {
      wait_vmem_all();
      __syncthreads();
    }
    const unsigned a_soff = (unsigned)(k0 + it + 1) * a_kstep;   // `it` counts from the segment's first k-step k0
    const unsigned b_soff = (unsigned)(k0 + it + 1) * b_kstep;
    const char* st = smem + PAR * G::STAGE;
    f32x4 a[G::T];
    float b[G::S][4];
#pragma unroll
    for (int j = 0; j < 4; j++) b[0][j] = *(const float*)(st + b_off + j * BN * 4);
    a[0] = *(const f32x4*)(st + a_off[0]);
    a[1] = *(const f32x4*)(st + 2048 * (BK / 32) + a_off[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < G::T; t++) {
      const int s = t / RB, rb = t % RB;
      if (t + 2 < G::T) {
        const int s2 = (t + 2) / RB, rb2 = (t + 2) % RB;
        a[t + 2] = *(const f32x4*)(st + rb2 * 16 * BK * 4 + a_off[s2]);
      }
      if (rb == 2 && s + 1 < G::S) {
#pragma unroll
        for (int j = 0; j < 4; j++) b[s + 1][j] = *(const float*)(st + b_off + (16 * (s + 1) + j) * BN * 4);
      }
      // this wave's LDS-DMA pieces for the next stage, one per step from step WINO_1X1_DMA0 on
      if (t >= WINO_1X1_DMA0 && t - WINO_1X1_DMA0 < PIECES) {
        if (more) issue_piece(PAR ^ 1, a_soff, b_soff, t - WINO_1X1_DMA0);
      }
      __builtin_amdgcn_sched_barrier(0);
      wait_lds1(G::wait_count(t));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (ABLATE & 4) asm volatile("" ::"v"(a[t][j]), "v"(b[s][j]));
        // The filter fragment is the MFMA's A operand and the pixel fragment its B operand (both are
        // "one value per lane, index lane & 15, k = lane >> 4", so the swap is free): D = C^T, a lane then holds
        // four CONSECUTIVE out-channels 16 w + 4 h + 0..3 of pixel rb*16 + r16 -- one 16-byte store, no staging.
        else acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][j], a[t][j], acc[rb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if ((ABLATE & 32768) && stamp_first == 0) stamp_first = __builtin_amdgcn_s_memrealtime();   // the first stage is about to be waited for
  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int it = 0;
#pragma unroll 1
    for (; it + 2 < len; it += 2) {
#if WINO_1X1_SK_PRIO
      if (SK) {   // experiment (tools only): wave priority falls with the progress through the range
        const int q = 4 * (sk_prog + it) / sk_range_len;
        if (q == 0) __builtin_amdgcn_s_setprio(3);
        else if (q == 1) __builtin_amdgcn_s_setprio(2);
        else if (q == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
#endif
      body(P0{}, std::true_type{}, it);
      body(P1{}, std::true_type{}, it + 1);
    }

    if (it + 2 == len) {
      body(P0{}, std::true_type{}, it);
      body(P1{}, std::false_type{}, it + 1);
    } else {
      body(P0{}, std::false_type{}, it);
    }
  }
#if WINO_1X1_SK_PRIO
  sk_prog += len;
#endif

  if (ABLATE & 32768) {
    if (threadIdx.x == 0) {
      sk.dbg[(size_t)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();                 // overwritten by every segment: the last one stays
      if (first_seg_stamp) sk.dbg[(size_t)blockIdx.x * 8 + 1] = stamp_first;
    }
  }
  // ---- epilogue: BN (+residual) (+ReLU).
  // The accumulators are C^T tiles (see the MFMA above): lane (r16, h) holds out-channels n0 + 16 w + 4 h + 0..3 of
  // pixel rows m0 + rb*16 + r16, rb = 0..6 -- BN with four per-channel scales per lane.
  //  * No residual: one 16-byte store per row block straight from registers (a store instruction covers 16 rows x
  //    64 contiguous bytes).  No LDS image, no barrier: a wave leaves as soon as its own MFMAs are done.  (Round 1
  //    staged every tile through LDS -- two barriers, 28 ds_write_b32 + 7 ds_read_b128 per lane -- to store whole
  //    256 / 512-byte rows: 128->512 34.3 -> 31.0 us, 64->256 15.5 -> 14.5 without it.)
  //  * With a residual: the 112 x BN tile goes through LDS (the pipeline stages are free now; one ds_write_b128 per
  //    row block) and leaves as whole rows -- 16 B per lane, 512 / 256 contiguous bytes per row -- with the skip
  //    tensor read the same way.  Read 64 bytes per row and wave, the skip tensor's 128-byte lines are fetched by
  //    two waves at different times; inside the bottleneck block, where that tensor comes from HBM (a repeated
  //    stand-alone launch finds it in the Infinity Cache), the last layer went 116.6 -> 123.6 us and 252.8 ->
  //    273.1 MB per launch that way.  Image [row][col] floats, the 16-float column group XORed with (row>>2)&3.
  constexpr bool direct_epi = !add_res;
  // (opaque copies for the staged path: everything in it that only depends on the tile's position and the lane id
  //  would otherwise be computed before the main loop and carried across it -- at 128 VGPRs that means scratch
  //  spills.  The direct path keeps its hoistable form: made opaque too, 256->1024 100.2 -> 102.0 us, 64->256 +3.6 %.)
  long m0_e = m0;
  int lane_e = lane;
  asm volatile("" : "+s"(m0_e));
  asm volatile("" : "+v"(lane_e));
  const int r16_e = lane_e & 15, h_e = lane_e >> 4;
  if (!direct_epi || SK)
  __syncthreads();   // every wave is done with the pipeline stages; no LDS-DMA is in flight
  if (SK && !(k0 == 0 && len == nk)) {
    // Partial segment.  The tile is finished by whoever learns that all of its other segments
    // have been published: a range's last segment first looks at the tile's counter -- its
    // neighbours started their share of the tile long ago, so it usually finds them all there and
    // finalizes straight from its registers, publishing nothing.  Otherwise: publish, draw a
    // ticket, and unless that was the last one move on.
    constexpr unsigned SLAB = NW * RB * 1024;
    const auto rsrc_slab = make_rsrc(sk.slabs, (unsigned)((size_t)2 * Gsk * SLAB));
    const unsigned slab_voff = (unsigned)((w * RB * 64 + lane) * 16);
    // the ranges that share this row tile, in k order (their workgroups for column block nb
    // are the logical workgroups g * NBLK + nb)
    const long long x0 = (long long)mb * nk, x1 = x0 + nk - 1;
    int gA = rg, gB = rg;
    while (sk_u0(gA) > x0) gA--;
    while (gB + 1 < Gr && sk_u0(gB + 1) <= x1) gB++;
    const unsigned others = (unsigned)(gB - gA);
    bool finish = false;
    if (u >= uend) {   // last segment of the range
      if (tid == 0)
        *(volatile unsigned*)smem = __hip_atomic_load(sk.tickets + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      finish = (unsigned)__builtin_amdgcn_readfirstlane(*(volatile unsigned*)smem) == others;
    }
    if (!finish) {
      // slot 2lg for the segment that continues a tile (head of lg's range), 2lg+1 for the one that starts one
      const unsigned my_slot = 2u * (unsigned)lg + (k0 == 0 ? 1u : 0u);   // (same value as below)
#pragma unroll
      for (int rb = 0; rb < RB; rb++) slab_store16(acc[rb], rsrc_slab, slab_voff + rb * 1024, my_slot * SLAB);
      wait_vmem_all();   // the write-through stores of every wave have left ...
      __syncthreads();   // (and everyone has read the counter word above)
      if (tid == 0)      // ... before the workgroup's ticket
        *(volatile unsigned*)smem = __hip_atomic_fetch_add(sk.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const unsigned drawn = (unsigned)__builtin_amdgcn_readfirstlane(*(volatile unsigned*)smem);
      if (drawn != others) {   // someone else finishes the tile
        // ... unless the counter was not zero when the launch began (a launch that died mid-way before this one):
        // say so on the host-visible word; the library then refuses the stream until it is reset
        if (drawn > others && tid == 0) __hip_atomic_store(sk.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        continue;
      }
    }
    if (tid == 0) {   // self-cleaning counter: the next launch finds 0 again.  Subtracted, not stored: a counter that
                      // was not zero at launch stays off and the tile's last drawer is certain to see > others.  A
                      // finisher that only LOOKED took nothing: it subtracts the others' tickets and checks that
                      // nobody drew in between (nobody can, when the count it saw was true).
      const unsigned take = finish ? others : others + 1u;
      const unsigned before = __hip_atomic_fetch_sub(sk.tickets + tile, take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before != take) __hip_atomic_store(sk.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // The sum runs over the segments in k order, whoever finishes: ((s0 + s1) + s2) + ...  When
    // this workgroup's own segment is s0 or s1 it stays in the accumulators and the others are
    // added to it in order (s1 + s0 == s0 + s1 bitwise).  From position 2 on -- ranges much
    // shorter than a tile -- s0 + s1 has to be formed first: the own segment goes through its slab
    // like the others (published now if the look at the counter skipped that).
    const int pos = rg - gA;
    const unsigned my_slot = 2u * (unsigned)lg + (k0 == 0 ? 1u : 0u);
    if (pos >= 2 && finish) {
#pragma unroll
      for (int rb = 0; rb < RB; rb++) slab_store16(acc[rb], rsrc_slab, slab_voff + rb * 1024, my_slot * SLAB);
      wait_vmem_all();
    }
#pragma unroll 1
    for (int g = gA; g <= gB; g++) {
      if (pos < 2 && g == rg) continue;
      const unsigned slot = 2u * (unsigned)(g * NBLK + nb) + (g == gA ? 1u : 0u);
      f32x4 t[RB];
#pragma unroll
      for (int rb = 0; rb < RB; rb++) t[rb] = slab_load16(rsrc_slab, slab_voff + rb * 1024, slot * SLAB);
#pragma unroll
      for (int rb = 0; rb < RB; rb++) acc[rb] = (pos >= 2 && g == gA) ? t[rb] : acc[rb] + t[rb];
    }
    __syncthreads();   // everyone has read the ticket word before the image overwrites it
  }
  if (direct_epi) {
    if (ABLATE & 512) {   // price the stores: keep the accumulators (and with them the MFMAs) alive
#pragma unroll
      for (int rb = 0; rb < RB; rb++) asm volatile("" ::"v"(acc[rb]));
      continue;
    }
    const bool raw = flags & WINO_INTERNAL_NO_BN;   // plain GEMM: no scale / bias vectors at all
    const int ch = n0 + 16 * w + 4 * h;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (!raw) {
#pragma unroll
      for (int j = 0; j < 4; j++) { sc[j] = bnScale[ch + j]; bi[j] = bnBias[ch + j]; }
    }
    auto store_rows = [&](auto stream_c) {
#pragma unroll
      for (int rb = 0; rb < RB; rb++) {
        const long grow = m0 + rb * 16 + r16;
        f32x4 val = sc * acc[rb] + bi;
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; j++) val[j] = fmaxf(val[j], 0.f);
        }
        if (grow < M) {
          // c_padded: row = pixel (n, y, x) of the H x W map -> interior of [N][H+2][W+2][Kout]
          // (its zero ring is written by the ring pass at the top of the kernel)
          const long orow = c_padded ? padded_row(grow, pg) : grow;
          if (decltype(stream_c)::value) __builtin_nontemporal_store(val, (f32x4*)(Cout + orow * Kout + ch));
          else *(f32x4*)(Cout + orow * Kout + ch) = val;
        }
      }
    };
    // (two copies under one uniform branch, one per store form: inside a shared loop the optimizer folds the
    //  two stores into one plain store)
    if (stream_out) store_rows(std::true_type{});
    else store_rows(std::false_type{});
    continue;
  }
  float* img = (float*)smem;
  {
    const bool raw = flags & WINO_INTERNAL_NO_BN;   // plain GEMM: no scale / bias vectors at all
    const int cl = 16 * w + 4 * h_e;                   // this lane's four columns inside the tile
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (!raw) {
#pragma unroll
      for (int j = 0; j < 4; j++) { sc[j] = bnScale[n0 + cl + j]; bi[j] = bnBias[n0 + cl + j]; }
    }
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
      const int row = rb * 16 + r16_e;
      *(f32x4*)(img + row * BN + (cl ^ (((row >> 2) & 3) << 4))) = sc * acc[rb] + bi;   // (ReLU after the skip is added)
    }
  }
  __syncthreads();
  if (ABLATE & 512) continue;
  {
    constexpr int LPR = BN / 4;          // lanes per output row
    constexpr int RPI = 64 / LPR;        // rows per store instruction
    constexpr int RPW = BM / NW;         // rows per wave
    static_assert(RPW % RPI == 0, "rows per wave must be a whole number of store instructions");
    const int c4 = (lane_e % LPR) * 4;
    // Two copies of the row loop under one uniform branch, one per store form: inside a shared
    // loop the optimizer folds the two stores into one plain store, and an opaque pointer that
    // prevents that costs the cached form 1-2 % on the short-K layers.
    auto store_rows = [&](auto stream_c) {
#pragma unroll
      for (int k = 0; k < RPW / RPI; k++) {
        const int row = w * RPW + k * RPI + lane_e / LPR;
        f32x4 val = *(const f32x4*)(img + row * BN + (c4 ^ (((row >> 2) & 3) << 4)));
        const long grow = m0_e + row;
        if (grow < M) {
          if (add_res) {
            const f32x4 r = *(const f32x4*)(R + grow * Kout + n0 + c4);
            val += r;
            if (relu) {
#pragma unroll
              for (int j = 0; j < 4; j++) val[j] = fmaxf(val[j], 0.f);
            }
          }
          // c_padded: row = pixel (n, y, x) of the H x W map -> interior of [N][H+2][W+2][Kout]
          // (its zero ring is written by the ring pass at the top of the kernel)
          const long orow = c_padded ? padded_row(grow, pg) : grow;
          if (decltype(stream_c)::value) __builtin_nontemporal_store(val, (f32x4*)(Cout + orow * Kout + n0 + c4));
          else *(f32x4*)(Cout + orow * Kout + n0 + c4) = val;
        }
      }
    };
    if (stream_out) store_rows(std::true_type{});
    else store_rows(std::false_type{});
  }
  }   // segments
  if (ABLATE == 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    wino_clk_slot_1x1[2] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_1x1[3] = __builtin_amdgcn_s_memrealtime();
  }
  if (ABLATE & 32768) {
    if (threadIdx.x == 0) {
      sk.dbg[(size_t)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
      sk.dbg[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime();
    }
  }
}


}  // namespace gemm1x1
}  // namespace wino
