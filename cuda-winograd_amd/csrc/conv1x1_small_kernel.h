// Latency form of the 1x1-conv GEMM + folded BN (+ReLU) for FEW pixel rows -- the reference's own
// protocol is one image, M = 196 rows (kernel_512_one_128 & co at N = 1: Kernel128_one.cu:98,316,
// Kernel256_one.cu:100,318).
//
// At M = 196 the LDS-staged kernel (conv1x1_kernel.h: 112-row x 64/128-column tiles) has 4-8 tiles for
// 256 CUs; its split-K form spreads them over 32-64 workgroups and pays a serial gather of up to eight
// 28-56 KB slabs by whoever arrives last (1024->256: 19.4 us for 0.1 GFLOP).  Here the output is cut into
// BLOCKS of (16 RT) x (16 CT) -- RT x CT MFMA tiles held by ONE wave (RT = CT = 1 at M = 196: 13 x Kout/16
// blocks) -- a workgroup is 4 waves = 4 / KS blocks side by side whose K loop is split over KS waves (KS = 4:
// 1024->256 is 208 workgroups, every wave contracts 256 channels = 64 MFMAs); the KS partial blocks meet in
// LDS.  No cross-workgroup reduction, no scratch, no tickets.  Operands come straight from global memory in
// MFMA fragment layout:
//   pixel fragment  : lane (m = lane & 15, h = lane >> 4) loads A[m][16 s + 4 h .. + 3]   (one 16-byte load per
//                     row tile and 16-channel super-chunk s; MFMA k-step jj contracts channel 16 s + 4 h + jj)
//   filter fragment : lane (n = lane & 15, h) loads B[16 s + 4 h + jj][n0 + n], jj = 0..3  (four 4-byte loads per
//                     column tile: B stays in the reference's [Cin][Kout] layout, Kernel128_one.cu:40-42)
// As in the big kernel the filter fragment is the MFMA's A operand and the pixel fragment its B operand, so a
// lane ends up with four CONSECUTIVE out-channels of one pixel: BN with four scales, one 16-byte store per tile.
// The form is bound by the bytes every wave pulls through its CU's vector memory path: a 16 x 16 block costs
// 128 Cin bytes, a 32 x 32 block (RT = CT = 2: each fragment feeds two MFMA tiles) 256 Cin for four times the
// output -- half the bytes per FLOP, which is what carries the form from a handful of images to a dozen
// (conv1x1.hip: small1_plan).  The chained forms of the bottleneck block travel too (flags as in the tiled kernel:
// WINO_A_PADDED / WINO_C_PADDED = the operand is the padded [N][H+2][W+2][.] tensor of the 3x3 layer, the output's zero
// ring written by a flat pass over the grid; WINO_ADD_RESIDUAL = + residual before the ReLU), wave-uniform branches.
// CT = 4 is the WIDE form: a wave's 64 columns are cut into four tiles of STRIDED columns -- tile c = columns
// n0 + 4 j + c, j = 0..15 -- so that lane (j, h) reads B[k][n0 + 4 j .. + 3] with ONE 16-byte load per k-step and uses
// component c as tile c's operand: a quarter of the filter load instructions, each touching 8 whole cache lines (four
// k-rows x 256 B) where the four-byte loads of the other forms touch 4 half lines per 256 B.  After the MFMAs register
// r of tile c is column n0 + 16 h + 4 r + c: the four tiles' components r form 16-byte stores again.
#pragma once
#include "conv1x1_kernel.h"

namespace wino {
namespace gemm1x1 {

template <int KS, int RT = 1, int CT = 1>
__global__ void __launch_bounds__(256)
conv1x1_small_kernel(const float* __restrict__ A, const float* __restrict__ B,
                     const float* __restrict__ bnBias, const float* __restrict__ bnScale,
                     const float* __restrict__ Res, float* __restrict__ Cout, long M, int Cin, int Kout, int flags,
                     const PadGeo pg) {
  static_assert(KS == 1 || KS == 2 || KS == 4, "waves per block");
  static_assert((RT == 1 || RT == 2) && (CT == 1 || CT == 2 || CT == 4), "MFMA tiles per wave");
  constexpr bool WIDE = CT == 4;           // strided column tiles, 16-byte filter loads
  constexpr int CB = 4 / KS;               // blocks per workgroup, side by side
  constexpr int GS = RT * CT == 1 ? 8 : RT * CT >= 8 ? 2 : 4;   // super-chunks per register buffer; two buffers in flight
  constexpr int NT = RT * CT;
  __shared__ f32x4 red[4][NT][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cb = w / KS, kq = w % KS;
  // in-kernel clock of the launch (wino_diag_last_clock): block 0's first wave stamps its entry and its exit
  const bool clk = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  if (clk) {
    wino_clk_slot_1x1[0] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_1x1[1] = __builtin_amdgcn_s_memrealtime();
  }
  const int r16 = lane & 15, h = lane >> 4;
  const bool relu = flags & WINO_RELU, a_padded = flags & WINO_A_PADDED, c_padded = flags & WINO_C_PADDED;
  const bool add_res = flags & WINO_ADD_RESIDUAL;
  if (c_padded) {
    // ring pass: the padded output's zero ring (the 3x3 layer's padding) as a flat list of 16-byte units --
    // images x ring pixels x Kout/4 units -- split over the grid (as in the tiled kernel)
    const unsigned upp = (unsigned)Kout >> 2;
    const unsigned rpx = 2 * pg.Wp + 2 * (pg.Hp - 2);   // ring pixels per image
    const unsigned imgs = fastdiv((unsigned)M, pg.d_hw);
    const unsigned long long U = (unsigned long long)imgs * rpx * upp;
    const unsigned long long nblk = (unsigned long long)gridDim.x * gridDim.y, bid = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned u_begin = (unsigned)(U * bid / nblk), u_end = (unsigned)(U * (bid + 1ull) / nblk);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (unsigned u = u_begin + threadIdx.x; u < u_end; u += 256) {
      const unsigned pid = u / upp, unit = u - pid * upp;
      const unsigned n = pid / rpx, qq = pid - n * rpx;
      const unsigned y = qq < pg.Wp ? 0u : qq < 2 * pg.Wp ? pg.Hp - 1 : qq < 2 * pg.Wp + pg.Hp - 2 ? qq - 2 * pg.Wp + 1 : qq - 2 * pg.Wp - (pg.Hp - 2) + 1;
      const unsigned x = qq < pg.Wp ? qq : qq < 2 * pg.Wp ? qq - pg.Wp : qq < 2 * pg.Wp + pg.Hp - 2 ? 0u : pg.Wp - 1;
      *(f32x4*)(Cout + ((size_t)(n * pg.Hp + y) * pg.Wp + x) * Kout + unit * 4) = zero4;
    }
  }
  // blockIdx.x = column group: workgroups are dealt to the XCDs round-robin in x-fastest order, so the row blocks
  // that read one column slice of B share an XCD and its L2 (the column groups are a multiple of 8 for every
  // Kout % 128 == 0): B is then fetched once per launch instead of once per XCD
  const long m0 = (long)blockIdx.y * (16 * RT);
  const int n0 = ((int)blockIdx.x * CB + cb) * (16 * CT);
  const int kspan = Cin / KS;              // channels this wave contracts (a multiple of 16: checked on the host)
  const int nsc = kspan >> 4;
  // folded BN of this lane's out-channels: requested now, used at the very end
  // (WIDE: sc[r] = the scales of columns n0 + 16 h + 4 r .. + 3, the four tiles' components r)
  f32x4 sc[CT], bi[CT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = WIDE ? n0 + 16 * h + 4 * c + j : n0 + 16 * c + 4 * h + j;
      sc[c][j] = bnScale[col];
      bi[c][j] = bnBias[col];
    }
  const float* ap[RT];
#pragma unroll
  for (int r = 0; r < RT; r++) {
    long m = m0 + 16 * r + r16;
    m = m < M ? m : M - 1;                 // rows past the end read a valid row (never stored)
    if (a_padded) m = padded_row(m, pg);
    ap[r] = A + m * Cin + kq * kspan + 4 * h;
  }
  const float* bp = B + (size_t)(kq * kspan + 4 * h) * Kout + n0 + (WIDE ? 4 * r16 : r16);

  auto load_group = [&](int g, f32x4 (*a)[RT], float (*b)[CT][4]) {
#pragma unroll
    for (int i = 0; i < GS; i++) {
      int s = g * GS + i;
      s = s < nsc ? s : nsc - 1;           // past the end: re-read the last one (never multiplied)
#pragma unroll
      for (int r = 0; r < RT; r++) a[i][r] = *(const f32x4*)(ap[r] + s * 16);
      if constexpr (WIDE) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const f32x4 w = *(const f32x4*)(bp + (size_t)(s * 16 + jj) * Kout);   // columns n0 + 4 r16 .. + 3 of k-row 16 s + 4 h + jj
#pragma unroll
          for (int c = 0; c < 4; c++) b[i][c][jj] = w[c];
        }
      } else {
#pragma unroll
        for (int c = 0; c < CT; c++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) b[i][c][jj] = bp[(size_t)(s * 16 + jj) * Kout + 16 * c];
      }
    }
  };
  f32x4 acc[RT][CT];
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int c = 0; c < CT; c++) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int g, const f32x4 (*a)[RT], const float (*b)[CT][4]) {
#pragma unroll
    for (int i = 0; i < GS; i++) {
      if (g * GS + i >= nsc) break;        // wave-uniform: the ragged last group
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
          for (int c = 0; c < CT; c++)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i][c][jj], a[i][r][jj], acc[r][c], 0, 0, 0);
    }
  };
  const int ngroups = (nsc + GS - 1) / GS;
  f32x4 a0[GS][RT], a1[GS][RT];
  float b0[GS][CT][4], b1[GS][CT][4];
  load_group(0, a0, b0);
  if (ngroups > 1) load_group(1, a1, b1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int g = 0; g < ngroups; g += 2) {
    compute(g, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 2 < ngroups) load_group(g + 2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 >= ngroups) break;
    compute(g + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 3 < ngroups) load_group(g + 3, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // the KS partial blocks meet in the block's first wave, in k order (bitwise reproducible)
  if (KS > 1) {
    if (kq > 0) {
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int c = 0; c < CT; c++) red[w][r * CT + c][lane] = acc[r][c];
    }
    __syncthreads();
    if (kq > 0) return;
#pragma unroll
    for (int j = 1; j < KS; j++)
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int c = 0; c < CT; c++) acc[r][c] += red[w + j][r * CT + c][lane];
  }
  // epilogue: lane (r16, h) holds out-channels n0 + 16 c + 4 h + 0..3 of pixel row m0 + 16 r + r16
#pragma unroll
  for (int r = 0; r < RT; r++) {
    const long row = m0 + 16 * r + r16;
    const long orow = c_padded && row < M ? padded_row(row, pg) : row;
#pragma unroll
    for (int c = 0; c < CT; c++) {
      // WIDE: c is the register index here: the four tiles' components c are columns n0 + 16 h + 4 c .. + 3
      f32x4 val;
      if constexpr (WIDE) val = (f32x4){acc[r][0][c], acc[r][1][c], acc[r][2][c], acc[r][3][c]};
      else val = acc[r][c];
      val = sc[c] * val + bi[c];
      const int col = n0 + (WIDE ? 16 * h + 4 * c : 16 * c + 4 * h);
      if (add_res && row < M) val += *(const f32x4*)(Res + row * Kout + col);   // the residual is never padded
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; j++) val[j] = fmaxf(val[j], 0.f);
      }
      if (row < M) *(f32x4*)(Cout + orow * Kout + col) = val;
    }
  }
  if (clk) {
    wino_clk_slot_1x1[2] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_1x1[3] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace gemm1x1
}  // namespace wino
