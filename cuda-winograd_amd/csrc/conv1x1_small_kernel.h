// Latency form of the 1x1-conv GEMM + folded BN (+ReLU) for TINY pixel counts -- the reference's own
// protocol is one image, M = 196 rows (kernel_512_one_128 & co at N = 1: Kernel128_one.cu:98,316,
// Kernel256_one.cu:100,318).
//
// At M = 196 the LDS-staged kernel (conv1x1_kernel.h: 112-row x 64/128-column tiles) has 4-8 tiles for
// 256 CUs; its split-K form spreads them over 32-64 workgroups and pays a serial gather of up to eight
// 28-56 KB slabs by whoever arrives last (1024->256: 19.4 us for 0.1 GFLOP).  Here the output is cut into
// 16 x 16 blocks (one MFMA tile: 13 x Kout/16 blocks at M = 196), a workgroup is 4 waves = 4 / KS blocks
// whose K loop is split over KS waves (KS = 4: 1024->256 is 208 workgroups, every wave contracts 256
// channels = 64 MFMAs); the KS partial tiles (1 KB each) meet in LDS.  No cross-workgroup reduction, no
// scratch, no tickets.  Operands come straight from global memory in MFMA fragment layout:
//   pixel fragment  : lane (m = lane & 15, h = lane >> 4) loads A[m][16 s + 4 h .. + 3]   (one 16-byte load per
//                     16-channel super-chunk s; MFMA k-step jj contracts channel 16 s + 4 h + jj)
//   filter fragment : lane (n = lane & 15, h) loads B[16 s + 4 h + jj][n0 + n], jj = 0..3  (four 4-byte loads:
//                     B stays in the reference's [Cin][Kout] layout, Kernel128_one.cu:40-42, untransposed)
// As in the big kernel the filter fragment is the MFMA's A operand and the pixel fragment its B operand, so a
// lane ends up with four CONSECUTIVE out-channels of one pixel: BN with four scales, one 16-byte store.
// Plain layers only (no padded operands, no residual): the chained block never runs at these sizes.
#pragma once
#include "conv1x1_kernel.h"

namespace wino {
namespace gemm1x1 {

constexpr int SMALL1_GS = 8;   // super-chunks per register buffer (8 + 32 loads); two buffers in flight

template <int KS>
__global__ void __launch_bounds__(256)
conv1x1_small_kernel(const float* __restrict__ A, const float* __restrict__ B,
                     const float* __restrict__ bnBias, const float* __restrict__ bnScale,
                     float* __restrict__ Cout, long M, int Cin, int Kout, int relu) {
  static_assert(KS == 1 || KS == 2 || KS == 4, "waves per block");
  constexpr int CB = 4 / KS;               // 16-column blocks per workgroup
  __shared__ f32x4 red[4][64];
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
  // blockIdx.x = column group: workgroups are dealt to the XCDs round-robin in x-fastest order, so the row blocks
  // that read one 16*CB-column slice of B share an XCD and its L2 (the column groups are a multiple of 8 for every
  // Kout % 128 == 0): B is then fetched once per launch instead of once per XCD
  const long m0 = (long)blockIdx.y * 16;
  const int n0 = ((int)blockIdx.x * CB + cb) * 16;
  const int kspan = Cin / KS;              // channels this wave contracts (a multiple of 16: checked on the host)
  const int nsc = kspan >> 4;
  long m = m0 + r16;
  m = m < M ? m : M - 1;                   // rows past the end read a valid row (never stored)
  // folded BN of this lane's four out-channels: requested now, used at the very end
  const int ch = n0 + 4 * h;
  f32x4 sc, bi;
#pragma unroll
  for (int j = 0; j < 4; j++) { sc[j] = bnScale[ch + j]; bi[j] = bnBias[ch + j]; }
  const float* ap = A + m * Cin + kq * kspan + 4 * h;
  const float* bp = B + (size_t)(kq * kspan + 4 * h) * Kout + n0 + r16;

  auto load_group = [&](int g, f32x4* a, float (*b)[4]) {
#pragma unroll
    for (int i = 0; i < SMALL1_GS; i++) {
      int s = g * SMALL1_GS + i;
      s = s < nsc ? s : nsc - 1;           // past the end: re-read the last one (never multiplied)
      a[i] = *(const f32x4*)(ap + s * 16);
#pragma unroll
      for (int jj = 0; jj < 4; jj++) b[i][jj] = bp[(size_t)(s * 16 + jj) * Kout];
    }
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int g, const f32x4* a, const float (*b)[4]) {
#pragma unroll
    for (int i = 0; i < SMALL1_GS; i++) {
      if (g * SMALL1_GS + i >= nsc) break;   // wave-uniform: the ragged last group
#pragma unroll
      for (int jj = 0; jj < 4; jj++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i][jj], a[i][jj], acc, 0, 0, 0);
    }
  };
  const int ngroups = (nsc + SMALL1_GS - 1) / SMALL1_GS;
  f32x4 a0[SMALL1_GS], a1[SMALL1_GS];
  float b0[SMALL1_GS][4], b1[SMALL1_GS][4];
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
  // the KS partial tiles of a block meet in its first wave, in k order (bitwise reproducible)
  if (KS > 1) {
    if (kq > 0) red[w][lane] = acc;
    __syncthreads();
    if (kq > 0) return;
#pragma unroll
    for (int j = 1; j < KS; j++) acc += red[w + j][lane];
  }
  // epilogue: lane (r16, h) holds out-channels n0 + 4 h + 0..3 of pixel row m0 + r16
  f32x4 val = sc * acc + bi;
  if (relu) {
#pragma unroll
    for (int j = 0; j < 4; j++) val[j] = fmaxf(val[j], 0.f);
  }
  if (m0 + r16 < M) *(f32x4*)(Cout + (m0 + r16) * Kout + ch) = val;
  if (clk) {
    wino_clk_slot_1x1[2] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_1x1[3] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace gemm1x1
}  // namespace wino
