// Latency-oriented variant of the fused Winograd F(2x2,3x3) kernel for SMALL batches
// (the reference's own operating point is N = 1: `./Test 0`, `./Test 1`).
//
// At N = 1 a 256->256 layer has 49 tiles: the throughput kernel (64 tiles x 64 out-channels per
// workgroup) would run on 4 of the 256 CUs.  Here a workgroup owns 16 tiles x
// 16 out-channels x all 16 Winograd points (64 accumulator VGPRs), so the same layer spreads
// over 4 x 16 = 64 CUs; the C-loop, the only serial part, is split over the workgroup's 4 waves
// (one per SIMD) whose partial accumulators are summed through LDS.  There is no LDS and no barrier:
// the MFMA A/B fragment layouts ("one tile row / one out-channel column per lane, channel pair
// by lane group") are read straight from global memory (16-byte loads), the next 16-channel
// super-chunks are prefetched into registers while the current one is transformed and multiplied.
// Requires C % 16 == 0 (the dispatcher falls back to the throughput kernel otherwise).
// Same arithmetic, same packed filter buffer and same output contract as the big kernel.
#pragma once
#include "wino_f2_fused_kernel.h"

namespace wino {
namespace fused {

constexpr int SMALL_WAVES = 4;  // waves per workgroup; each takes every 4th 16-channel super-chunk

__global__ void __launch_bounds__(64 * SMALL_WAVES)
wino_f2_small_kernel(const float* __restrict__ in, const float* __restrict__ Uq,
                     const float* __restrict__ bnBias, const float* __restrict__ bnScale,
                     float* __restrict__ out, int N, int C, int K, int relu) {
  __shared__ f32x4 red[SMALL_WAVES - 1][16][64];  // partial accumulators of waves 1..3 (48 KB)
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t16 = lane & 15, h = lane >> 4;
  const int tb16 = blockIdx.x, kq = blockIdx.y;
  const int totalTiles = N * WINO_TILES;
  const int KBLK = K >> 6;

  // The C-loop walks 16-channel super-chunks; lane group h owns channels 4h..4h+3 of each (one
  // 16-byte load per pixel / per point), and MFMA k-step jj of a super-chunk contracts channel
  // 4h+jj -- any channel<->k assignment is valid as long as A and B agree.
  // A fragment source: this lane's tile
  int g = tb16 * 16 + t16;
  g = g < totalTiles ? g : totalTiles - 1;
  const TileCoord tca = decode_tile(g);
  const float* a_src = in + ((size_t)(tca.n * WINO_HW + 2 * tca.ty) * WINO_HW + 2 * tca.tx) * C + 4 * h;
  // B fragment source: out-channel k = kq*16 + t16 inside the packed filter
  // [C/8][K/64][16 pts][64 k][8 c]: channels 4h..4h+3 of super-chunk s live in 8-channel chunk
  // 2s + (h>>1), 16-byte half (h&1) ^ bit3(kl) (see u_pos in wino_f2_fused.hip)
  const int k = kq * 16 + t16, kb = k >> 6, kl = k & 63;
  const size_t b_chunk_stride = (size_t)KBLK * U_CHUNK_FLOATS;
  const float* b_src = Uq + (size_t)(h >> 1) * b_chunk_stride + ((size_t)kb * 16 * 64 + kl) * 8 +
                       (((h & 1) ^ ((kl >> 3) & 1)) << 2);

  f32x4 acc[16];
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nsuper = C / 16;
  auto load_chunk = [&](int sc, f32x4* dd, f32x4* bb) {
    sc = sc < nsuper ? sc : nsuper - 1;  // past the end: re-read the last one, unused
    const float* ap = a_src + sc * 16;
    const float* bp = b_src + (size_t)sc * 2 * b_chunk_stride;
#pragma unroll
    for (int px = 0; px < 16; px++)
      dd[px] = *(const f32x4*)(ap + (size_t)((px >> 2) * WINO_HW + (px & 3)) * C);
#pragma unroll
    for (int e = 0; e < 16; e++) bb[e] = *(const f32x4*)(bp + e * 512);
  };
  auto compute = [&](const f32x4* d, const f32x4* bfr) {
    f32x4 tmp[16], v[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      tmp[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
      tmp[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
      tmp[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
      tmp[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[i * 4 + 0] = tmp[i * 4 + 0] - tmp[i * 4 + 2];
      v[i * 4 + 1] = tmp[i * 4 + 1] + tmp[i * 4 + 2];
      v[i * 4 + 2] = tmp[i * 4 + 2] - tmp[i * 4 + 1];
      v[i * 4 + 3] = tmp[i * 4 + 1] - tmp[i * 4 + 3];
    }
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
#pragma unroll
      for (int e = 0; e < 16; e++)
        acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e][jj], bfr[e][jj], acc[e], 0, 0, 0);
  };
  // Two super-chunks of operands (2 x 32 sixteen-byte loads: the vmcnt counter holds 63) are
  // kept in flight in registers -- a one-wave workgroup may use the whole 512-VGPR file.  The
  // loop is unrolled by two with named buffers and each refill is pinned (sched_barrier)
  // ahead of the compute it overlaps, or hipcc sinks the loads to their first use and the
  // kernel pays one full memory latency per chunk.
  // wave q contracts super-chunks q, q+4, q+8, ...: the serial chain is 4x shorter, the four
  // partial accumulators are summed through LDS at the end.
  f32x4 d0[16], b0[16], d1[16], b1[16];
  load_chunk(q, d0, b0);
  load_chunk(q + SMALL_WAVES, d1, b1);
  __builtin_amdgcn_sched_barrier(0);
  for (int it = q; it < nsuper; it += 2 * SMALL_WAVES) {
    compute(d0, b0);
    __builtin_amdgcn_sched_barrier(0);
    load_chunk(it + 2 * SMALL_WAVES, d0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (it + SMALL_WAVES < nsuper) compute(d1, b1);
    __builtin_amdgcn_sched_barrier(0);
    load_chunk(it + 3 * SMALL_WAVES, d1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (q > 0) {
#pragma unroll
    for (int e = 0; e < 16; e++) red[q - 1][e][lane] = acc[e];
  }
  __syncthreads();
  if (q > 0) return;
#pragma unroll
  for (int e = 0; e < 16; e++)
#pragma unroll
    for (int ww = 0; ww < SMALL_WAVES - 1; ww++) acc[e] += red[ww][e][lane];

  // epilogue (C/D layout: col = lane&15 = out-channel, row = 4*(lane>>4)+r = tile)
  const float sc = bnScale[k], bi = bnBias[k];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int gt = tb16 * 16 + 4 * h + r;
    if (gt >= totalTiles) continue;
    const TileCoord tc = decode_tile(gt);
    float* o = out + (size_t)tc.n * WINO_HW * WINO_HW * K + k;
    const int oy = 1 + 2 * tc.ty, ox = 1 + 2 * tc.tx;
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float m0 = acc[0 * 4 + j][r], m1 = acc[1 * 4 + j][r];
      const float m2 = acc[2 * 4 + j][r], m3 = acc[3 * 4 + j][r];
      t0[j] = m0 + m1 + m2;
      t1[j] = m1 - m2 - m3;
    }
    float y[4];
    y[0] = t0[0] + t0[1] + t0[2];
    y[1] = t0[1] - t0[2] - t0[3];
    y[2] = t1[0] + t1[1] + t1[2];
    y[3] = t1[1] - t1[2] - t1[3];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      float val = sc * y[p] + bi;
      if (relu) val = fmaxf(val, 0.f);
      o[(size_t)((oy + (p >> 1)) * WINO_HW + ox + (p & 1)) * K] = val;
    }
    // zero ring (the next 3x3 layer's padding, Kernel128_winograd.cu:163,243)
    if (tc.ty == 0) {
      o[(size_t)(ox)*K] = 0.f;
      o[(size_t)(ox + 1) * K] = 0.f;
      if (tc.tx == 0) o[0] = 0.f;
      if (tc.tx == 6) o[(size_t)15 * K] = 0.f;
    }
    if (tc.ty == 6) {
      o[(size_t)(15 * WINO_HW + ox) * K] = 0.f;
      o[(size_t)(15 * WINO_HW + ox + 1) * K] = 0.f;
      if (tc.tx == 0) o[(size_t)(15 * WINO_HW) * K] = 0.f;
      if (tc.tx == 6) o[(size_t)(15 * WINO_HW + 15) * K] = 0.f;
    }
    if (tc.tx == 0) {
      o[(size_t)(oy * WINO_HW) * K] = 0.f;
      o[(size_t)((oy + 1) * WINO_HW) * K] = 0.f;
    }
    if (tc.tx == 6) {
      o[(size_t)(oy * WINO_HW + 15) * K] = 0.f;
      o[(size_t)((oy + 1) * WINO_HW + 15) * K] = 0.f;
    }
  }
}

}  // namespace fused
}  // namespace wino
