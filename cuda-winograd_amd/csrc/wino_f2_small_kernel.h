// Latency-oriented variant of the fused Winograd F(2x2,3x3) kernel for SMALL batches
// (the reference's own operating point is N = 1: `./Test 0`, `./Test 1`,
// Kernel128_winograd.cu:263-265 / Kernel256_winograd.cu:266-268 at one image).
//
// At N = 1 a 256->256 layer has 49 tiles and 4.2 MB of filters: the work is tiny, what costs is (i) how
// many CUs pull the operands -- one CU takes in 60-70 GB/s of scattered 16-byte loads, so a block's
// operands must be spread over many CUs -- and (ii) the latency chain launch -> loads -> MFMAs -> store.
//
// Work decomposition.  An output BLOCK is 16 tiles x 16 out-channels (one MFMA tile per Winograd point).
// Its contraction runs over (16-channel super-chunk, point row): nsuper x 4 / PR TASKS, where a task is
// one super-chunk for PR of the four rows of the 4 x 4 point grid:
//     PR = 4: 16 patch pixels + 16 filter points = 32 sixteen-byte loads per lane, 64 MFMAs
//     PR = 2: 12 pixels (3 patch rows) + 8 points, 32 MFMAs        PR = 1: 8 pixels + 4 points, 16 MFMAs
// A workgroup is 4 waves (one per SIMD); S workgroups share a block (gridDim.z = S, "C-split"), so the
// block's tasks are dealt round-robin over 4 S waves.  PR and S are chosen on the host so that the grid
// just fills the CUs: 256 channels N = 1 is 64 blocks x S = 4 (PR = 4, one task per wave); 128 channels
// N = 1 is 32 blocks x S = 8 with PR = 1.  (Round 2's kernel had no C-split: 64 / 32 workgroups on 256 CUs,
// 18.8 / 15.1 us, bound by the load bandwidth of the few busy CUs.)
//
// No LDS staging and no barrier in the loop: the MFMA A/B fragment layouts ("one tile row / one out-channel
// column per lane, channel by lane group") are read straight from global memory with 16-byte loads (lane
// group h owns channels 4h..4h+3 of a super-chunk; MFMA k-step jj contracts channel 4h+jj -- any
// channel<->k assignment is valid as long as A and B agree), the next task's operands are in flight in
// registers while the current one is transformed and multiplied.
//
// Reduction, two levels, all on POST-transform values (A^T m A is linear, and a block's 2x2 outputs are
// 4 KB where its 16 accumulator tiles are 16 KB):
//   1. each wave applies its part of A^T m A in-lane; waves 1..3 hand their 4 KB to wave 0 through LDS;
//   2. S > 1: wave 0 publishes the workgroup's partial block as a write-through slab (4 x 16-byte sc1
//      stores per lane), drains them, and draws ONE ticket on the block's counter; whoever draws the last
//      ticket loads all S slabs at once (4 S loads in flight), adds them in split order (bitwise
//      reproducible whoever finishes), applies BN + ReLU and stores.  Nobody waits for anybody -- the same
//      slab / ticket rules as the throughput kernel (wino_f2_fused_kernel.h).
// Requires C % 16 == 0 and the 14x14 map (the dispatcher takes the throughput kernel otherwise).
// Same arithmetic, same packed filter buffer and same output contract as the big kernel.
// Two kernels: wino_f2_small_kernel<PR> (16 x 16 blocks, the N = 1 forms above) and, below it,
// wino_f2_small2_kernel<CT> (blocks of 16 tiles x 32 / 64 out-channels per wave for a few images up to ~20).
#pragma once
#include "wino_f2_fused_kernel.h"

namespace wino {
namespace fused {

constexpr int SMALL_WAVES = 4;          // waves per workgroup
constexpr int SMALL_MAX_SPLIT = 8;      // S <= 8: the finisher keeps 4 S sixteen-byte loads in flight
constexpr int SMALL_SLAB_BYTES = 4096;  // a block's pre-BN 2x2 outputs: 16 tiles x 16 k x 4 px x 4 B

struct SmallParams {
  const float* in;
  const float* Uq;
  const float* bnBias;
  const float* bnScale;
  float* out;
  int N, C, K, relu;
  float* slabs;              // [block][S] x (block width / 16) x 4 KB (S > 1 only)
  unsigned* tickets;         // [block]
  unsigned* err;             // host-visible word: set when a ticket counter was found dirty (S > 1 only)
  unsigned long long* dbg;   // timeline build only (DIAG, tools/small_timeline): 8 stamps per workgroup
};

// DIAG = true is the timeline build (tools/small_timeline.hip): wave 0 of every workgroup stores s_memrealtime
// (100 MHz, chip-wide) at entry, operands requested, MFMAs done, LDS level done, slab drained, ticket drawn, gather
// landed, exit.  The product kernel is DIAG = false.
template <int PR, bool DIAG = false>
__global__ void __launch_bounds__(64 * SMALL_WAVES)
wino_f2_small_kernel(const SmallParams prm) {
  static_assert(PR == 1 || PR == 2 || PR == 4, "point rows per task");
  constexpr int NPR = 4 / PR;      // tasks per super-chunk
  constexpr int NROW = PR == 4 ? 4 : PR + 1;   // patch rows a task reads
  constexpr int NPX = 4 * NROW, NPT = 4 * PR;  // sixteen-byte loads per lane: pixels, points
  __shared__ f32x4 red[SMALL_WAVES - 1][4][64];  // post-transform partials of waves 1..3 (12 KB)
  const float* __restrict__ in = prm.in;
  const float* __restrict__ Uq = prm.Uq;
  const int N = prm.N, C = prm.C, K = prm.K;
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t16 = lane & 15, h = lane >> 4;
  // blockIdx.x = out-channel block: blocks are dealt to the XCDs round-robin in x-fastest order, so the workgroups
  // that read one filter slice (all tile blocks, all C-splits of a kq) share an XCD and its L2 (K/16 is a multiple of
  // 8 for every K % 128 == 0); the slice is then fetched once per launch instead of once per tile block
  const int tb16 = blockIdx.y, kq = blockIdx.x;
  const int S = gridDim.z, split = blockIdx.z;
  auto mark = [&](int i) {
    if (DIAG && threadIdx.x == 0) {
      __builtin_amdgcn_sched_barrier(0);
      prm.dbg[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + i] = __builtin_amdgcn_s_memrealtime();   // (any order: the tool sorts)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  mark(0);
  // in-kernel clock of the launch (wino_diag_last_clock): block 0's first wave stamps its entry and its exit
  const bool clk = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
  if (clk) {
    wino_clk_slot_3x3[0] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_3x3[1] = __builtin_amdgcn_s_memrealtime();
  }
  auto clk_exit = [&]() {
    if (clk) {
      wino_clk_slot_3x3[2] = __builtin_amdgcn_s_memtime();
      wino_clk_slot_3x3[3] = __builtin_amdgcn_s_memrealtime();
    }
  };
  const int totalTiles = N * WINO_TILES;
  const int KBLK = K >> 6;

  // this wave's tasks: t = gw, gw + 4 S, ...  (t = super-chunk * NPR + row group; 4 S is a multiple of NPR,
  // so the row group prg is the same for all of a wave's tasks)
  const int gw = split * SMALL_WAVES + q;
  const int prg = gw % NPR;
  const int nsuper = C / 16;
  const int ntask = nsuper * NPR;
  const int stride = SMALL_WAVES * S;
  // Patch rows the wave reads ("slots") and the signs of its B^T d rows, by PR:
  //   PR = 4: slots = rows 0..3;  tmp0 = s0 - s2, tmp1 = s1 + s2, tmp2 = s2 - s1, tmp3 = s1 - s3
  //   PR = 2 (point rows 2g, 2g+1): slots = (d0, d2, d1) for g = 0, (d2, d1, d3) for g = 1;
  //           tmp0 = s0 - s1, tmp1 = s1 + sg * s2 with sg = +1 / -1        (the throughput kernel's rule)
  //   PR = 1 (point row i): slots = (d0,d2) (d1,d2) (d2,d1) (d1,d3); tmp0 = s0 + sg * s1, sg = -1 +1 -1 -1
  int slot_row[NROW];
  float sg = 1.f;
  if constexpr (PR == 4) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) slot_row[kk] = kk;
  } else if constexpr (PR == 2) {
    slot_row[0] = prg ? 2 : 0;
    slot_row[1] = prg ? 1 : 2;
    slot_row[2] = prg ? 3 : 1;
    sg = prg ? -1.f : 1.f;
  } else {
    slot_row[0] = prg == 0 ? 0 : prg == 2 ? 2 : 1;
    slot_row[1] = prg == 3 ? 3 : prg == 2 ? 1 : 2;
    sg = prg == 1 ? 1.f : -1.f;
  }

  // A fragment source: this lane's tile, channels 4h..4h+3 of the super-chunk
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
                       (((h & 1) ^ ((kl >> 3) & 1)) << 2) + (size_t)(PR * prg) * 4 * 512;   // the wave's first point

  // folded BN of this lane's four out-channels kq*16 + 4h .. + 3 (see the operand swap in compute()): requested
  // now, used by the finisher at the very end (loaded there, they were one more memory round trip on the critical path)
  const int k4 = kq * 16 + 4 * h;
  f32x4 sc4, bi4;
#pragma unroll
  for (int r = 0; r < 4; r++) { sc4[r] = prm.bnScale[k4 + r]; bi4[r] = prm.bnBias[k4 + r]; }

  f32x4 acc[NPT];
#pragma unroll
  for (int e = 0; e < NPT; e++) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load_task = [&](int t, f32x4* dd, f32x4* bb) {
    const int sc = t / NPR;
    const float* ap = a_src + sc * 16;
    const float* bp = b_src + (size_t)sc * 2 * b_chunk_stride;
#pragma unroll
    for (int kk = 0; kk < NROW; kk++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        dd[kk * 4 + j] = *(const f32x4*)(ap + (size_t)(slot_row[kk] * WINO_HW + j) * C);
#pragma unroll
    for (int e = 0; e < NPT; e++) bb[e] = *(const f32x4*)(bp + e * 512);
  };
  auto compute = [&](const f32x4* d, const f32x4* bfr) {
    f32x4 tmp[NPT], v[NPT];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if constexpr (PR == 4) {
        tmp[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
        tmp[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
        tmp[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
        tmp[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
      } else if constexpr (PR == 2) {
        tmp[0 * 4 + j] = d[0 * 4 + j] - d[1 * 4 + j];
        tmp[1 * 4 + j] = d[1 * 4 + j] + sg * d[2 * 4 + j];
      } else {
        tmp[j] = d[j] + sg * d[4 + j];
      }
    }
#pragma unroll
    for (int i = 0; i < PR; i++) {
      v[i * 4 + 0] = tmp[i * 4 + 0] - tmp[i * 4 + 2];
      v[i * 4 + 1] = tmp[i * 4 + 1] + tmp[i * 4 + 2];
      v[i * 4 + 2] = tmp[i * 4 + 2] - tmp[i * 4 + 1];
      v[i * 4 + 3] = tmp[i * 4 + 1] - tmp[i * 4 + 3];
    }
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
#pragma unroll
      for (int e = 0; e < NPT; e++)
        // the filter fragment as the MFMA's A operand, the transformed pixels as its B operand (both are "one value
        // per lane, index lane & 15, k = lane >> 4": the swap is free): D = C^T, register r of lane (t16, h) is
        // out-channel kq*16 + 4h + r of tile t16 -- four CONSECUTIVE out-channels per lane, 16-byte output stores
        acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[e][jj], v[e][jj], acc[e], 0, 0, 0);
  };
  // Two tasks' operands are kept in flight in registers (a one-wave-per-SIMD workgroup may use the
  // whole 512-VGPR file).  The loop is unrolled by two with named buffers and each refill is pinned
  // (sched_barrier) ahead of the compute it overlaps, or hipcc sinks the loads to their first use and
  // the kernel pays one full memory latency per task.
  f32x4 d0[NPX], b0[NPT], d1[NPX], b1[NPT];
  int t = gw;
  if (t < ntask) load_task(t, d0, b0);
  if (t + stride < ntask) load_task(t + stride, d1, b1);
  __builtin_amdgcn_sched_barrier(0);
  mark(1);
#pragma unroll 1
  while (t < ntask) {
    compute(d0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 * stride < ntask) load_task(t + 2 * stride, d0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + stride >= ntask) break;
    compute(d1, b1);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 3 * stride < ntask) load_task(t + 3 * stride, d1, b1);
    __builtin_amdgcn_sched_barrier(0);
    t += 2 * stride;
  }

  // ---- the wave's part of A^T m A (C/D layout after the operand swap: col = lane&15 = tile, row = 4*(lane>>4)+r =
  // out-channel 4h + r of the block).
  // Per point row i:  c0(i) = m_i0 + m_i1 + m_i2,  c1(i) = m_i1 - m_i2 - m_i3;  then
  //   Y[0][b] = c_b(0) + c_b(1) + c_b(2),   Y[1][b] = c_b(1) - c_b(2) - c_b(3)
  // of which this wave adds the terms of its rows (selects on the wave-uniform row index: exact, and
  // an Inf in one part cannot turn another into NaN).  y[r] = the 2x2 pixels (p = 2a + b) of this lane's tile, out-channel 4h+r.
  f32x4 y[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float c[PR][2];
#pragma unroll
    for (int i = 0; i < PR; i++) {
      const float m0 = acc[i * 4 + 0][r], m1 = acc[i * 4 + 1][r];
      const float m2 = acc[i * 4 + 2][r], m3 = acc[i * 4 + 3][r];
      c[i][0] = m0 + m1 + m2;
      c[i][1] = m1 - m2 - m3;
    }
#pragma unroll
    for (int bb = 0; bb < 2; bb++) {
      if constexpr (PR == 4) {
        y[r][bb] = c[0][bb] + c[1][bb] + c[2][bb];
        y[r][2 + bb] = c[1][bb] - c[2][bb] - c[3][bb];
      } else if constexpr (PR == 2) {
        const float sum = c[0][bb] + c[1][bb];
        y[r][bb] = prg ? c[0][bb] : sum;                 // rows (0,1): c0 + c1;  rows (2,3): c2
        y[r][2 + bb] = prg ? -sum : c[1][bb];            // rows (0,1): c1;       rows (2,3): -(c2 + c3)
      } else {
        y[r][bb] = prg == 3 ? 0.f : c[0][bb];
        y[r][2 + bb] = prg == 0 ? 0.f : prg == 1 ? c[0][bb] : -c[0][bb];
      }
    }
  }
  if (DIAG) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  mark(2);
  // ---- level 1: the workgroup's four partial blocks meet in wave 0 (in wave order)
  if (q > 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) red[q - 1][r][lane] = y[r];
  }
  __syncthreads();
  if (q > 0) return;
#pragma unroll
  for (int ww = 0; ww < SMALL_WAVES - 1; ww++)
#pragma unroll
    for (int r = 0; r < 4; r++) y[r] += red[ww][r][lane];

  mark(3);
  // ---- level 2: the S workgroups of a block meet through write-through slabs + one ticket per workgroup
  if (S > 1) {
    const int block = tb16 * (K >> 4) + kq;
    const auto rsrc_slab = make_rsrc(prm.slabs, (unsigned)((size_t)gridDim.x * gridDim.y * S * SMALL_SLAB_BYTES));
    const unsigned base = (unsigned)(block * S) * SMALL_SLAB_BYTES;
#pragma unroll
    for (int r = 0; r < 4; r++)
      slab_store16(y[r], rsrc_slab, (unsigned)((r * 64 + lane) * 16), base + (unsigned)split * SMALL_SLAB_BYTES);
    wait_vmem_all();   // the write-through stores have left ...
    mark(4);
    unsigned old = 0;
    if (lane == 0)     // ... before the ticket
      old = __hip_atomic_fetch_add(prm.tickets + block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    mark(5);
    if (old != (unsigned)(S - 1)) {
      // a counter that was not zero when the launch began (an aborted launch before this one): say so
      if (old >= (unsigned)S && lane == 0) __hip_atomic_store(prm.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      clk_exit();
      return;          // another workgroup finishes the block
    }
    if (lane == 0)     // self-cleaning counter: the next launch finds 0 again (subtracted, not stored: a counter
                       // that was not zero at launch stays off, and the block's last drawer is certain to see >= S)
      __hip_atomic_fetch_sub(prm.tickets + block, (unsigned)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 part[SMALL_MAX_SPLIT][4];
#pragma unroll
    for (int s = 0; s < SMALL_MAX_SPLIT; s++) {
      if (s < S) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          part[s][r] = slab_load16(rsrc_slab, (unsigned)((r * 64 + lane) * 16), base + (unsigned)s * SMALL_SLAB_BYTES);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) y[r] = part[0][r];
    if (DIAG) { wait_vmem_all(); mark(6); }
#pragma unroll
    for (int s = 1; s < SMALL_MAX_SPLIT; s++) {
      if (s < S) {
#pragma unroll
        for (int r = 0; r < 4; r++) y[r] += part[s][r];
      }
    }
  }

  // ---- finalize: BN + ReLU + store (and the block's share of the zero ring).  One tile and four consecutive
  // out-channels per lane: every store is 16 bytes, the four lane groups of a tile cover 64 contiguous bytes.
  // (Round 3's first cut held one out-channel of four tiles per lane: 16 four-byte stores and their drain were
  // 1.6 us of the finisher's 8.8; tools/small_timeline.)
  float* __restrict__ out = prm.out;
  const int relu = prm.relu;
  const int gt = tb16 * 16 + t16;
  if (gt < totalTiles) {
    const TileCoord tc = decode_tile(gt);
    float* o = out + (size_t)tc.n * WINO_HW * WINO_HW * K + k4;
    const int oy = 1 + 2 * tc.ty, ox = 1 + 2 * tc.tx;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 4; p++) {
      f32x4 val = {y[0][p], y[1][p], y[2][p], y[3][p]};
      val = sc4 * val + bi4;
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; r++) val[r] = fmaxf(val[r], 0.f);
      }
      *(f32x4*)(o + (size_t)((oy + (p >> 1)) * WINO_HW + ox + (p & 1)) * K) = val;
    }
    // zero ring (the next 3x3 layer's padding, Kernel128_winograd.cu:163,243)
    if (tc.ty == 0) {
      *(f32x4*)(o + (size_t)(ox)*K) = zero4;
      *(f32x4*)(o + (size_t)(ox + 1) * K) = zero4;
      if (tc.tx == 0) *(f32x4*)(o) = zero4;
      if (tc.tx == 6) *(f32x4*)(o + (size_t)15 * K) = zero4;
    }
    if (tc.ty == 6) {
      *(f32x4*)(o + (size_t)(15 * WINO_HW + ox) * K) = zero4;
      *(f32x4*)(o + (size_t)(15 * WINO_HW + ox + 1) * K) = zero4;
      if (tc.tx == 0) *(f32x4*)(o + (size_t)(15 * WINO_HW) * K) = zero4;
      if (tc.tx == 6) *(f32x4*)(o + (size_t)(15 * WINO_HW + 15) * K) = zero4;
    }
    if (tc.tx == 0) {
      *(f32x4*)(o + (size_t)(oy * WINO_HW) * K) = zero4;
      *(f32x4*)(o + (size_t)((oy + 1) * WINO_HW) * K) = zero4;
    }
    if (tc.tx == 6) {
      *(f32x4*)(o + (size_t)(oy * WINO_HW + 15) * K) = zero4;
      *(f32x4*)(o + (size_t)((oy + 1) * WINO_HW + 15) * K) = zero4;
    }
  }
  if (DIAG) { wait_vmem_all(); mark(7); }
  clk_exit();
}


// ---------------------------------------------------------------------------------------------------------------
// The same kernel with WIDER blocks per wave, for the batches between the reference's one image and the throughput
// kernel's range: a wave holds CT MFMA tiles side by side (16 tiles x 16 CT out-channels) of 8 points (PR = 2: two
// rows of the point grid), so that every pixel fragment -- the expensive operand: a 16-byte load per lane whose 64
// lanes touch 16 cache lines, against 8 for a filter fragment -- and its B^T d B transform feed CT MFMAs.  Measured
// per wave-task (4 waves per CU, operands in L2): 0.11 us per pixel load, 0.05 per filter load; a 16 x 16 block costs
// 16 + 16 loads per 64 MFMAs, a 16 x 32 block 12 + 16, a 16 x 64 block 12 + 32 per 128 -- the form is bound by what
// a CU's vector memory path takes in, not by the MFMAs (8 passes each).  (Blocks of 32 tiles -- two pixel fragments
// per filter fragment -- were built and measured too: 30-38 us where these take 21-28, the pixel loads being the
// dear ones; not kept.)  128 accumulator registers at CT = 4 leave no room for a second operand buffer: the next
// task's loads are issued PROGRESSIVELY instead -- its pixels as soon as the transform has consumed the current
// ones, each out-channel block's filter points as soon as that block's MFMAs are done.  Reductions, slabs, tickets
// and the finalize are the 16 x 16 kernel's, per tile of the block.
template <int CT>
__global__ void __launch_bounds__(64 * SMALL_WAVES)
wino_f2_small2_kernel(const SmallParams prm) {
  static_assert(CT == 2 || CT == 4, "MFMA tiles per wave");
  __shared__ f32x4 red[SMALL_WAVES - 1][CT * 4][64];   // post-transform partials of waves 1..3 (48 KB at CT = 4)
  const float* __restrict__ in = prm.in;
  const float* __restrict__ Uq = prm.Uq;
  const int N = prm.N, C = prm.C, K = prm.K;
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t16 = lane & 15, h = lane >> 4;
  const int tb16 = blockIdx.y, kqq = blockIdx.x;       // x = out-channel block: the workgroups sharing a filter slice share an XCD
  const int S = gridDim.z, split = blockIdx.z;
  const bool clk = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
  if (clk) {
    wino_clk_slot_3x3[0] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_3x3[1] = __builtin_amdgcn_s_memrealtime();
  }
  auto clk_exit = [&]() {
    if (clk) {
      wino_clk_slot_3x3[2] = __builtin_amdgcn_s_memtime();
      wino_clk_slot_3x3[3] = __builtin_amdgcn_s_memrealtime();
    }
  };
  const int totalTiles = N * WINO_TILES;
  const int KBLK = K >> 6;
  // tasks: t = super-chunk * 2 + row group (point rows 2g, 2g+1); wave gw takes t = gw, gw + 4 S, ... (same g throughout)
  const int gw = split * SMALL_WAVES + q;
  const int prg = gw & 1;
  const int ntask = (C / 16) * 2;
  const int stride = SMALL_WAVES * S;
  // patch rows ("slots") of the row group: (d0, d2, d1) / (d2, d1, d3); tmp0 = s0 - s1, tmp1 = s1 + sg * s2
  const int slot0 = prg ? 2 : 0, slot1 = prg ? 1 : 2, slot2 = prg ? 3 : 1;
  const float sg = prg ? -1.f : 1.f;

  int g = tb16 * 16 + t16;
  g = g < totalTiles ? g : totalTiles - 1;
  const TileCoord tca = decode_tile(g);
  const float* a_src = in + ((size_t)(tca.n * WINO_HW + 2 * tca.ty) * WINO_HW + 2 * tca.tx) * C + 4 * h;
  const size_t b_chunk_stride = (size_t)KBLK * U_CHUNK_FLOATS;
  const float* b_src[CT];
#pragma unroll
  for (int c = 0; c < CT; c++) {
    const int k = (kqq * CT + c) * 16 + t16, kb = k >> 6, kl = k & 63;
    b_src[c] = Uq + (size_t)(h >> 1) * b_chunk_stride + ((size_t)kb * 16 * 64 + kl) * 8 + (((h & 1) ^ ((kl >> 3) & 1)) << 2) +
               (size_t)(2 * prg) * 4 * 512;   // the wave's first point
  }
  // folded BN of this lane's out-channels (kqq*CT + c)*16 + 4h .. + 3: requested now, used by the finisher
  f32x4 sc4[CT], bi4[CT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sc4[c][r] = prm.bnScale[(kqq * CT + c) * 16 + 4 * h + r];
      bi4[c][r] = prm.bnBias[(kqq * CT + c) * 16 + 4 * h + r];
    }

  f32x4 acc[8][CT];
#pragma unroll
  for (int e = 0; e < 8; e++)
#pragma unroll
    for (int c = 0; c < CT; c++) acc[e][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 d[12], b[CT][8];
  auto load_a = [&](int t) {
    const float* ap = a_src + (t >> 1) * 16;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      d[0 * 4 + j] = *(const f32x4*)(ap + (size_t)(slot0 * WINO_HW + j) * C);
      d[1 * 4 + j] = *(const f32x4*)(ap + (size_t)(slot1 * WINO_HW + j) * C);
      d[2 * 4 + j] = *(const f32x4*)(ap + (size_t)(slot2 * WINO_HW + j) * C);
    }
  };
  auto load_b = [&](int t, int c) {
    const float* bp = b_src[c] + (size_t)(t >> 1) * 2 * b_chunk_stride;
#pragma unroll
    for (int e = 0; e < 8; e++) b[c][e] = *(const f32x4*)(bp + e * 512);
  };
  int t = gw;
  if (t < ntask) {
    load_a(t);
#pragma unroll
    for (int c = 0; c < CT; c++) load_b(t, c);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (; t < ntask; t += stride) {
    const bool more = t + stride < ntask;
    f32x4 tmp[8], v[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      tmp[0 * 4 + j] = d[0 * 4 + j] - d[1 * 4 + j];
      tmp[1 * 4 + j] = d[1 * 4 + j] + sg * d[2 * 4 + j];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      v[i * 4 + 0] = tmp[i * 4 + 0] - tmp[i * 4 + 2];
      v[i * 4 + 1] = tmp[i * 4 + 1] + tmp[i * 4 + 2];
      v[i * 4 + 2] = tmp[i * 4 + 2] - tmp[i * 4 + 1];
      v[i * 4 + 3] = tmp[i * 4 + 1] - tmp[i * 4 + 3];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) load_a(t + stride);            // the pixels of the next task: d is dead from here on
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CT; c++) {
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int e = 0; e < 8; e++)
          // filter fragment = the MFMA's A operand, transformed pixels its B operand: register r of lane (t16, h) is
          // out-channel 4h + r of tile t16 (see the 16 x 16 kernel)
          acc[e][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[c][e][jj], v[e][jj], acc[e][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) load_b(t + stride, c);       // ... and this out-channel block's points, behind its MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the wave's part of A^T m A: y[c][r] = the 2x2 pixels (p = 2a + b) of tile t16, out-channel block c, channel 4h + r
  f32x4 y[CT][4];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float cc[2][2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const float m0 = acc[i * 4 + 0][c][r], m1 = acc[i * 4 + 1][c][r];
        const float m2 = acc[i * 4 + 2][c][r], m3 = acc[i * 4 + 3][c][r];
        cc[i][0] = m0 + m1 + m2;
        cc[i][1] = m1 - m2 - m3;
      }
#pragma unroll
      for (int bb = 0; bb < 2; bb++) {
        const float sum = cc[0][bb] + cc[1][bb];
        y[c][r][bb] = prg ? cc[0][bb] : sum;             // rows (0,1): c0 + c1;  rows (2,3): c2
        y[c][r][2 + bb] = prg ? -sum : cc[1][bb];        // rows (0,1): c1;       rows (2,3): -(c2 + c3)
      }
    }
  // ---- level 1: the workgroup's four partial blocks meet in wave 0 (in wave order)
  if (q > 0) {
#pragma unroll
    for (int i = 0; i < CT * 4; i++) red[q - 1][i][lane] = y[i >> 2][i & 3];
  }
  __syncthreads();
  if (q > 0) return;
#pragma unroll
  for (int ww = 0; ww < SMALL_WAVES - 1; ww++)
#pragma unroll
    for (int i = 0; i < CT * 4; i++) y[i >> 2][i & 3] += red[ww][i][lane];

  // ---- level 2: the S workgroups of a block meet through write-through slabs + one ticket per workgroup
  if (S > 1) {
    constexpr unsigned SLAB = CT * SMALL_SLAB_BYTES;
    const int block = tb16 * (int)gridDim.x + kqq;
    const auto rsrc_slab = make_rsrc(prm.slabs, (unsigned)((size_t)gridDim.x * gridDim.y * S * SLAB));
    const unsigned base = (unsigned)(block * S) * SLAB;
#pragma unroll
    for (int i = 0; i < CT * 4; i++)
      slab_store16(y[i >> 2][i & 3], rsrc_slab, (unsigned)((i * 64 + lane) * 16), base + (unsigned)split * SLAB);
    wait_vmem_all();   // the write-through stores have left ...
    unsigned old = 0;
    if (lane == 0)     // ... before the ticket
      old = __hip_atomic_fetch_add(prm.tickets + block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(S - 1)) {
      if (old >= (unsigned)S && lane == 0) __hip_atomic_store(prm.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      clk_exit();
      return;          // another workgroup finishes the block
    }
    if (lane == 0)     // self-cleaning counter (subtracted, not stored: see the 16 x 16 kernel)
      __hip_atomic_fetch_sub(prm.tickets + block, (unsigned)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // all S slabs, added in split order (bitwise reproducible whoever finishes); one tile's 4 registers at a time
    // keeps 4 S loads in flight
#pragma unroll
    for (int c = 0; c < CT; c++) {
      f32x4 part[SMALL_MAX_SPLIT][4];
#pragma unroll
      for (int s = 0; s < SMALL_MAX_SPLIT; s++) {
        if (s < S) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            part[s][r] = slab_load16(rsrc_slab, (unsigned)(((c * 4 + r) * 64 + lane) * 16), base + (unsigned)s * SLAB);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) y[c][r] = part[0][r];
#pragma unroll
      for (int s = 1; s < SMALL_MAX_SPLIT; s++) {
        if (s < S) {
#pragma unroll
          for (int r = 0; r < 4; r++) y[c][r] += part[s][r];
        }
      }
    }
  }

  // ---- finalize: BN + ReLU + 16-byte stores (and the tile's share of the zero ring)
  const int gt = tb16 * 16 + t16;
  if (gt >= totalTiles) { clk_exit(); return; }
  float* __restrict__ out = prm.out;
  const int relu = prm.relu;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const TileCoord tc = decode_tile(gt);
  const int oy = 1 + 2 * tc.ty, ox = 1 + 2 * tc.tx;
#pragma unroll
  for (int c = 0; c < CT; c++) {
    float* o = out + (size_t)tc.n * WINO_HW * WINO_HW * K + (kqq * CT + c) * 16 + 4 * h;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      f32x4 val = {y[c][0][p], y[c][1][p], y[c][2][p], y[c][3][p]};
      val = sc4[c] * val + bi4[c];
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; r++) val[r] = fmaxf(val[r], 0.f);
      }
      *(f32x4*)(o + (size_t)((oy + (p >> 1)) * WINO_HW + ox + (p & 1)) * K) = val;
    }
    // zero ring (the next 3x3 layer's padding, Kernel128_winograd.cu:163,243)
    if (tc.ty == 0) {
      *(f32x4*)(o + (size_t)(ox)*K) = zero4;
      *(f32x4*)(o + (size_t)(ox + 1) * K) = zero4;
      if (tc.tx == 0) *(f32x4*)(o) = zero4;
      if (tc.tx == 6) *(f32x4*)(o + (size_t)15 * K) = zero4;
    }
    if (tc.ty == 6) {
      *(f32x4*)(o + (size_t)(15 * WINO_HW + ox) * K) = zero4;
      *(f32x4*)(o + (size_t)(15 * WINO_HW + ox + 1) * K) = zero4;
      if (tc.tx == 0) *(f32x4*)(o + (size_t)(15 * WINO_HW) * K) = zero4;
      if (tc.tx == 6) *(f32x4*)(o + (size_t)(15 * WINO_HW + 15) * K) = zero4;
    }
    if (tc.tx == 0) {
      *(f32x4*)(o + (size_t)(oy * WINO_HW) * K) = zero4;
      *(f32x4*)(o + (size_t)((oy + 1) * WINO_HW) * K) = zero4;
    }
    if (tc.tx == 6) {
      *(f32x4*)(o + (size_t)(oy * WINO_HW + 15) * K) = zero4;
      *(f32x4*)(o + (size_t)((oy + 1) * WINO_HW + 15) * K) = zero4;
    }
  }
  clk_exit();
}

}  // namespace fused
}  // namespace wino
