// Latency-oriented variant of the fused Winograd F(2x2,3x3) kernel for SMALL batches
// (the reference's own operating point is N = 1: `./Test 0`, `./Test 1`,
// Kernel128_winograd.cu:263-265 / Kernel256_winograd.cu:266-268 at one image).
//
// At N = 1 a 256->256 layer has 49 tiles and 4.2 MB of filters: the work is tiny, what costs is (i) how
// many CUs pull the operands -- one CU takes in 50-70 GB/s of 16-byte loads, so a block's operands must be
// spread over many CUs -- and (ii) the latency chain launch -> loads -> MFMAs -> store.
//
// Work decomposition.  An output BLOCK is 16 tiles x 16 CT out-channels, CT = 1, 2 or 4 MFMA tiles held side by
// side by every wave (CT = 1 at one image; wider from a few images on, see below).  Its contraction runs over
// TASKS = (16-channel super-chunk, row group): two rows of the 4 x 4 point grid = 8 points = 3 of the 4 patch
// rows, 32 CT MFMAs.  A workgroup is 4 waves (one per SIMD); S workgroups share a block (gridDim.z = S,
// "C-split"): in ROUND r workgroup `split` takes the four tasks of super-chunks 2 (split + r S), + 1 -- its wave
// q the super-chunk + (q >> 1), row group q & 1.  S and CT are chosen on the host so that the grid just fills the
// CUs: 256 channels N = 1 is 64 blocks x S = 4, one round.  (Round 2's kernel had no C-split: 64 / 32 workgroups on
// 256 CUs, 18.8 / 15.1 us, bound by the load bandwidth of the few busy CUs.)
//
// Operands.  The form is bound by what a CU's vector memory path takes in, not by the MFMAs (8 passes each), and a
// PIXEL fragment loaded per wave in MFMA layout is the dear operand: its 64 lanes touch 16 cache lines (16 tiles x
// 64 B -- half of every line; the other half is the neighbouring super-chunk, which another wave of the same
// workgroup would load), against 8 whole lines for a FILTER fragment (measured, per wave-task with 4 waves per CU:
// 0.083 us against 0.052-0.075).  So
//   * the pixels are staged through LDS: the four waves of a workgroup -- same 16 tiles, 32 consecutive channels per
//     round -- fetch the round's patches TOGETHER, 16 tiles x 16 px x 32 channels = 256 whole 128-byte lines, 8
//     loads per wave of 8 whole lines each (a third of the line requests), write them to LDS (two 32 KB stages, one
//     barrier per round; the next round's loads fly behind this round's MFMAs) and read them back in fragment
//     layout.  Against per-wave pixel loads (the first cut of this kernel): 13-18 % faster at CT = 2, 4-10 % at
//     CT = 4, 5-9 % at CT = 1, never slower (profiles/r3/latency_ldsa.json, latency_ct1.json);
//   * wider blocks: every pixel fragment and its B^T d B feed CT MFMAs (per 64 MFMAs: 24 + 16 loads at CT = 1,
//     12 + 16 at CT = 2, 6 + 16 at CT = 4).  (Blocks of 32 TILES -- two pixel fragments per filter fragment -- were
//     built and measured too: 30-38 us where the wide ones take 21-28; not kept.);
//   * the filter fragments go straight from global memory to registers (lane (k = lane & 15, h) reads 16 B of the
//     packed filter: out-channel k, channels 4h .. 4h+3 of the super-chunk; any channel <-> MFMA-k assignment is
//     valid as long as both operands agree).  At CT = 4 the 128 accumulator registers leave no room for a second
//     buffer of them: the next task's points are requested PROGRESSIVELY, each out-channel block's behind that
//     block's MFMAs.
// LDS image of a stage: row (tile, px) of 128 B = 8 units of 16 B (unit u = channels 4u .. 4u+3 of the round's 32);
// row' = 16 tile + (px ^ (tile & 1)), unit' = u ^ ((tile >> 1) & 7): the 16 lanes of a ds_read_b128 group (tiles
// 0-3,12-15 of one lane group h and 4-11 of the next, MI355X_MICROARCH "LDS") then hit 64 distinct banks, and the 8
// lanes of a ds_write_b128 group write one whole row.
//
// Reduction, two levels, all on POST-transform values (A^T m A is linear, and a block's 2x2 outputs are
// 4 KB per MFMA tile where its 16 accumulator tiles are 16 KB):
//   1. each wave applies its part of A^T m A in-lane; waves 1..3 hand their 4 CT KB to wave 0 through LDS;
//   2. S > 1: wave 0 publishes the workgroup's partial block as a write-through slab (16-byte sc1 stores),
//      drains them, and draws ONE ticket on the block's counter; whoever draws the last ticket loads all S
//      slabs (4 S loads in flight), adds them in split order (bitwise reproducible whoever finishes),
//      applies BN + ReLU and stores.  Nobody waits for anybody -- the same slab / ticket rules as the
//      throughput kernel (wino_f2_fused_kernel.h).
// The MFMA's A operand is the filter fragment and its B operand the transformed pixels, so a lane ends up with
// four CONSECUTIVE out-channels of one tile: BN with four scales, 16-byte stores.
// Requires C % 16 == 0.  GEN = false is the reference's 14x14 map (geometry folded into the code); GEN = true carries the
// feature map in the arguments (ResNet's other stages): tile decode by multipliers, patch rows /
// columns of a clipped last tile row / column clamped into the tensor (a Winograd output depends on its own 3x3 window
// only, so the kept outputs are unaffected), surplus outputs not stored, the zero ring by a flat pass over the grid.
// Same arithmetic, same packed filter buffer and same output contract as the big kernel.
#pragma once
#include "wino_f2_fused_kernel.h"

namespace wino {
namespace fused {

constexpr int SMALL_WAVES = 4;          // waves per workgroup
constexpr int SMALL_MAX_SPLIT = 8;      // S <= 8: the finisher keeps 4 S sixteen-byte loads in flight
constexpr int SMALL_SLAB_BYTES = 4096;  // an MFMA tile's pre-BN 2x2 outputs: 16 tiles x 16 k x 4 px x 4 B

struct SmallParams {
  const float* in;
  const float* Uq;
  const float* bnBias;
  const float* bnScale;
  float* out;
  int N, C, K, relu;
  float* slabs;              // [block][S] x CT x 4 KB (S > 1 only)
  unsigned* tickets;         // [block]
  unsigned* err;             // host-visible word: set when a ticket counter was found dirty (S > 1 only)
  unsigned long long* dbg;   // timeline build only (DIAG, tools/small_timeline): 8 stamps per workgroup
  Geo geo;                   // GEN = true only: the feature map (other than the reference's 14 x 14)
};

// DIAG = true is the timeline build (tools/small_timeline.hip): wave 0 of every workgroup stores s_memrealtime
// (100 MHz, chip-wide) at entry, first stage in LDS, MFMAs done, LDS level done, slab drained, ticket drawn, gather
// landed, exit.  The product kernel is DIAG = false.
template <int CT, bool GEN = false, bool DIAG = false>
__global__ void __launch_bounds__(64 * SMALL_WAVES)
wino_f2_small_kernel(const SmallParams prm) {
  static_assert(CT == 1 || CT == 2 || CT == 4, "MFMA tiles per wave");
  constexpr int STAGE = 16 * 16 * 128;                   // one round's patches: 16 tiles x 16 px x 32 channels
  static_assert(2 * STAGE >= (SMALL_WAVES - 1) * CT * 4 * 64 * 16, "the reduction image reuses the stages");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const float* __restrict__ in = prm.in;
  const float* __restrict__ Uq = prm.Uq;
  const int N = prm.N, C = prm.C, K = prm.K;
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t16 = lane & 15, h = lane >> 4;
  const int tb16 = blockIdx.y, kqq = blockIdx.x;       // x = out-channel block: the workgroups sharing a filter slice share an XCD
  const int S = gridDim.z, split = blockIdx.z;
  auto mark = [&](int i) {
    if (DIAG && threadIdx.x == 0) {
      __builtin_amdgcn_sched_barrier(0);
      prm.dbg[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + i] = __builtin_amdgcn_s_memrealtime();   // (any order: the tool sorts)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  mark(0);
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
  const Geo geo = prm.geo;
  const int Hp = GEN ? geo.Hp : WINO_HW, Wp = GEN ? geo.Wp : WINO_HW;
  if constexpr (GEN) {
    // ring pass: the output's zero ring (the next 3x3 layer's padding) as a flat list of 16-byte units -- images x ring
    // pixels x K/4 units -- split over the grid (the 14x14 build writes each tile's share with the tile, below)
    const unsigned upp = (unsigned)K >> 2;
    const unsigned rpx = 2u * Wp + 2u * (Hp - 2);       // ring pixels per image
    const unsigned long long U = (unsigned long long)N * rpx * upp;
    const unsigned long long nblk = (unsigned long long)gridDim.x * gridDim.y * gridDim.z;
    const unsigned long long bid = ((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned u_begin = (unsigned)(U * bid / nblk), u_end = (unsigned)(U * (bid + 1ull) / nblk);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (unsigned u = u_begin + threadIdx.x; u < u_end; u += 64 * SMALL_WAVES) {
      const unsigned pid = u / upp, unit = u - pid * upp;
      const unsigned n = pid / rpx, qq = pid - n * rpx;
      const unsigned uW = (unsigned)Wp, uH = (unsigned)Hp;
      const unsigned y = qq < uW ? 0u : qq < 2 * uW ? uH - 1 : qq < 2 * uW + uH - 2 ? qq - 2 * uW + 1 : qq - 2 * uW - (uH - 2) + 1;
      const unsigned x = qq < uW ? qq : qq < 2 * uW ? qq - uW : qq < 2 * uW + uH - 2 ? 0u : uW - 1;
      *(f32x4*)(prm.out + ((size_t)(n * uH + y) * uW + x) * K + unit * 4) = zero4;
    }
  }
  const int totalTiles = N * (GEN ? (int)geo.tiles : WINO_TILES);
  const int KBLK = K >> 6;
  // Round r of workgroup `split` = tasks 4 (split + r S) + q, q = 0..3: super-chunks sc0 = 2 (split + r S) and sc0 + 1
  // (32 consecutive channels), row groups 0 and 1 of each.  Wave q: super-chunk sc0 + (q >> 1), row group q & 1.
  const int nsuper = C / 16, ntask = nsuper * 2;
  const int stride = SMALL_WAVES * S;
  const int nrounds = (ntask + stride - 1) / stride;
  const int p = q >> 1, prg = q & 1;
  const int slot0 = prg ? 2 : 0, slot1 = prg ? 1 : 2, slot2 = prg ? 3 : 1;
  const float sg = prg ? -1.f : 1.f;

  // ---- staging: wave q fetches tiles 4q .. 4q+3 of the block, 2 loads per tile (8 px x 8 units each)
  const int px_l = lane >> 3, u_l = lane & 7;
  const float* a_tile[4];
  unsigned w_off[4];                                    // LDS byte offset of this lane's unit in px half 0
  int e_off[GEN ? 4 : 1][2];                            // this lane's element offset inside a tile's patch, px half 0 / 1
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int T = 4 * q + j;
    int g = tb16 * 16 + T;
    g = g < totalTiles ? g : totalTiles - 1;
    const TileCoord tc = decode_tile_g<GEN>(g, geo);
    a_tile[j] = in + ((size_t)(tc.n * Hp + 2 * tc.ty) * Wp + 2 * tc.tx) * C;
    w_off[j] = (unsigned)((T * 16 + (px_l ^ (T & 1))) * 128 + ((u_l ^ ((T >> 1) & 7)) << 4));
    if constexpr (GEN) {
      // odd H or W: the last tile row / column reaches one row / column past the padded tensor -- clamp into it
      const int col = 2 * tc.tx + (px_l & 3), colc = (col < Wp ? col : Wp - 1) - 2 * tc.tx;
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const int row = 2 * tc.ty + 2 * hf + (px_l >> 2), rowc = (row < Hp ? row : Hp - 1) - 2 * tc.ty;
        e_off[j][hf] = (rowc * Wp + colc) * C;
      }
    }
  }
  if constexpr (!GEN) {   // (px = 8 half + px_l: row 2 half + (px_l >> 2))
    e_off[0][0] = ((px_l >> 2) * WINO_HW + (px_l & 3)) * C;
    e_off[0][1] = e_off[0][0] + 2 * WINO_HW * C;
  }
  f32x4 stg[8];
  auto load_a = [&](int r) {
    const int sc0 = 2 * (split + r * S);
    // a super-chunk past the end (odd nsuper; a workgroup without tasks in a ragged last round): re-read a valid one
    int sc = sc0 + (u_l >> 2);
    sc = sc < nsuper ? sc : (sc0 < nsuper ? sc0 : 0);
    const int c_off = sc * 16 + (u_l & 3) * 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      stg[2 * j] = *(const f32x4*)(a_tile[j] + e_off[GEN ? j : 0][0] + c_off);
      stg[2 * j + 1] = *(const f32x4*)(a_tile[j] + e_off[GEN ? j : 0][1] + c_off);
    }
  };
  auto store_a = [&](int stage) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      *(f32x4*)(base + w_off[j]) = stg[2 * j];
      *(f32x4*)(base + w_off[j] + 8 * 128) = stg[2 * j + 1];   // px + 8: eight rows on (px ^ 1 stays inside the half)
    }
  };
  // ---- fragment reads: lane (t16, h) = tile t16, unit 4 p + h; patch rows slot0/1/2, columns 0..3.  Column j of an odd
  // tile sits in row px ^ 1: columns (0, 2) at +A, +A + 256 and (1, 3) at +128 - A, +384 - A with A = 128 (t16 & 1)
  unsigned r_even[3], r_odd[3];
  {
    const unsigned lane_base = (unsigned)(t16 * 2048 + (((4 * p + h) ^ ((t16 >> 1) & 7)) << 4)), A = (unsigned)((t16 & 1) * 128);
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const unsigned row = (unsigned)((kk == 0 ? slot0 : kk == 1 ? slot1 : slot2) * 512);
      r_even[kk] = lane_base + row + A;
      r_odd[kk] = lane_base + row + 128 - A;
    }
  }

  const size_t b_chunk_stride = (size_t)KBLK * U_CHUNK_FLOATS;
  const float* b_src[CT];
#pragma unroll
  for (int c = 0; c < CT; c++) {
    const int k = (kqq * CT + c) * 16 + t16, kb = k >> 6, kl = k & 63;
    b_src[c] = Uq + (size_t)(h >> 1) * b_chunk_stride + ((size_t)kb * 16 * 64 + kl) * 8 + (((h & 1) ^ ((kl >> 3) & 1)) << 2) +
               (size_t)(2 * prg) * 4 * 512;   // the wave's first point
  }
  f32x4 acc[8][CT];
#pragma unroll
  for (int e = 0; e < 8; e++)
#pragma unroll
    for (int c = 0; c < CT; c++) acc[e][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 b[CT][8];
  auto load_b = [&](int t, int c) {
    const float* bp = b_src[c] + (size_t)(t >> 1) * 2 * b_chunk_stride;
#pragma unroll
    for (int e = 0; e < 8; e++) b[c][e] = *(const f32x4*)(bp + e * 512);
  };
  int t = SMALL_WAVES * split + q;
  load_a(0);
  if (t < ntask) {
#pragma unroll
    for (int c = 0; c < CT; c++) load_b(t, c);
  }
  __builtin_amdgcn_sched_barrier(0);
  store_a(0);
  __syncthreads();
  mark(1);
#pragma unroll 1
  for (int r = 0; r < nrounds; r++, t += stride) {
    const bool more_a = r + 1 < nrounds;
    const bool more_b = t + stride < ntask;
    if (t < ntask) {
      const char* base = smem + (r & 1) * STAGE;
      f32x4 d[12], tmp[8], v[8];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        d[kk * 4 + 0] = *(const f32x4*)(base + r_even[kk]);
        d[kk * 4 + 1] = *(const f32x4*)(base + r_odd[kk]);
        d[kk * 4 + 2] = *(const f32x4*)(base + r_even[kk] + 256);
        d[kk * 4 + 3] = *(const f32x4*)(base + r_odd[kk] + 256);
      }
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
      if (more_a) load_a(r + 1);             // the next round's patches, in flight behind this round's MFMAs (requested
      __builtin_amdgcn_sched_barrier(0);     // only now: before the transform their 32 staging registers would not fit)
#pragma unroll
      for (int c = 0; c < CT; c++) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
          for (int e = 0; e < 8; e++)
            acc[e][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[c][e][jj], v[e][jj], acc[e][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more_b) load_b(t + stride, c);     // this out-channel block's points of the next task, behind its MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    else if (more_a) load_a(r + 1);          // (a wave without a task this round still stages its share)
    if (more_a) store_a((r + 1) & 1);        // (stage (r+1)&1 was last read in round r-1: every wave has passed that round's barrier)
    __syncthreads();
  }

  // folded BN of this lane's out-channels (kqq*CT + c)*16 + 4h .. + 3: requested now (in the loop its 8 CT registers
  // were the ones that spilled at CT = 4), used by the finisher behind the two reduction levels
  f32x4 sc4[CT], bi4[CT];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      sc4[c][r] = prm.bnScale[(kqq * CT + c) * 16 + 4 * h + r];
      bi4[c][r] = prm.bnBias[(kqq * CT + c) * 16 + 4 * h + r];
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
  if (DIAG) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  mark(2);
  // ---- level 1: the workgroup's four partial blocks meet in wave 0 (in wave order); the image reuses the stages
  // (the loop's last barrier is behind every wave's last fragment read)
  f32x4 (*red)[CT * 4][64] = (f32x4(*)[CT * 4][64])smem;
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

  mark(3);
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
    mark(4);
    unsigned old = 0;
    if (lane == 0)     // ... before the ticket
      old = __hip_atomic_fetch_add(prm.tickets + block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    mark(5);
    if (old != (unsigned)(S - 1)) {
      if (old >= (unsigned)S && lane == 0) __hip_atomic_store(prm.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      clk_exit();
      return;          // another workgroup finishes the block
    }
    if (lane == 0)     // self-cleaning counter (subtracted, not stored: see the 16 x 16 kernel)
      __hip_atomic_fetch_sub(prm.tickets + block, (unsigned)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

  if (DIAG && S > 1) { wait_vmem_all(); mark(6); }
  // ---- finalize: BN + ReLU + 16-byte stores (and the tile's share of the zero ring)
  const int gt = tb16 * 16 + t16;
  if (gt >= totalTiles) { clk_exit(); return; }
  float* __restrict__ out = prm.out;
  const int relu = prm.relu;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const TileCoord tc = decode_tile_g<GEN>(gt, geo);
  const int oy = 1 + 2 * tc.ty, ox = 1 + 2 * tc.tx;
#pragma unroll
  for (int c = 0; c < CT; c++) {
    float* o = out + (size_t)tc.n * Hp * Wp * K + (kqq * CT + c) * 16 + 4 * h;
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
      f32x4 val = {y[c][0][pp], y[c][1][pp], y[c][2][pp], y[c][3][pp]};
      val = sc4[c] * val + bi4[c];
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; r++) val[r] = fmaxf(val[r], 0.f);
      }
      // (odd H or W: the last tile row / column computes one output row / column too many; it must not reach the ring)
      if (!GEN || (oy + (pp >> 1) <= Hp - 2 && ox + (pp & 1) <= Wp - 2))
        *(f32x4*)(o + (size_t)((oy + (pp >> 1)) * Wp + ox + (pp & 1)) * K) = val;
    }
    if constexpr (GEN) continue;   // (the ring pass at the top wrote the ring)
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

}  // namespace fused
}  // namespace wino
