// The fused Winograd F(2x2,3x3) kernel, as a header so that the library
// (wino_f2_fused.hip) and the ablation tool (tools/ablate_fused.hip) compile the same source.
//
// ABLATE is a debug knob (0 = the product kernel); non-zero values produce wrong results and
// exist only to price the parts of the kernel:
//   1 skip the raw-patch LDS-DMA      2 skip the filter LDS-DMA      4 skip the MFMAs
//   8 skip the per-chunk wait+barrier 16 stamp the main loop (in-kernel clock)
//   32 skip the A-path reads+transform 64 skip the B-fragment reads  512 skip the output stores
//   1024 skip the stream-K slab hand-off (partial segments are dropped)
//   32768 timeline (tools/ablate_fused ... t): s_memrealtime (100 MHz, chip-wide) at kernel entry, first MFMA,
//         start of the last epilogue and exit of every workgroup, 8 uint64 per workgroup in prm.dbg
//
// Work decomposition: the launch's work is a set of "chunk iterations"
//   (item, chunk),  item = (tile block, k block),  chunk = 8 input channels,
// nTB * K/64 * C/8 of them, all of equal cost.  G logical workgroups (at most one per CU) share
// them in two phases:
//   * whole-item rounds: ndp = items / G rounds in which workgroup l owns item r*G + l outright.
//     All workgroups then walk the same chunk index at the same time, so the K/64 k-blocks that
//     share a tile block's patches and the tile blocks that share a filter chunk hit each other's
//     lines in the XCD's L2 (about 80 % hit rate over the whole launch, profiles/);
//   * a stream-K tail for the remaining items % G items: their chunk iterations, in item-major
//     order, are cut into G equal contiguous ranges.  At 256 channels and N = 128: 392 items on
//     256 CUs = one whole item (32 iterations) + 17 tail iterations each = 49 per CU, instead of
//     the 64 a second whole-item round would cost.  (All 392 items as one stream-K range were
//     measured too: the scattered channel phases halve the L2 hit rate and quadruple HBM traffic.)
// A tail range cuts an item into at most two partial SEGMENTS per workgroup (the head and the tail
// of its range).  A partial segment applies A^T m A to its partial sums (the inverse transform is
// linear), publishes the 64 KB pre-BN result as a write-through slab and draws a ticket on the
// item's counter; the wave that draws the last ticket adds the other segments' slabs (in segment
// order, whoever is last: the result is bitwise reproducible), applies BN + ReLU and stores.
// Nobody ever waits on another workgroup, so there is no residency assumption and no deadlock; the
// hand-off follows the write-through recipe (sc1 stores, the storing wave drains vmcnt, one relaxed
// agent-scope ticket add, sc1 loads by the reducer).  The tail runs FIRST: its hand-offs then
// complete in the middle of the launch and every workgroup ends with a whole item.
#pragma once
#include "wino_common.h"

#include <type_traits>

namespace wino {
namespace fused {

constexpr int TB = 64;                       // tiles per workgroup
constexpr int KB = 64;                       // out-channels per workgroup
constexpr int BC = 8;                        // in-channels per pipeline stage
constexpr int NTHREADS = 512;
constexpr int RAW_BYTES = TB * 16 * BC * 4;  // 32768: one raw-patch stage
constexpr int U_BYTES = 16 * KB * BC * 4;    // 32768: one filter stage
constexpr int N_RSTAGE = 2, N_USTAGE = 3;
constexpr int LDS_BYTES = N_RSTAGE * RAW_BYTES + N_USTAGE * U_BYTES;  // 163840 = all 160 KiB of the CU
constexpr int U_CHUNK_FLOATS = 16 * KB * BC; // 8192 floats per (c-chunk, k-block)
constexpr int PF = 2;                        // filter-fragment prefetch distance (points); 2..6 measured equal
constexpr int SLAB_BYTES = TB * 4 * KB * 4;  // 65536: pre-BN output of one item (64 tiles x 2x2 px x 64 k)
#ifndef WINO_UNROLL2
#define WINO_UNROLL2 1   // two copies of the loop body, one per raw-stage parity: the stage is an immediate of the patch reads
#endif
#ifndef WINO_SCALAR_BTDB
#define WINO_SCALAR_BTDB 0
#endif
#ifndef WINO_DMA0
#define WINO_DMA0 4   // tools/ablate_fused, current loop: 0 / 2 / 4 / 6 / 8 give 42.7 / 42.5 / 42.45 / 42.5 / 42.7 cycles per MFMA
#endif
constexpr int DMA0 = WINO_DMA0;              // first point-step that issues an LDS-DMA piece

// s_waitcnt lgkmcnt(n) alone (vmcnt/expcnt fields at "no wait"); n folds to a literal once the
// point loop is unrolled.
__device__ __forceinline__ void wait_lds(int n) {
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
// LDS requests issued at the top of pinned step q of the point loop (see the kernel): always 2
// filter-fragment reads (step q+2 of this chunk, or step q-14 of the NEXT chunk on steps
// 14, 15), then 2 patch reads while q < 6 (a wave reads 12 of its tiles' 16 patch pixels).
constexpr int lds_nr(int q) { return q < 6 ? 2 : 0; }
constexpr int lds_n(int q) { return 2 + lds_nr(q); }
// How many LDS requests are younger than the last one step e's consumers need: the filter
// fragments of point e (requested two steps earlier; for e < 2 before the barrier, which
// drains lgkmcnt) and, on steps 2,4,6,8, the patch pixels requested at steps e-2, e-1.
constexpr int lds_wait_count(int e) {
  // the fragments of point e were requested at step e-PF, before that step's patch reads
  int after = 15;
  if (e >= PF) {
    after = lds_nr(e - PF);
    for (int q = e - PF + 1; q <= e; q++) after += lds_n(q);
  }
  return after > 15 ? 15 : after;
}

// 8-byte LDS read at an absolute LDS byte address held in a register.  (`smem + offset` leaves a
// `v_add_u32 v, 0, v` per read in the loop -- the dynamic-LDS base is a symbol the optimizer does not
// fold -- so the fragment address registers carry the base themselves and are used as pointers.)
__device__ __forceinline__ f32x2 lds_read2(int addr) {
  return *(const __attribute__((address_space(3))) f32x2*)(size_t)(unsigned)addr;   // (size_t: the host pass parses this with 64-bit pointers)
}
struct TileCoord {
  int n, ty, tx;
};
__device__ __forceinline__ TileCoord decode_tile(int g) {
  TileCoord t;
  t.n = g / WINO_TILES;
  const int rem = g - t.n * WINO_TILES;
  t.ty = rem / 7;
  t.tx = rem - t.ty * 7;
  return t;
}

// Feature-map geometry of a launch: H x W outputs = ceil(H/2) x ceil(W/2) tiles per image, inside
// padded (H+2) x (W+2) tensors.  The reference's 14x14 stage is a compile-time specialisation
// (GEN = false: the divisions by 49 and 7 fold into multiplies); other sizes (ResNet's 56x56 and
// 28x28 stages) carry the numbers in the kernel arguments, with the two
// divisors as Granlund-Montgomery multipliers (FastDiv, wino_common.h).
struct Geo {
  int Hp, Wp;              // padded extents
  unsigned tiles, tiles_x;  // tiles per image, per tile row
  FastDiv d_tiles, d_tx;
};
template <bool GEN>
__device__ __forceinline__ TileCoord decode_tile_g(int g, const Geo& geo) {
  if (!GEN) return decode_tile(g);
  TileCoord t;
  t.n = (int)fastdiv((unsigned)g, geo.d_tiles);
  const unsigned rem = (unsigned)g - (unsigned)t.n * geo.tiles;
  t.ty = (int)fastdiv(rem, geo.d_tx);
  t.tx = (int)(rem - (unsigned)t.ty * geo.tiles_x);
  return t;
}

// Stream-K bookkeeping shared by the kernel and the host (T < 2^31, G <= 65535): logical
// workgroup l of G owns the iterations [sk_start(l), sk_start(l+1)) of T = q*G + rem, i.e.
// floor(l*T/G) = l*q + floor(l*rem/G) in 32-bit arithmetic.
__host__ __device__ inline unsigned sk_start(unsigned l, unsigned q, unsigned rem, unsigned G) {
  return l * q + l * rem / G;
}

// Which of a k-group's Gp tail ranges the workgroup at position j (consecutive positions share an XCD) runs.  Ranges
// of equal length q start at channel phase (q * range) mod nchunks, which repeats with period P = nchunks / gcd(q,
// nchunks); ranked by phase, rank r belongs to the ranges a + P * b with a = inv * r mod P.  Position j = r * copies +
// b, so that the workgroups of one XCD hold ranges of CONSECUTIVE phases: a filter chunk (k-block, channel chunk) one
// of them has fetched is what its neighbours ask for one iteration later -- an L2 hit instead of a fetch per
// workgroup.  P = 1, inv = 0, copies = Gp is the identity (unequal ranges, or a period that does not divide Gp).
__host__ __device__ inline int tail_range_of(int j, int P, int inv, int copies) {
  const int r = j / copies, b = j - r * copies;
  return (inv * r) % P + P * b;
}
__device__ __forceinline__ int tail_range_of_fast(int j, int P, int inv, int copies, FastDiv d_P, FastDiv d_copies) {
  const int r = (int)fastdiv((unsigned)j, d_copies), b = j - r * copies;
  const int ir = inv * r;
  return ir - (int)fastdiv((unsigned)ir, d_P) * P + P * b;
}

// The kernel's only argument.  The fields below the line are used by the epilogue alone: it
// re-reads them from the kernarg segment each time instead of keeping ~20 scalar registers
// (pointers + two buffer descriptors) alive across the main loop, which is out of SGPRs.
struct FusedParams {
  const float* in;
  const float* Uq;
  int N, C, K, relu, nTB;
  int ndp;                     // whole-item rounds: items / gridDim.x
  unsigned sk_q, sk_rem;       // tail, per k-group: (items % gridDim.x) / kp * C/8 = sk_q * (gridDim.x / kp) + sk_rem iterations
  int kp;                      // k-groups of the tail: K/64 (gridDim.x a multiple of it) or 1
  int Gp;                      // gridDim.x / kp: ranges (= positions) per k-group
  int ph_P, ph_inv, ph_copies; // which tail range the j-th position of a group runs (tail_range_of below)
  FastDiv d_kp, d_P, d_copies; // the three divisors as multipliers: a runtime integer division is ~40 instructions on
                               // this machine, and the first cut of the k-groups paid five of them in the prologue and
                               // five in every epilogue (+0.7 us on a 40 us launch)
  Geo geo;                     // feature-map geometry (read by the GEN = true build only)
  // ---- epilogue only ----
  const float* bnBias;
  const float* bnScale;
  float* out;
  float* slabs;
  unsigned* tickets;
  unsigned* err;               // host-visible word, set when a ticket is drawn on a counter that was not zero at launch
  unsigned long long* dbg;     // diagnostic builds only (ABLATE & (16 | 2048)): where the stamps go
};

// The clock the chip holds inside the product kernel: workgroup 0 of every launch stores {s_memtime,
// s_memrealtime} here at its entry and at its exit (four 8-byte stores per launch from one lane; nothing
// reads them on the device).  wino_diag_last_clock() copies them out: bench.py takes the clock OF its timed
// launches from the last one of the burst instead of from a separate stamped build run afterwards.
__device__ unsigned long long wino_clk_slot_3x3[4];

// TAIL = the launch has a stream-K tail (any partial segment at all).  Launches of whole items only -- every
// shape whose items fit or divide the grid -- compile the hand-off out of the epilogue: 128 channels N = 128
// 40.1 -> 39.6 us, 256 channels N = 64 71.7 -> 71.0.  (Tried for the launches WITH a tail in the same step: the
// gather with two slabs in flight and the accumulators re-zeroed at the end of the epilogue to make room -- no
// gain at small batches, where an item has up to 8 segments, +0.8..1.2 % at 256 channels N = 96 / 256.)
template <int ABLATE, bool GEN = false, bool TAIL = true>
__global__ void __launch_bounds__(NTHREADS, 2)
wino_f2_fused_kernel(const FusedParams prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const float* in = prm.in;
  const float* Uq = prm.Uq;
  asm volatile("" : "+s"(in), "+s"(Uq));
  const int N = prm.N, C = prm.C, K = prm.K;
  const unsigned sk_q = prm.sk_q, sk_rem = prm.sk_rem;
  const int ndp = prm.ndp;

  // XCD-aware block -> logical workgroup: blocks b and b+8 share an XCD (its L2), so consecutive
  // logical workgroups -- which walk consecutive items, i.e. the K/64 k-blocks that read the same
  // input tiles -- are placed on the same XCD: XCD x = blockIdx % 8 gets the logical range that
  // starts at x*(G/8) + min(x, G%8).
  const int KBLK = K >> 6;
  const int nchunks = C / BC;
  const int G = gridDim.x;
  const int lg = (int)(blockIdx.x & 7) * (G >> 3) + ((int)(blockIdx.x & 7) < (G & 7) ? (int)(blockIdx.x & 7) : (G & 7)) +
                 (int)(blockIdx.x >> 3);
  if (ABLATE & 32768) {
    if (threadIdx.x == 0) {
      prm.dbg[(size_t)lg * 8 + 0] = __builtin_amdgcn_s_memrealtime();
      prm.dbg[(size_t)lg * 8 + 4] = __builtin_amdgcn_s_memtime();
    }
  }
  if (ABLATE == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    wino_clk_slot_3x3[0] = __builtin_amdgcn_s_memtime();
    wino_clk_slot_3x3[1] = __builtin_amdgcn_s_memrealtime();
  }
  auto clk_exit = [&]() {
    if (ABLATE == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
      wino_clk_slot_3x3[2] = __builtin_amdgcn_s_memtime();
      wino_clk_slot_3x3[3] = __builtin_amdgcn_s_memrealtime();
    }
  };
  // tail: items ndp*G .. ; then the rounds.  The tail is cut PER OUT-CHANNEL BLOCK (kp = K/64 groups when the grid
  // is a multiple of it): group k = lg % kp owns the tail items of k-block k (one per tile block, walked in
  // tile-block order: item tail_item0 + kp*j + k), (sk_q * Gp + sk_rem) chunk iterations cut into Gp = G / kp equal
  // ranges, range lp = lg / kp.  The kp workgroups with the same lp -- neighbours on one XCD -- then walk the SAME
  // tile blocks and channel chunks at the same time and share every patch line in the XCD's L2, exactly as the
  // whole-item rounds do.  (Round 2 cut one item-major list into G ranges: the k-blocks of a tile block sat at
  // different channel phases, every one of them fetched the patches for itself: HBM-side fetch 164 MB per launch
  // at the reference's 256 channels, N = 128, against 38 MB compulsory.)  kp = 1 is that old scheme.
  const int tail_item0 = ndp * G;
  const int kpg = prm.kp, Gp = prm.Gp;
  const int lpos = __builtin_amdgcn_readfirstlane((int)fastdiv((unsigned)lg, prm.d_kp)), grp = lg - lpos * kpg;
  const int lp = __builtin_amdgcn_readfirstlane(tail_range_of_fast(lpos, prm.ph_P, prm.ph_inv, prm.ph_copies, prm.d_P, prm.d_copies));   // the group's tail range this workgroup runs
  const unsigned t_begin = __builtin_amdgcn_readfirstlane(sk_start(lp, sk_q, sk_rem, Gp));
  const int Lt = (int)(__builtin_amdgcn_readfirstlane(sk_start(lp + 1, sk_q, sk_rem, Gp)) - t_begin);   // tail iterations of this workgroup
  const int L = Lt + ndp * nchunks;                                     // all its chunk iterations

  // ---- ring pass: the output's zero ring (the next 3x3 layer's padding, Kernel128_winograd.cu:
  // 163,243) is written here, once per launch, as a flat list of 16-byte units -- N images x 60
  // ring pixels x K/4 units -- split evenly over the workgroups; the stores are fully coalesced
  // and drain while the first LDS-DMA pieces are in flight.  (Per tile in the epilogue, the ring
  // cost more than the tiles' own stores: sparse predicated stores and their address arithmetic.)
  auto ring_pass = [&]() {
    if (ABLATE & 512) return;
    const unsigned Hp = GEN ? (unsigned)prm.geo.Hp : (unsigned)WINO_HW, Wp = GEN ? (unsigned)prm.geo.Wp : (unsigned)WINO_HW;
    const unsigned rpx = 2u * Wp + 2u * (Hp - 2u);               // ring pixels per image (60 at 16 x 16)
    const unsigned upp = (unsigned)K >> 2;                       // units per ring pixel
    const unsigned long long U = (unsigned long long)N * rpx * upp;
    const unsigned u_begin = (unsigned)(U * (unsigned)lg / (unsigned)G);
    const unsigned u_end = (unsigned)(U * ((unsigned)lg + 1u) / (unsigned)G);
    const auto rsrc_ring = make_rsrc(prm.out, (unsigned)((size_t)N * Hp * Wp * K * sizeof(float)));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (unsigned u = u_begin + threadIdx.x; u < u_end; u += NTHREADS) {
      const unsigned pid = u / upp, unit = u - pid * upp;
      const unsigned n = pid / rpx, q = pid - n * rpx;
      // q: [0, Wp) row 0, [Wp, 2Wp) the last row, then column 0 and the last column of rows 1..Hp-2
      const unsigned y = q < Wp ? 0u : q < 2 * Wp ? Hp - 1 : q < 2 * Wp + Hp - 2 ? q - 2 * Wp + 1 : q - 2 * Wp - (Hp - 2) + 1;
      const unsigned x = q < Wp ? q : q < 2 * Wp ? q - Wp : q < 2 * Wp + Hp - 2 ? 0u : Wp - 1;
      buf_store16(zero4, rsrc_ring, (((n * Hp + y) * Wp + x) * (unsigned)K + unit * 4u) * (unsigned)sizeof(float), 0);
    }
  };
  if (L <= 0) {   // more workgroups than iterations: this one only has its share of the ring
    ring_pass();
    clk_exit();
    return;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Point split: wave (wt, ph) owns 16 tiles x ALL 64 out-channels x 8 of the 16 Winograd points -- rows
  // i = 2 ph, 2 ph + 1 of the 4 x 4 point grid.  The two waves of a tile block then share no B^T d B
  // work (with 16 points per wave both computed the same V): a wave needs 3 of the 4 patch rows (12
  // reads) and 16 packed adds per iteration instead of 16 reads and 32 adds.  Timing-only ablation of
  // exactly that (half the adds, 12 reads): 124.3 -> 116.7 us at 256 channels, 41.9 -> 39.3 at 128.
  // The price is an exchange of partial A^T m A sums between the pair in the epilogue, after which
  // wave (wt, ph) owns out-channels [32 ph, 32 ph + 32) of its 16 tiles exactly as wave (wt, wk) did.
  const int wt = w >> 1;  // which 16-tile block of the 64
  const int ph = w & 1;   // which half of the point rows; after the epilogue exchange: which 32-channel half

  const unsigned u_off = lane * 16;
  const unsigned u_chunk_stride = (unsigned)(KBLK * U_CHUNK_FLOATS * sizeof(float));
  // buffer descriptors (wave-uniform): everything loop-variant goes into the scalar offset
  const auto rsrc_in = make_rsrc(in, (unsigned)((size_t)N * (GEN ? prm.geo.Hp * prm.geo.Wp : WINO_HW * WINO_HW) * C * sizeof(float)));
  const auto rsrc_u = make_rsrc(Uq, (unsigned)((size_t)16 * C * K * sizeof(float)));

  // ---- fragment read addresses (launch invariant) -------------------------------
  const int t16 = lane & 15, h = lane >> 4;
  // A: tile row tl = wt*16 + t16; 8-byte quarter h holds channels 2h, 2h+1 of the chunk
  const int lds0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // absolute LDS address of smem[0]
  const int a_base = lds0 + (wt * 16 + t16) * 512 + ((h ^ (((t16 >> 3) & 1) << 1)) << 3);
  const int a_sw = t16 & 7;
  // Fragment reads must stay plain ds_read_b64: that form banks on 64 dwords, for which the
  // XOR layouts are conflict-free.  hipcc would fuse two reads off one base register into
  // ds_read2_b64 / ds_read2st64_b64, which bank on 32 dwords (2-way conflicts here, half the
  // bytes per clock).  Hiding how the bases relate (empty asm) prevents the fusion; pixels px
  // and px+8 share a base register but are never read in the same pinned step.
  // The three patch rows R0, R1, R2 this wave reads are (d0, d2, d1) for ph = 0 and (d2, d1, d3) for
  // ph = 1: then B^T d, rows 2 ph and 2 ph + 1, is  R0 - R1  and  R1 + s2 * R2  with s2 = +1 / -1 for
  // both halves (one code path).  One address register per pixel (12; the raw stage is selected by
  // bit 15 of each, flipped after every iteration).
  int a_adr[3][4];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int row = ph ? (k == 0 ? 2 : k == 1 ? 1 : 3) : (k == 0 ? 0 : k == 1 ? 2 : 1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int px = 4 * row + j;
      a_adr[k][j] = a_base + (((px & 7) ^ a_sw) << 5) + ((px >> 3) << 8);
    }
  }
#define A_OFF(k, j) (a_adr[k][j])
  // B: logical column block cb' of this wave is out-channel block cb' ^ 2 ph of the item (k_local =
  // (cb' ^ 2 ph) * 16 + t16): cb' = 0, 1 are the two blocks the wave keeps after the epilogue exchange,
  // cb' = 2, 3 the two it hands to its partner -- compile-time indices either way.  The byte offset of
  // the wave's first point (8 ph) inside a filter stage is folded in.
  int b_base[4];
#pragma unroll
  for (int cb = 0; cb < 4; cb++) {
    const int kl = (cb ^ (2 * ph)) * 16 + t16;
    b_base[cb] = lds0 + N_RSTAGE * RAW_BYTES + ph * (8 * 2048) + kl * 32 + ((h ^ (((kl >> 3) & 1) << 1)) << 3);
  }
  asm volatile("" : "+v"(b_base[1]), "+v"(b_base[2]), "+v"(b_base[3]));

  // B^T d B pieces (d, tmp, v are [row i][col j] = index 4i + j; Winograd point e = 4i + j), on
  // channel pairs (v_pk_add_f32)
  typedef f32x2 P2;
#if WINO_SCALAR_BTDB
  // experiment: plain v_sub_f32 / v_add_f32 pairs instead of v_pk_add_f32 (asm: the SLP vectoriser
  // would fuse plain C++ back into packed ops)
  auto sub2 = [](const P2& a, const P2& b) {
    P2 r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.x));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.y));
    return r;
  };
  auto add2 = [](const P2& a, const P2& b) {
    P2 r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.x));
    asm("v_add_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.y));
    return r;
  };
#else
  auto sub2 = [](const P2& a, const P2& b) { return a - b; };
  auto add2 = [](const P2& a, const P2& b) { return a + b; };
#endif
  const float s2f = ph ? -1.f : 1.f;
  const P2 s2 = {s2f, s2f};
  // d[k*4 + j] = patch row R_k, column j;  tmp[i'*4 + j] = (B^T d) row 2 ph + i', column j;
  // v[p], p = 4 i' + j = the wave's point p (Winograd point e = 8 ph + p)
  auto tmp_col = [&](P2* tmp, const P2* d, int j) {  // B^T d, column j, the wave's two rows
    tmp[0 * 4 + j] = sub2(d[0 * 4 + j], d[1 * 4 + j]);
    tmp[1 * 4 + j] = d[1 * 4 + j] + s2 * d[2 * 4 + j];
  };
  auto v_point = [&](P2* v, const P2* tmp, int e) {  // (B^T d) B, point e (0..7)
    const int i = e >> 2, j = e & 3;
    if (j == 0) v[e] = sub2(tmp[i * 4 + 0], tmp[i * 4 + 2]);
    if (j == 1) v[e] = add2(tmp[i * 4 + 1], tmp[i * 4 + 2]);
    if (j == 2) v[e] = sub2(tmp[i * 4 + 2], tmp[i * 4 + 1]);
    if (j == 3) v[e] = sub2(tmp[i * 4 + 1], tmp[i * 4 + 3]);
  };

  unsigned long long st_wait = 0, st_comp = 0, st_epi = 0, st_prev = 0;   // ABLATE & 2048: phase stamps
  unsigned long long st_ph[4] = {0, 0, 0, 0};   // epilogue phases: barrier, A^T m A, slab+ticket, gather+finalize
  auto stamp = [&]() -> unsigned long long {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };

  // Opaque pointer to the kernel's own arguments (constant address space: scalar loads).  The rare
  // paths below (tile-block switch of the DMA stream, epilogue) re-read what they need through it
  // instead of keeping those scalars alive across the main loop, which is out of SGPRs.
  typedef const __attribute__((address_space(4))) FusedParams* KernargPtr;
  auto kernarg = []() {
    KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
  };
  auto load_geo = [](KernargPtr kp) {   // member-wise: the struct lives in the constant address space
    Geo g;
    g.Hp = kp->geo.Hp; g.Wp = kp->geo.Wp; g.tiles = kp->geo.tiles; g.tiles_x = kp->geo.tiles_x;
    g.d_tiles.m = kp->geo.d_tiles.m; g.d_tiles.l = kp->geo.d_tiles.l;
    g.d_tx.m = kp->geo.d_tx.m; g.d_tx.l = kp->geo.d_tx.l;
    return g;
  };

  // ---- the DMA stream: walks (item, chunk) linearly, two iterations ahead of the MFMAs ------
  // raw stage layout: [tile 0..63][unit' 0..31] of 16 B; unit' = px'*2 + half'.
  // LDS unit (t, px', half') holds pixel px = px' ^ (t&7), channel half = half' ^ bit3(t).
  // wave-instruction q = 8*j + w (j = 0..3) covers tiles 2q, 2q+1.
  // (byte offsets from `in`, 32-bit: the DMA then uses the scalar-base + vector-offset address
  //  form and advancing to the next chunk is a scalar add instead of a 64-bit VALU add per piece)
  unsigned raw_off[4];        // per-lane source offsets of the DMA stream's tile block
  unsigned d_soff_raw = 0;    // + chunk * 32 B
  unsigned d_soff_u = 0;      // filter chunk of (k block, chunk), this wave's 1-KiB column
  int d_item, d_chunk, d_tail, d_tb = -1;   // d_tail: tail iterations the stream still has to issue
  auto dma_set_item = [&](int item) {   // wave-uniform; the per-lane part only when the tile block changes
    const int tb = item / KBLK, kb = item - tb * KBLK;
    d_item = item;
    if (tb != d_tb) {
      d_tb = tb;
      KernargPtr kp = kernarg();
      const Geo geo = load_geo(kp);
      const int Wp = GEN ? geo.Wp : WINO_HW, Hp = GEN ? geo.Hp : WINO_HW;
      const int C = kp->C, totalTiles = kp->N * (GEN ? (int)geo.tiles : WINO_TILES);
      const int up = lane & 31;
      const int pxp = up >> 1, halfp = up & 1;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int tl = 16 * j + 2 * w + (lane >> 5);
        const int px = pxp ^ (tl & 7);
        const int half = halfp ^ ((tl >> 3) & 1);
        int g = tb * TB + tl;
        g = g < totalTiles ? g : totalTiles - 1;  // clamp: padded rows read a valid tile
        const TileCoord tc = decode_tile_g<GEN>(g, geo);
        const int y = 2 * tc.ty + (px >> 2), x = 2 * tc.tx + (px & 3);
        raw_off[j] = (unsigned)((((size_t)(tc.n * Hp + y) * Wp + x) * C + half * 4) * sizeof(float));
      }
    }
    d_soff_raw = __builtin_amdgcn_readfirstlane((unsigned)(d_chunk * (BC * sizeof(float))));
    d_soff_u = __builtin_amdgcn_readfirstlane((unsigned)((kb * U_CHUNK_FLOATS + w * 256) * sizeof(float)) + d_chunk * u_chunk_stride);
    // ("+s": pins the loop-carried scalars to SGPRs.  Short of SGPRs, the compiler otherwise keeps
    //  one of them in a VGPR and wraps every LDS-DMA that uses it in a waterfall loop.)
    asm volatile("" : "+s"(d_soff_raw), "+s"(d_soff_u));
  };
  auto pin_dma_state = [&]() {   // "+s": keep the loop-carried scalars in SGPRs (see dma_set_item)
    d_item = __builtin_amdgcn_readfirstlane(d_item);
    d_chunk = __builtin_amdgcn_readfirstlane(d_chunk);
    d_tail = __builtin_amdgcn_readfirstlane(d_tail);
    d_tb = __builtin_amdgcn_readfirstlane(d_tb);
    asm volatile("" : "+s"(d_item), "+s"(d_chunk), "+s"(d_tail), "+s"(d_tb));
  };
  auto dma_advance = [&]() {
    if (d_tail > 0 && --d_tail == 0) {          // the tail range is issued: on to the whole items
      d_chunk = 0;
      dma_set_item(lg);
    } else if (++d_chunk == nchunks) {
      d_chunk = 0;
      dma_set_item(d_tail > 0 ? d_item + kernarg()->kp : d_item + G);   // the next tail item of this k-group / round
    } else {
      d_soff_raw += (unsigned)(BC * sizeof(float));
      d_soff_u += u_chunk_stride;
      asm volatile("" : "+s"(d_soff_raw), "+s"(d_soff_u));
    }
    pin_dma_state();
  };
  // LDS map: [R0 32K][R1 32K][U0 32K][U1 32K][U2 32K]; R = raw 4x4 patches, U = filter chunk.
  // Both DMA streams run TWO iterations ahead of the MFMAs, across item boundaries:
  //  * raw_{i+1} is consumed one iteration early: while the MFMAs of iteration i run (operands:
  //    V_i in registers, U_i in LDS) the wave reads raw_{i+1} from LDS and transforms it into
  //    V_{i+1}, in place over the V registers of points that have already retired, so the
  //    matrix pipe never waits for the input transform (2 raw stages suffice);
  //  * U_{i+1} is already visible during iteration i (3 filter stages), so the first filter
  //    fragments of iteration i+1 are requested BEFORE the iteration barrier and the MFMAs
  //    resume right after it instead of eating an LDS round trip.
  auto issue_raw1 = [&](int rstage, int j) {  // one 1-KiB piece
    if (ABLATE & 1) return;
    dma16_buf(rsrc_in, raw_off[j], d_soff_raw, smem + rstage * RAW_BYTES + (8 * j + w) * 1024);
  };
  auto issue_u1 = [&](int ustage, int j) {
    if (ABLATE & 2) return;
    dma16_buf(rsrc_u, u_off, d_soff_u + j * 8192,
              smem + N_RSTAGE * RAW_BYTES + ustage * U_BYTES + (8 * j + w) * 1024);
  };

  // ---- the compute stream's position --------------------------------------------
  // (readfirstlane: the divisions run on the vector unit; without it the walkers' scalar state --
  //  and with it the LDS-DMA scalar offsets -- would be treated as divergent)
  int c_tail_vg = Lt;              // tail iterations still to compute
  // (The current item and the kind of the current segment are looked at by the epilogue and the segment switch
  //  only: they ride through the main loop in VGPRs, like pend_vg below, and are read back with readfirstlane
  //  where they are needed.  The loop has no SGPR to spare.)
  int c_item_vg = __builtin_amdgcn_readfirstlane(Lt > 0 ? tail_item0 + (int)(t_begin / (unsigned)nchunks) * kpg + grp : lg);
  int c_chunk = __builtin_amdgcn_readfirstlane(Lt > 0 ? (int)(t_begin % (unsigned)nchunks) : 0);
  int seg_kind_vg = 0;             // bit 1: the current segment starts its item, bit 0: it is the whole item
  // a partial segment whose ticket is drawn in the next epilogue (-1: none).  Parked in a VGPR across the
  // main loop ("+v"; read back with readfirstlane in the epilogue): the loop has no SGPR to spare -- as a
  // scalar it pushed a buffer descriptor into VGPR lanes, four v_readlane per iteration.
  int pend_vg = -1;
  asm volatile("" : "+v"(pend_vg));

  f32x4 acc[8][4];   // [the wave's point p][logical column block cb']
#pragma unroll
  for (int e = 0; e < 8; e++)
#pragma unroll
    for (int cb = 0; cb < 4; cb++) acc[e][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  P2 v[8];           // V_it (the wave's 8 points) at the top of iteration `it`; rewritten in place with V_{it+1}
  // A pinned step s = 0..15 of an iteration is (point p = s >> 1, column-block pair s & 1): 4 MFMAs.
  f32x2 bfn[PF][2];  // filter fragments of steps 0..PF-1 of the next iteration (requested pre-barrier)

  // ---- prologue: iterations 0 and 1 in flight; V_0 and the first fragments un-pipelined -----
  d_chunk = c_chunk;
  d_tail = Lt;
  dma_set_item(c_item_vg);
  asm volatile("" : "+v"(c_item_vg));
#pragma unroll
  for (int j = 0; j < 4; j++) issue_raw1(0, j);
#pragma unroll
  for (int j = 0; j < 4; j++) issue_u1(0, j);
  // The DMA walker never leaves this workgroup's range: it stops on the last chunk, and the
  // last two iterations of the range fetch that chunk again into stages nobody reads (see body()).
  if ((ABLATE & 32768) && tid == 0) prm.dbg[(size_t)lg * 8 + 6] = __builtin_amdgcn_s_memrealtime();   // first stage requested
  if (L > 1) dma_advance();
  ring_pass();   // in the shadow of the first pieces' flight
  if (!(ABLATE & 8)) {
    wait_vmem_all();
    __syncthreads();
  }
  if ((ABLATE & 32768) && tid == 0) prm.dbg[(size_t)lg * 8 + 7] = __builtin_amdgcn_s_memrealtime();   // ... and landed
  if (L > 1) {
#pragma unroll
    for (int j = 0; j < 4; j++) issue_raw1(1, j);
#pragma unroll
    for (int j = 0; j < 4; j++) issue_u1(1, j);
    if (L > 2) dma_advance();
  }
  {
    P2 d[12];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int j = 0; j < 4; j++) d[k * 4 + j] = lds_read2(A_OFF(k, j));
    P2 tmp[8];
#pragma unroll
    for (int j = 0; j < 4; j++) tmp_col(tmp, d, j);
#pragma unroll
    for (int e = 0; e < 8; e++) v_point(v, tmp, e);
#pragma unroll
    for (int e = 0; e < PF; e++) {   // steps 0, 1: point 0, column-block pairs 0 and 1
      bfn[e][0] = lds_read2(b_base[2 * (e & 1) + 0] + (e >> 1) * 2048);
      bfn[e][1] = lds_read2(b_base[2 * (e & 1) + 1] + (e >> 1) * 2048);
    }
  }
#if !WINO_UNROLL2
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int j = 0; j < 4; j++) a_adr[k][j] ^= RAW_BYTES;   // iteration 0 reads raw_1 from R1
#endif
  if (ABLATE & 32768) {
    if (tid == 0) prm.dbg[(size_t)lg * 8 + 1] = __builtin_amdgcn_s_memrealtime();
  }
  if (ABLATE & 16) {  // diagnostic build only: in-kernel clock = d(s_memtime) / d(s_memrealtime)
    // The start stamps go to memory at once: kept in SGPRs across the main loop (which has none to
    // spare) they pushed loop-carried scalars into VGPRs and the build ran 15 % slower than the
    // product kernel -- not a faithful probe of its clock.
    if (tid == 0) {
      prm.dbg[(size_t)lg * 4 + 0] = __builtin_amdgcn_s_memtime();
      prm.dbg[(size_t)lg * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
  }

  // One pipeline step = iteration `it`.  ONE instantiation: the raw stage that holds raw_{it+1}
  // is selected by bit 15 of the twelve patch address registers a_adr[][] (R0 at 0, R1 at 32768; they
  // are flipped after every iteration), the filter stage (it % 3) is a run-time offset added to
  // the two fragment base registers, the DMA destinations are scalar.
  // The schedule inside is pinned with sched_barrier(0): left alone, hipcc sinks every
  // ds_read to just before its first use and the wave eats one LDS latency per point.
  int ub_cur[4], ub_nxt[4];   // b_base[] + the byte offset of filter stage it % 3 / (it + 1) % 3
#pragma unroll
  for (int cb = 0; cb < 4; cb++) {
    ub_cur[cb] = b_base[cb];
    ub_nxt[cb] = b_base[cb] + U_BYTES;
  }
  // (WINO_UNROLL2: the body exists twice, once per parity of `it`; iteration `it` reads raw_{it+1} from
  //  R[(it + 1) & 1], which is then a compile-time offset of the patch reads -- a_adr[][] stay on R0 and the
  //  twelve v_xor toggles per iteration go away.)
  auto body = [&](auto par_c, int it, int rs_dma, int us_cur, int us_nxt, int us_dma) {
    constexpr int ROFF = WINO_UNROLL2 ? (decltype(par_c)::value ? 0 : RAW_BYTES) : 0;
    if (ABLATE & 2048) { const unsigned long long t = stamp(); if (it) st_comp += t - st_prev; st_prev = t; }
    if (!(ABLATE & 8)) {
      wait_vmem_all();   // my DMA pieces of raw_{it+1} and U_{it+1} have landed
      __syncthreads();   // everyone's have; everyone is done with the stages refilled below
    }
    if (ABLATE & 2048) { const unsigned long long t = stamp(); st_wait += t - st_prev; st_prev = t; }
    // The 8 LDS-DMA pieces this wave contributes per iteration (4 of raw_{it+2} into R[it&1], 4 of
    // U_{it+2} into U[(it+2)%3]) are issued one per step in steps DMA0..DMA0+7 instead of in a
    // burst here: an LDS-DMA instruction holds the wave's issue port for >100 cycles, and
    // spread out the SIMD's other wave covers that with its MFMAs; starting at step 4 leaves the
    // last pieces a third of an iteration of flight time before the next vmcnt(0).
    // They are issued unconditionally: a branch per piece (the compiler routes half of the
    // conditions through a VALU compare) cost ~30 of the ~210 instructions between an iteration's
    // first and last MFMA.  In the last two iterations of the range, where nothing is left to
    // fetch, the walker stands on the range's last chunk (valid addresses) and the pieces land in
    // R[it&1] / U[(it+2)%3] like any others: free stages, disjoint from the ones an epilogue
    // stages its stores in, and drained by the next iteration's vmcnt(0) or by the one before exit.
    // raw_{it+1}: the stage is in a_adr[][]
    // fragment base addresses of U_it and U_{it+1}: ub_cur[] / ub_nxt[], loop-carried registers that are
    // rotated in the iteration's tail (where the wave would wait at the barrier anyway) -- formed here, at
    // the top, their eight adds sit in front of the iteration's first MFMAs; left inside the reads,
    // hipcc re-adds the stage offset before every one.  (us_cur / us_nxt: kept for the interface.)
    (void)us_cur; (void)us_nxt;
    // (WINO_UNROLL2: the two copies of the body use the two register sets in swapped roles, so the rotation
    //  is four adds into the set that has just gone dead instead of four moves and four adds)
    const int (&ucur)[4] = (WINO_UNROLL2 && decltype(par_c)::value) ? ub_nxt : ub_cur;
    const int (&unxt)[4] = (WINO_UNROLL2 && decltype(par_c)::value) ? ub_cur : ub_nxt;
    // filter fragment of step s (point s >> 1, column blocks 2 (s & 1) + c)
    auto bread = [&](const int (&base)[4], int s2_, int c) {
      return lds_read2(base[2 * (s2_ & 1) + c] + (s2_ >> 1) * 2048);
    };

    f32x2 bf[16][2];
#pragma unroll
    for (int e = 0; e < PF; e++) { bf[e][0] = bfn[e][0]; bf[e][1] = bfn[e][1]; }
    P2 d[12], tmp[8];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int pt = e >> 1, cbp = e & 1;
      // -- top of the step: every LDS request of this step, before any MFMA.  Consumers sit
      //    at least one step later, so their waits are counted (lgkmcnt(N)), not drains.
      if (ABLATE & 64) {
        const f32x2 fake = {v[(e + 3) & 7].x, v[(e + 5) & 7].y};
        if (e + PF < 16) { bf[e + PF][0] = fake; bf[e + PF][1] = fake; }
        else { bfn[e + PF - 16][0] = fake; bfn[e + PF - 16][1] = fake; }
      } else if (e + PF < 16) {  // filter fragments of step e+PF
        bf[e + PF][0] = bread(ucur, e + PF, 0);
        bf[e + PF][1] = bread(ucur, e + PF, 1);
      } else {                   // ... and of steps 0, 1 of the next iteration
        bfn[e + PF - 16][0] = bread(unxt, e + PF - 16, 0);
        bfn[e + PF - 16][1] = bread(unxt, e + PF - 16, 1);
      }
      if (e >= DMA0 && e < DMA0 + 4) {
        issue_raw1(rs_dma, e - DMA0);
      } else if (e >= DMA0 + 4 && e < DMA0 + 8) {
        issue_u1(us_dma, e - DMA0 - 4);
      }
      // next iteration's A operand rides along: steps 0-5 read its 12 patch pixels (two per step, in
      // the order R0[j], R1[j], R2[j] column by column); B^T d B itself is written below, after the
      // step's MFMAs, and is deliberately NOT pinned to its step: the optimizer sinks it behind the
      // last MFMA of the iteration, where it runs as one burst of packed adds while the SIMD's other
      // wave still has MFMAs to issue.  Measured alternatives (with 16 points per wave), all slower:
      // pinned to steps 2-15 (scalar or packed, +3..+8 %: arithmetic between a wave's MFMAs delays its
      // own next MFMA, and when it waits on a patch read the whole wave stalls behind it, in-order
      // issue); B^T d as the loop-carried state with every point formed one step ahead of its use
      // (+8 %); the two waves of a SIMD phase-shifted (+6 %).
      // (After the last iteration this works on stale LDS; the result is never used -- cheaper
      // than a branch.)
      if (e < 6 && !(ABLATE & 32)) {
        const int q0 = 2 * e, q1 = 2 * e + 1;   // read index q -> (row k = q % 3, column j = q / 3)
        d[(q0 % 3) * 4 + q0 / 3] = lds_read2(A_OFF(q0 % 3, q0 / 3) + ROFF);
        d[(q1 % 3) * 4 + q1 / 3] = lds_read2(A_OFF(q1 % 3, q1 / 3) + ROFF);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABLATE & 96)) wait_lds(lds_wait_count(e));
      __builtin_amdgcn_sched_barrier(0);
      if (e >= 3 && e <= 7 && e != 5 && !(ABLATE & 32) && !(ABLATE & 8192)) tmp_col(tmp, d, e == 3 ? 0 : e == 4 ? 1 : e == 6 ? 2 : 3);
      if ((ABLATE & 8192) && e == 7) {
#pragma unroll
        for (int i = 0; i < 12; i++) asm volatile("" :: "v"(d[i]));
      }
      const P2 a = v[pt];
      const f32x2 b0 = bf[e][0], b1 = bf[e][1];
      if (ABLATE & 4) {  // keep the operands live, skip the matrix pipe
        asm volatile("" ::"v"(a.x), "v"(a.y), "v"(b0.x), "v"(b0.y), "v"(b1.x), "v"(b1.y));
      } else {
        // Tied destination: the accumulate chain stays in place (same vDst as SrcC is the
        // back-to-back form the matrix pipe forwards without wait states).  Left to the register
        // allocator, the builtin form ping-pongs every accumulator through a temporary tuple
        // (dst != SrcC), which costs the dependent MFMA several passes.  The compiler does not
        // see MFMA hazards of inline asm: the only VALU access to acc[] is in the epilogue, behind
        // explicit s_nops.
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[pt][2 * cbp + 0]) : "v"(a.x), "v"(b0.x));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[pt][2 * cbp + 1]) : "v"(a.x), "v"(b1.x));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[pt][2 * cbp + 0]) : "v"(a.y), "v"(b0.y));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[pt][2 * cbp + 1]) : "v"(a.y), "v"(b1.y));
      }
      if ((ABLATE & 4096) && !(ABLATE & 8192) && e == 9) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("" :: "v"(tmp[i]));
      }
      if (!(ABLATE & 32) && !(ABLATE & (4096 | 8192))) {
        // point p retires with step 2 p + 1; all patch reads are in by step 7
        if (e >= 9 && e < 15) v_point(v, tmp, e - 9);   // points 0..5
        if (e == 15) {
          v_point(v, tmp, 6);
          v_point(v, tmp, 7);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- per-wave epilogue (the DMA pipeline keeps running underneath) -------------------------
  // Slab image of one partial segment: [wave 0..7][q = 2r+cb][lane] of 16 B (the 2x2 output
  // pixels of tile row 4h+r, out-channel cb*16+t16, pre-BN).  Waves hand their 8 KiB parts over
  // independently: wave w of every segment of an item draws on tickets[8*item + w].
  //
  // A segment ends with the item's last chunk or with the range.  Whole segments are finalized
  // from registers.  A partial segment publishes its slab part; if more work follows (it is the
  // head of the range) its ticket is drawn one segment later, when the stores have long drained;
  // the range's last segment draws at once.  Whoever draws an item's last ticket gathers all
  // parts (in segment order: bitwise reproducible whoever reduces) and finalizes.
  //
  // Everything the epilogue needs beyond the accumulators is re-derived here from an OPAQUE copy
  // of the lane / wave id and from the kernarg segment, so that nothing epilogue-only is hoisted
  // out of the segment loop and kept in registers across the main loop (which has none to spare).
  // rfree / ufree = byte offsets of the two LDS stages nobody needs at a segment boundary
  // (raw_{it+1} is already in registers, U_it is spent); each wave takes 8 KiB of them.
  auto epilogue = [&](bool last_of_range, int rfree, int ufree) {
    unsigned long long ph_t = 0;
    auto phase = [&](int k) { if (ABLATE & 2048) { const unsigned long long t = stamp(); if (k >= 0) st_ph[k] += t - ph_t; ph_t = t; } };
    phase(-1);
    // every wave is done reading those two stages (a bare barrier: no memory counter needs to
    // drain here, and __syncthreads() would wait for the LDS-DMA pieces in flight)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the last MFMAs' results must have landed before VALU code reads / rewrites the accumulators
    // (the hazard recognizer cannot see through the inline-asm MFMAs)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    phase(0);
    int ln = lane, wv = w;
    asm volatile("" : "+v"(ln));
    asm volatile("" : "+s"(wv));
    const int e_t16 = ln & 15, e_h = ln >> 4, e_wt = wv >> 1, e_wk = wv & 1;
    char* wreg = smem + (wv < 4 ? rfree + wv * 8192 : ufree + (wv - 4) * 8192);
    KernargPtr kp = kernarg();
    const Geo geo = load_geo(kp);
    const int Wp = GEN ? geo.Wp : WINO_HW, Hp = GEN ? geo.Hp : WINO_HW;
    const int N = kp->N, K = kp->K, relu = kp->relu, KBLK = K >> 6, totalTiles = N * (GEN ? (int)geo.tiles : WINO_TILES);
    const unsigned sk_q = kp->sk_q, sk_rem = kp->sk_rem;
    const int tail_item0 = kp->ndp * G;
    // k-groups of the tail: this workgroup is range lp_e of group grp_e (slab slots are numbered by (tail range,
    // group)); worked out only by the partial-segment paths below, with multipliers instead of divisions
    struct TailPos { int kpg, Gp, lp, grp; FastDiv d_kp; };
    auto tail_pos = [&]() {
      TailPos t;
      t.kpg = kp->kp; t.Gp = kp->Gp;
      t.d_kp.m = kp->d_kp.m; t.d_kp.l = kp->d_kp.l;
      FastDiv dP, dC;
      dP.m = kp->d_P.m; dP.l = kp->d_P.l; dC.m = kp->d_copies.m; dC.l = kp->d_copies.l;
      const int lpos_e = (int)fastdiv((unsigned)lg, t.d_kp);
      t.grp = lg - lpos_e * t.kpg;
      t.lp = tail_range_of_fast(lpos_e, kp->ph_P, kp->ph_inv, kp->ph_copies, dP, dC);
      return t;
    };
    const float* bnBias = kp->bnBias;
    const float* bnScale = kp->bnScale;
    unsigned* tickets = kp->tickets;
    const auto rsrc_out = make_rsrc(kp->out, (unsigned)((size_t)N * Hp * Wp * K * sizeof(float)));
    const auto rsrc_slab = make_rsrc(kp->slabs, (unsigned)((size_t)2 * G * SLAB_BYTES));
    const unsigned slab_voff = (unsigned)((wv * 8 * 64 + ln) * 16);

    auto load_bn = [&](int item, float (&sc)[2], float (&bi)[2]) {
      const int kb = item % KBLK;
#pragma unroll
      for (int cb = 0; cb < 2; cb++) {
        const int kc = kb * KB + e_wk * 32 + cb * 16 + e_t16;
        sc[cb] = bnScale[kc];
        bi[cb] = bnBias[kc];
      }
    };
    auto draw_ticket = [&](int item) -> unsigned {   // one returning agent-scope add per wave
      unsigned old = 0;
      if (ln == 0)
        old = __hip_atomic_fetch_add(tickets + (size_t)item * 8 + wv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return old;   // meaningful in lane 0; readfirstlane at the use
    };

    // folded BN of this segment's item (this lane's two out-channels); in flight during A^T m A
    const int c_item = __builtin_amdgcn_readfirstlane(c_item_vg), seg_kind = __builtin_amdgcn_readfirstlane(seg_kind_vg);
    float bn_sc[2], bn_bi[2];
    load_bn(c_item, bn_sc, bn_bi);
    // The previous segment's deferred ticket is drawn now.
    int pend_item = __builtin_amdgcn_readfirstlane(pend_vg);
    unsigned pend_old = 0;
    if (TAIL && pend_item >= 0) pend_old = draw_ticket(pend_item);   // in flight while A^T m A runs
    // A^T m A (C/D layout: col = lane&15, row = 4*(lane>>4)+r).  The wave holds point rows i = 2 ph and
    // 2 ph + 1 (local rows 0, 1) for all four column blocks.  Per (tile row r, column block cb):
    //   column transform inside each point row:  c0(i) = m_i0 + m_i1 + m_i2,  c1(i) = m_i1 - m_i2 - m_i3
    //   row transform:  Y[0][b] = c_b(0) + c_b(1) + c_b(2),  Y[1][b] = c_b(1) - c_b(2) - c_b(3)
    // of which this wave can form the part of its two rows:
    //   ph = 0:  P[0][b] = c_b(0) + c_b(1),  P[1][b] = c_b(1)      ph = 1:  P[0][b] = c_b(2),  P[1][b] = -(c_b(2) + c_b(3))
    // (selects on the wave-uniform ph, no multiplies by 0 / +-1: exact, and an Inf in one part cannot
    // turn the other into NaN).  Pixel index p = 2a + b.
    // The parts of logical column blocks 2, 3 go to the partner wave (w ^ 1, the same tiles) through
    // this wave's 8 KB of the free LDS stages; the partner's parts of MY blocks 0, 1 come back the same
    // way.  y[r][cb] = the 2x2 output pixels of tile row 4h+r, out-channel (32 ph) + cb*16 + t16 --
    // from here on exactly what the wave held when it owned 16 points of 32 out-channels.
    const bool ph1 = (wv & 1) != 0;
    auto part = [&](int r, int cb) {
      float c[2][2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const float m0 = acc[i * 4 + 0][cb][r], m1 = acc[i * 4 + 1][cb][r];
        const float m2 = acc[i * 4 + 2][cb][r], m3 = acc[i * 4 + 3][cb][r];
        c[i][0] = m0 + m1 + m2;
        c[i][1] = m1 - m2 - m3;
      }
      f32x4 o;
#pragma unroll
      for (int bb = 0; bb < 2; bb++) {
        const float sum = c[0][bb] + c[1][bb];
        o[bb] = ph1 ? c[0][bb] : sum;          // a = 0
        o[2 + bb] = ph1 ? -sum : c[1][bb];     // a = 1
      }
      return o;
    };
    f32x4 y[4][2];
    {
      char* const xmine = wreg + ln * 16;
      const char* const xpart = smem + ((wv ^ 1) < 4 ? rfree + (wv ^ 1) * 8192 : ufree + ((wv ^ 1) - 4) * 8192) + ln * 16;
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++) *(f32x4*)(xmine + (2 * r + cb) * 1024) = part(r, 2 + cb);
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++) y[r][cb] = part(r, cb);
      // every wave's outgoing parts are in LDS (its own ds_writes drained) before anyone reads them
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++) y[r][cb] += *(const f32x4*)(xpart + (2 * r + cb) * 1024);
      // ... and read, before the finalize below stages output rows through the same LDS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int e = 0; e < 8; e++)
#pragma unroll
      for (int cb = 0; cb < 4; cb++) acc[e][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    phase(1);
    const bool whole = !TAIL || (seg_kind & 1) != 0;
    // up to two items to look at: [0] this segment's, [1] the deferred head segment's
    // (job1 is taken BEFORE this segment may re-arm pend_item with its own deferred ticket -- the second
    //  partial segment of a range that goes on to whole items.  Round 1 tested pend_item != c_item after the
    //  re-arm and lost the drawn item in exactly that case: harmless while the range's neighbour, whose
    //  ticket on that item waits behind a whole item, draws last; an item never finalized when this
    //  workgroup starts late -- a grid beyond the CU count, or CUs shared with another stream's kernel.)
    int job0 = -1, job1 = TAIL ? pend_item : -1;
    pend_item = -1;
    unsigned old0 = 0;
    if (whole) {
      job0 = c_item;
    } else if (!(ABLATE & 1024)) {
      // slab slot: 2l for the segment that continues an item (head of l's range), 2l+1 for the
      // one that starts an item
      const TailPos tp = tail_pos();
      const unsigned my_slot = 2u * (unsigned)(tp.lp * tp.kpg + tp.grp) + ((seg_kind & 2) ? 1u : 0u);
#pragma unroll
      for (int q = 0; q < 8; q++)
        slab_store16(y[q >> 1][q & 1], rsrc_slab, slab_voff + q * 1024, my_slot * SLAB_BYTES);
      // When is this segment's ticket drawn?  Whoever draws an item's last ticket gathers it, and a gather is
      // 3-5 us of work that only an epilogue can do:
      //  * the range's last segment: now (there is no later epilogue);
      //  * a segment that STARTS its item in a range that already had a partial segment (a "straddler": its
      //    range crosses an item boundary, so it pays one epilogue more than its neighbours): now.  Deferred to
      //    the final epilogue it was always its item's last arriver: the workgroups that are the slowest anyway
      //    gathered at the very end of the launch (round 1: exits spread over 106 .. 117.6 us at 256 channels,
      //    N = 128; tools/ablate_fused 256 256 t).  The wait costs little: the next iteration's vmcnt(0)
      //    would have waited for the same stores;
      //  * any other partial segment: at the next epilogue, when its stores have long drained.  The only tail
      //    segment of a range that lies inside one item thus draws at the final epilogue: these workgroups have
      //    the slack (no second partial epilogue), and they become the gatherers of the 3-segment items.
      // 256 channels: N = 128 118.7 -> 117.7 us, N = 96 98.5 -> 94.8; elsewhere within +-0.7 %.  (On top of this, the
      // gather's slabs requested all at once, the deferred item's before this segment's own finalize, with the
      // accumulators re-zeroed at the end of the epilogue to make room: 0.1-0.6 % slower everywhere.)
      const bool now = last_of_range || ((seg_kind & 2) && job1 >= 0);
      if (!now) {
        pend_item = c_item;        // ticket deferred to the next segment's epilogue
      } else {
        wait_vmem_all();           // this wave's write-through stores have left ...
        old0 = draw_ticket(c_item);   // ... before its ticket
        job0 = c_item;
      }
    }
    pend_vg = pend_item;
    asm volatile("" : "+v"(pend_vg));
    phase(2);
#pragma unroll 1
    for (int j = 0; j < (TAIL ? 2 : 1); j++) {
      const int item = j == 0 ? job0 : job1;
      if (item < 0) continue;
      if (!(j == 0 && whole)) {
        // which logical workgroups share `item`: walk outwards from lg.  With more workgroups
        // than iterations some own nothing; they are not segments.
        const TailPos tp = tail_pos();
        const int kpg = tp.kpg, Gp = tp.Gp, grp_e = tp.grp;
        const unsigned x0 = fastdiv((unsigned)(item - tail_item0), tp.d_kp) * (unsigned)nchunks, x1 = x0 + nchunks - 1;   // the group's tail space
        int gA = tp.lp, gB = tp.lp;
        while (sk_start(gA, sk_q, sk_rem, Gp) > x0) gA--;
        while (gB + 1 < Gp && sk_start(gB + 1, sk_q, sk_rem, Gp) <= x1) gB++;
        int nseg = 0;
        for (int g = gA; g <= gB; g++)
          nseg += sk_start(g + 1, sk_q, sk_rem, Gp) != sk_start(g, sk_q, sk_rem, Gp);
        const unsigned old = __builtin_amdgcn_readfirstlane(j == 0 ? old0 : pend_old);
        if (old != (unsigned)(nseg - 1)) {   // another workgroup's wave w will finish the item
          // ... unless the counter was not zero when the launch began (a launch that died mid-way before this
          // one): say so on the host-visible word; the library then refuses the stream until it is reset
          if (old >= (unsigned)nseg && ln == 0) __hip_atomic_store(kp->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          continue;
        }
        if (ln == 0)   // self-cleaning counter: the next launch finds 0 again.  (Subtracted, not stored: a counter
                       // that was NOT zero at launch then stays off by the same amount, and the item's last
                       // drawer is certain to see a value >= nseg -- the report above cannot be missed.)
          __hip_atomic_fetch_sub(tickets + (size_t)item * 8 + wv, (unsigned)nseg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // gather: all segments' parts, summed in segment order
        bool first = true;
#pragma unroll 1
        for (int g = gA; g <= gB; g++) {
          if (sk_start(g + 1, sk_q, sk_rem, Gp) == sk_start(g, sk_q, sk_rem, Gp)) continue;   // owns nothing
          const unsigned slot = 2u * (unsigned)(g * kpg + grp_e) + (first ? 1u : 0u);   // (tail range g, this group)
          f32x4 t[8];
#pragma unroll
          for (int q = 0; q < 8; q++) t[q] = slab_load16(rsrc_slab, slab_voff + q * 1024, slot * SLAB_BYTES);
#pragma unroll
          for (int q = 0; q < 8; q++) y[q >> 1][q & 1] = first ? t[q] : y[q >> 1][q & 1] + t[q];
          first = false;
        }
      }
      float sc2[2] = {bn_sc[0], bn_sc[1]}, bi2[2] = {bn_bi[0], bn_bi[1]};
      if (item != c_item) load_bn(item, sc2, bi2);   // the deferred head item
      if (ABLATE & 512) continue;  // price the store tail

      // ---- finalize: BN + ReLU, then the wave's 16 tiles x 2x2 px x 32 out-channels go through
      // its private 8 KiB of LDS so that they leave as whole 128-byte runs of the padded NHWC
      // output (16 B per lane, 8 runs per store).  (The zero ring is written by the ring pass at
      // the start of the kernel.)
      // Image: [tile 0..15][px 0..3][k 0..31] floats; the 16-float group index (2*px + cb) is
      // XORed with (tile>>2)&3 = the MFMA row group h, which makes the ds_write_b32 of the 4 row
      // groups and the ds_read_b128 of every 16 lanes hit 64 distinct banks.
      const int tb = item / KBLK, kb = item - tb * KBLK;
      int ep_wbase[4], ep_rbase[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        ep_wbase[jj] = e_h * 2048 + ((jj ^ e_h) << 6) + e_t16 * 4;                       // + r*512 + (g>>2)*256
        const int px = (ln >> 3) & 3, c = ln & 7;
        ep_rbase[jj] = (ln >> 5) * 512 + (((px * 2 + (c >> 2)) ^ jj) << 6) + (c & 3) * 16;   // + 2i*512
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
#pragma unroll
          for (int pp = 0; pp < 4; pp++) {
            float v1 = sc2[cb] * y[r][cb][pp] + bi2[cb];
            if (relu) v1 = fmaxf(v1, 0.f);
            const int g = pp * 2 + cb;
            *(float*)(wreg + ep_wbase[g & 3] + r * 512 + (g >> 2) * 256) = v1;
          }
        }
      }
      // lane -> run (lane>>3) = (tile 2i + (lane>>5), px (lane>>3)&3), 16-byte chunk lane&7
      // (tried: the tile arithmetic on the scalar unit, one tile per half-wave -- 40 % slower)
      const int px = (ln >> 3) & 3, pa = px >> 1, pb = px & 1;
      const unsigned kbyte = (unsigned)((kb * KB + e_wk * 32 + (ln & 7) * 4) * sizeof(float));
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const f32x4 val = *(const f32x4*)(wreg + ep_rbase[i >> 1] + i * 1024);
        const int g = tb * TB + e_wt * 16 + 2 * i + (ln >> 5);
        const bool live = g < totalTiles;
        const TileCoord tc = decode_tile_g<GEN>(live ? g : 0, geo);
        const int py = 1 + 2 * tc.ty + pa, pxx = 1 + 2 * tc.tx + pb;
        const unsigned img = (unsigned)(tc.n * Hp * Wp);
        // (odd H or W: the last tile row / column computes one output row / column too many, from
        //  patch rows that belong to the next image row or lie past the tensor -- the descriptor's
        //  range check returns 0 there; a Winograd output only depends on its own 3x3 window, so
        //  the kept outputs are unaffected, and the surplus ones must not reach the ring)
        const bool keep = live && (!GEN || (py <= Hp - 2 && pxx <= Wp - 2));
        // Non-temporal: the output is written once and, at 33 MB for the reference layer, exceeds the
        // 32 MB of L2 anyway.  Old and new libraries interleaved: 256ch 125.4 -> 123.9 us, 128ch
        // 43.5 -> 41.95, the other feature maps -0.4..-1.8 %; HBM bytes and L2 hit rate unchanged;
        // the bottleneck blocks, whose last launch reads this output at once, within +-0.3 %.
        // (A run-time choice between the two store forms cost more than it saved: its extra live
        // scalar pushed SGPR spill code into the MFMA loop.)
        if (keep) buf_store16_nt(val, rsrc_out, (unsigned)((img + py * Wp + pxx) * K * sizeof(float)) + kbyte, 0);
      }
    }
    phase(3);
  };

  // ================================ main loop =====================================
  // Per segment: the tight loop over its iterations, then the one epilogue site.
  {
    int us = 0;  // filter stage of iteration `it` (= it % 3)
    auto next = [](int s) { return s == 2 ? 0 : s + 1; };
    int it = 0;
#pragma unroll 1
    for (;;) {
      // iterations of this segment: to the end of the item, or of the tail range
      int c_tail = __builtin_amdgcn_readfirstlane(c_tail_vg);
      const int n = c_tail > 0 && c_tail < nchunks - c_chunk ? c_tail : nchunks - c_chunk;
      seg_kind_vg = c_chunk == 0 ? (n == nchunks ? 3 : 2) : 0;
      asm volatile("" : "+v"(seg_kind_vg));
      // the walker's state after this segment, settled now so that neither n nor c_chunk lives across the loop:
      // c_tail < 0 = "the tail ends with this segment"
      if (c_tail > 0 && (c_tail -= n) == 0) c_tail = -1;
      c_tail_vg = c_tail;
      asm volatile("" : "+v"(c_tail_vg));
      c_chunk = 0;   // every later segment starts its item
      int us_last = us;
      auto tail = [&](auto par_c) {   // everything between two bodies
#if !WINO_UNROLL2
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
          for (int j = 0; j < 4; j++) a_adr[k][j] ^= RAW_BYTES;
#endif
        if (it + 3 < L) dma_advance();
        us_last = us;
        us = next(us);
        it++;
        {   // rotate the filter-fragment bases: U_{it} was U_{it+1}; the new U_{it+1} is stage next(us)
          const int off = next(us) * U_BYTES;
#pragma unroll
          for (int cb = 0; cb < 4; cb++) {
#if WINO_UNROLL2
            // the set the body just used as "current" is dead: it becomes the next body's "next"
            if (decltype(par_c)::value) ub_nxt[cb] = b_base[cb] + off;
            else ub_cur[cb] = b_base[cb] + off;
#else
            ub_cur[cb] = ub_nxt[cb];
            ub_nxt[cb] = b_base[cb] + off;
#endif
          }
        }
      };
#if WINO_UNROLL2
      {
        int k = n;
#pragma unroll 1
        for (;;) {
          if (!(it & 1)) {
            body(std::integral_constant<int, 0>{}, it, 0, us, next(us), next(next(us)));
            tail(std::integral_constant<int, 0>{});
            if (--k == 0) break;
          }
          body(std::integral_constant<int, 1>{}, it, 1, us, next(us), next(next(us)));
          tail(std::integral_constant<int, 1>{});
          if (--k == 0) break;
        }
      }
#else
#pragma unroll 1
      for (int k = n; k > 0; k--) {
        body(std::integral_constant<int, 0>{}, it, it & 1, us, next(us), next(next(us)));
        tail(std::integral_constant<int, 0>{});
      }
#endif
      // the segment's last iteration was it-1: raw stage R[it & 1] and filter stage U[us_last] are free
      const bool last_of_range = it == L;
      if (ABLATE & 2048) { const unsigned long long t = stamp(); st_comp += t - st_prev; st_prev = t; }
      if ((ABLATE & 32768) && last_of_range && tid == 0) {
        KernargPtr kp = kernarg();
        kp->dbg[(size_t)lg * 8 + 2] = __builtin_amdgcn_s_memrealtime();
      }
      epilogue(last_of_range, (it & 1) * RAW_BYTES, N_RSTAGE * RAW_BYTES + us_last * U_BYTES);
      if (ABLATE & 2048) { const unsigned long long t = stamp(); st_epi += t - st_prev; st_prev = t; }
      if (last_of_range) break;
      c_tail = __builtin_amdgcn_readfirstlane(c_tail_vg);
      if (c_tail < 0) {   // tail done: first whole item
        c_item_vg = lg;
        c_tail_vg = 0;
      } else {
        c_item_vg += c_tail > 0 ? kernarg()->kp : G;
      }
      asm volatile("" : "+v"(c_item_vg), "+v"(c_tail_vg));
    }
  }
#undef A_OFF
  wait_vmem_all();   // no LDS-DMA of this wave may land after the workgroup's LDS has been given away
  clk_exit();

  // diagnostic builds: the stamps go to a buffer of their own (prm.dbg), never into an output
  if (ABLATE & 2048) {
    if (lane == 0) {
      unsigned long long* dbg = prm.dbg + ((size_t)lg * 8 + w) * 8;
      dbg[0] = st_wait;
      dbg[1] = st_comp;
      dbg[2] = st_epi;
      dbg[3] = st_ph[0];
      dbg[4] = st_ph[1];
      dbg[5] = st_ph[2];
      dbg[6] = st_ph[3];
    }
  }
  if (ABLATE & 32768) {
    if (tid == 0) {
      KernargPtr kp = kernarg();
      kp->dbg[(size_t)lg * 8 + 3] = __builtin_amdgcn_s_memrealtime();
      kp->dbg[(size_t)lg * 8 + 5] = __builtin_amdgcn_s_memtime();
    }
  }
  if (ABLATE & 16) {
    if (tid == 0) {
      KernargPtr kp = kernarg();
      unsigned long long* dbg = kp->dbg + (size_t)lg * 4;
      dbg[2] = __builtin_amdgcn_s_memtime();
      dbg[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

}  // namespace fused
}  // namespace wino
