// The fused Winograd F(2x2,3x3) kernel, ONE WAVE PER SIMD: 256-thread workgroups whose four waves
// each own 32 tiles x 32 out-channels x 16 points (64 accumulator tiles = all 256 AGPRs) and have
// the other 256 registers for operands.  Same items, LDS stages, LDS-DMA stream, work decomposition
// (whole-item rounds + stream-K tail), slab hand-off and output contract as the 8-wave kernel in
// wino_f2_fused_kernel.h -- read its header first; this file only documents what differs.
//
// Why: a wave issues in order.  With two MFMA waves per SIMD, a wave whose next instruction is an
// MFMA sits behind its mate's MFMA for up to 32 cycles and can issue nothing else meanwhile, so a
// wave's iteration costs (its MFMAs at the shared rate) + (all its LDS reads, LDS-DMAs, adds, scalar
// work): the 8-wave kernel measures 5150 cycles per iteration for 4096 of matrix-pipe work, and
// removing any non-MFMA component shortens it by about that component's issue time
// (tools/ablate_fused).  A single MFMA stream per SIMD never waits for the pipe: after the 4-cycle
// issue of an MFMA the wave has 28 cycles of shadow for other instructions (tools/coissue: a mate's
// or the wave's own VALU/LDS work does not slow an MFMA stream), and per MFMA this kernel needs
// ~0.4 LDS reads, 0.125 LDS-DMAs and 0.5 packed adds.
//
// Differences in detail:
//   * wave (wt, wk) = tiles [32wt, 32wt+32) x out-channels [32wk, 32wk+32): two 16-tile groups (tg)
//     x two 16-channel groups (kt); per point and 8-channel chunk 8 MFMAs, each A fragment (registers)
//     feeds 2, each B fragment (one ds_read_b64) feeds 2;
//   * per step: 2 filter-fragment reads, 3 patch reads (steps 0-10, column-major so that B^T d can
//     start on old reads), 1 LDS-DMA piece (16 per wave per iteration), 8 MFMAs;
//   * B^T d B just in time: the loop-carried state is tmpc = B^T d (64 registers); point e+1 is
//     formed during step e, points 12-15 during step 11, steps 12-15 rewrite tmpc from the new patch.
//     (In the 8-wave kernel this schedule was slower -- there every add sits in a wave that is also
//     waiting for the pipe; here it rides in the MFMA shadow.)
//   * epilogue: per wave 64 registers of A^T m A results, 16 KiB of LDS per wave (waves 0,1 share the
//     free raw stage, waves 2,3 the free filter stage), 16 stores of 8 x 128-byte runs; slab part
//     16 KiB per wave; tickets[8*item + wave] with 4 waves.
#pragma once
#include "wino_f2_fused_kernel.h"

namespace wino {
namespace fused4 {

using namespace fused;

constexpr int NT4 = 256;

// LDS requests at the top of step q: 2 filter fragments (point q+PF), then the patch reads
constexpr int n_raw(int q) { return 32 - 3 * q >= 3 ? 3 : (32 - 3 * q > 0 ? 32 - 3 * q : 0); }
constexpr int n_req(int q) { return 2 + n_raw(q); }
// requests younger than the filter fragments step e consumes (requested first thing in step e-PF)
constexpr int wait_count4(int e) {
  int after = 15;
  if (e >= PF) {
    after = n_raw(e - PF);
    for (int q = e - PF + 1; q <= e; q++) after += n_req(q);
  }
  return after > 15 ? 15 : after;
}
// patch read r (0..31), column-major, tile groups interleaved: never two reads that share a base
// register at a ds_read2-fusable distance in one step
constexpr int raw_tg(int r) { return r & 1; }
constexpr int raw_px(int r) { return (((r >> 1) & 3) << 2) | (r >> 3); }   // row (r>>1)&3, column r>>3

template <int ABLATE>
__global__ void __launch_bounds__(NT4, 1)
wino_f2_fused4_kernel(const FusedParams prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const float* __restrict__ in = prm.in;
  const float* __restrict__ Uq = prm.Uq;
  const int N = prm.N, C = prm.C, K = prm.K;
  const unsigned sk_q = prm.sk_q, sk_rem = prm.sk_rem;
  const int ndp = prm.ndp;

  const int KBLK = K >> 6;
  const int nchunks = C / BC;
  const int G = gridDim.x;
  const int lg = (int)(blockIdx.x & 7) * (G >> 3) + ((int)(blockIdx.x & 7) < (G & 7) ? (int)(blockIdx.x & 7) : (G & 7)) +
                 (int)(blockIdx.x >> 3);
  const int tail_item0 = ndp * G;
  const unsigned t_begin = __builtin_amdgcn_readfirstlane(sk_start(lg, sk_q, sk_rem, G));
  const int Lt = (int)(__builtin_amdgcn_readfirstlane(sk_start(lg + 1, sk_q, sk_rem, G)) - t_begin);
  const int L = Lt + ndp * nchunks;

  auto ring_pass = [&]() {   // see the 8-wave kernel
    if (ABLATE & 512) return;
    const unsigned upp = (unsigned)K >> 2;
    const unsigned long long U = (unsigned long long)N * 60u * upp;
    const unsigned u_begin = (unsigned)(U * (unsigned)lg / (unsigned)G);
    const unsigned u_end = (unsigned)(U * ((unsigned)lg + 1u) / (unsigned)G);
    const auto rsrc_ring = make_rsrc(prm.out, (unsigned)((size_t)N * WINO_HW * WINO_HW * K * sizeof(float)));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (unsigned u = u_begin + threadIdx.x; u < u_end; u += NT4) {
      const unsigned pid = u / upp, unit = u - pid * upp;
      const unsigned n = pid / 60u, q = pid - n * 60u;
      const unsigned y = q < 16 ? 0u : q < 32 ? (unsigned)(WINO_HW - 1) : q < 46 ? q - 31u : q - 45u;
      const unsigned x = q < 16 ? q : q < 32 ? q - 16u : q < 46 ? 0u : (unsigned)(WINO_HW - 1);
      buf_store16(zero4, rsrc_ring, (((n * WINO_HW + y) * WINO_HW + x) * (unsigned)K + unit * 4u) * (unsigned)sizeof(float), 0);
    }
  };
  if (L <= 0) {
    ring_pass();
    return;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = w >> 1;  // which 32-tile half of the 64
  const int wk = w & 1;   // which 32-channel half of the 64

  const unsigned u_off = lane * 16;
  const unsigned u_chunk_stride = (unsigned)(KBLK * U_CHUNK_FLOATS * sizeof(float));
  const auto rsrc_in = make_rsrc(in, (unsigned)((size_t)N * WINO_HW * WINO_HW * C * sizeof(float)));
  const auto rsrc_u = make_rsrc(Uq, (unsigned)((size_t)16 * C * K * sizeof(float)));

  // ---- fragment read addresses ---------------------------------------------------
  const int t16 = lane & 15, h = lane >> 4;
  // A: tile row tl = wt*32 + tg*16 + t16 (the XOR swizzle only sees t16); tile group 1 sits 8 KiB
  // further and gets base registers of its own (opaque: no ds_read2st64 fusion with group 0)
  const int a_base = (wt * 32 + t16) * 512 + ((h ^ (((t16 >> 3) & 1) << 1)) << 3);
  const int a_sw = t16 & 7;
  int a_lo[2][8];
#pragma unroll
  for (int p = 0; p < 8; p++) {
    a_lo[0][p] = a_base + ((p ^ a_sw) << 5);
    a_lo[1][p] = a_lo[0][p] + 16 * 512;
    asm volatile("" : "+v"(a_lo[1][p]));
  }
#define A_OFF4(tg, px) (a_lo[tg][(px) & 7] + (((px) >> 3) << 8))
  int b_base[2];
#pragma unroll
  for (int kt = 0; kt < 2; kt++) {
    const int kl = wk * 32 + kt * 16 + t16;
    b_base[kt] = N_RSTAGE * RAW_BYTES + kl * 32 + ((h ^ (((kl >> 3) & 1) << 1)) << 3);
  }
  asm volatile("" : "+v"(b_base[1]));

  typedef f32x2 P2;
  auto sub2 = [](const P2& a, const P2& b) { return a - b; };
  auto add2 = [](const P2& a, const P2& b) { return a + b; };
  auto ld2 = [](const char* p) { return *(const f32x2*)p; };
#define PIN2(val) asm volatile("" : "+v"(val))
  auto tmp_col = [&](P2* tmp, const P2* d, int j) {  // B^T d, column j
    tmp[0 * 4 + j] = sub2(d[0 * 4 + j], d[2 * 4 + j]);
    tmp[1 * 4 + j] = add2(d[1 * 4 + j], d[2 * 4 + j]);
    tmp[2 * 4 + j] = sub2(d[2 * 4 + j], d[1 * 4 + j]);
    tmp[3 * 4 + j] = sub2(d[1 * 4 + j], d[3 * 4 + j]);
  };
  auto v_of = [&](const P2* tmp, int e) -> P2 {  // (B^T d) B, point e
    const int i = e >> 2, j = e & 3;
    if (j == 0) return sub2(tmp[i * 4 + 0], tmp[i * 4 + 2]);
    if (j == 1) return add2(tmp[i * 4 + 1], tmp[i * 4 + 2]);
    if (j == 2) return sub2(tmp[i * 4 + 2], tmp[i * 4 + 1]);
    return sub2(tmp[i * 4 + 1], tmp[i * 4 + 3]);
  };

  unsigned long long stamp_c = 0, stamp_r = 0;
  unsigned long long st_wait = 0, st_comp = 0, st_epi = 0, st_prev = 0;   // ABLATE & 2048
  auto stamp = [&]() -> unsigned long long {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };

  typedef const __attribute__((address_space(4))) FusedParams* KernargPtr;
  auto kernarg = []() {
    KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
  };

  // ---- the DMA stream ------------------------------------------------------------
  // raw piece q = 4*j + w (j = 0..7) covers tiles 2q, 2q+1; filter piece q = 4*j + w is KiB q of the chunk
  unsigned raw_off[8];
  unsigned d_soff_raw = 0, d_soff_u = 0;
  int d_item, d_chunk, d_tail, d_tb = -1;
  auto dma_set_item = [&](int item) {
    const int tb = item / KBLK, kb = item - tb * KBLK;
    d_item = item;
    if (tb != d_tb) {
      d_tb = tb;
      KernargPtr kp = kernarg();
      const int C = kp->C, totalTiles = kp->N * WINO_TILES;
      const int up = lane & 31;
      const int pxp = up >> 1, halfp = up & 1;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int tl = 8 * j + 2 * w + (lane >> 5);
        const int px = pxp ^ (tl & 7);
        const int half = halfp ^ ((tl >> 3) & 1);
        int g = tb * TB + tl;
        g = g < totalTiles ? g : totalTiles - 1;
        const TileCoord tc = decode_tile(g);
        const int y = 2 * tc.ty + (px >> 2), x = 2 * tc.tx + (px & 3);
        raw_off[j] = (unsigned)((((size_t)(tc.n * WINO_HW + y) * WINO_HW + x) * C + half * 4) * sizeof(float));
      }
    }
    d_soff_raw = __builtin_amdgcn_readfirstlane((unsigned)(d_chunk * (BC * sizeof(float))));
    d_soff_u = __builtin_amdgcn_readfirstlane((unsigned)((kb * U_CHUNK_FLOATS + w * 256) * sizeof(float)) + d_chunk * u_chunk_stride);
    asm volatile("" : "+s"(d_soff_raw), "+s"(d_soff_u));
  };
  auto pin_dma_state = [&]() {
    d_item = __builtin_amdgcn_readfirstlane(d_item);
    d_chunk = __builtin_amdgcn_readfirstlane(d_chunk);
    d_tail = __builtin_amdgcn_readfirstlane(d_tail);
    d_tb = __builtin_amdgcn_readfirstlane(d_tb);
    asm volatile("" : "+s"(d_item), "+s"(d_chunk), "+s"(d_tail), "+s"(d_tb));
  };
  auto dma_advance = [&]() {
    if (d_tail > 0 && --d_tail == 0) {
      d_chunk = 0;
      dma_set_item(lg);
    } else if (++d_chunk == nchunks) {
      d_chunk = 0;
      dma_set_item(d_tail > 0 ? d_item + 1 : d_item + G);
    } else {
      d_soff_raw += (unsigned)(BC * sizeof(float));
      d_soff_u += u_chunk_stride;
      asm volatile("" : "+s"(d_soff_raw), "+s"(d_soff_u));
    }
    pin_dma_state();
  };
  auto issue_raw1 = [&](int rstage, int j) {
    if (ABLATE & 1) return;
    dma16_buf(rsrc_in, raw_off[j], d_soff_raw, smem + rstage * RAW_BYTES + (4 * j + w) * 1024);
  };
  auto issue_u1 = [&](int ustage, int j) {
    if (ABLATE & 2) return;
    dma16_buf(rsrc_u, u_off, d_soff_u + j * 4096,
              smem + N_RSTAGE * RAW_BYTES + ustage * U_BYTES + (4 * j + w) * 1024);
  };

  // ---- the compute stream's position ------------------------------------------------
  int c_tail = Lt;
  int c_item = __builtin_amdgcn_readfirstlane(Lt > 0 ? tail_item0 + (int)(t_begin / (unsigned)nchunks) : lg);
  int c_chunk = __builtin_amdgcn_readfirstlane(Lt > 0 ? (int)(t_begin % (unsigned)nchunks) : 0);
  int seg_c0 = c_chunk;
  int pend_item = -1;

  f32x4 acc[16][2][2];   // [point][tile group][channel group], AGPRs
#pragma unroll
  for (int e = 0; e < 16; e++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[e][q >> 1][q & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  P2 tmpc[2][16];    // B^T d of iteration `it`'s patches (rewritten in place in steps 12-15)
  P2 v0c[2];         // their point 0
  f32x2 bfn[PF][2];  // filter fragments of points 0..PF-1 of the next iteration

  // ---- prologue -----------------------------------------------------------------------
  d_chunk = c_chunk;
  d_tail = Lt;
  dma_set_item(c_item);
#pragma unroll
  for (int j = 0; j < 8; j++) issue_raw1(0, j);
#pragma unroll
  for (int j = 0; j < 8; j++) issue_u1(0, j);
  dma_advance();
  ring_pass();
  if (!(ABLATE & 8)) {
    wait_vmem_all();
    __syncthreads();
  }
  if (L > 1) {
#pragma unroll
    for (int j = 0; j < 8; j++) issue_raw1(1, j);
#pragma unroll
    for (int j = 0; j < 8; j++) issue_u1(1, j);
    dma_advance();
  }
#pragma unroll
  for (int tg = 0; tg < 2; tg++) {
    P2 d[16];
#pragma unroll
    for (int px = 0; px < 16; px++) d[px] = ld2(smem + A_OFF4(tg, px));
#pragma unroll
    for (int j = 0; j < 4; j++) tmp_col(tmpc[tg], d, j);
    v0c[tg] = v_of(tmpc[tg], 0);
  }
#pragma unroll
  for (int e = 0; e < PF; e++) {
    bfn[e][0] = *(const f32x2*)(smem + b_base[0] + e * 2048);
    bfn[e][1] = *(const f32x2*)(smem + b_base[1] + e * 2048);
  }
#pragma unroll
  for (int p = 0; p < 8; p++) { a_lo[0][p] ^= RAW_BYTES; a_lo[1][p] ^= RAW_BYTES; }   // iteration 0 reads raw_1 from R1
  if (ABLATE & 16) {
    stamp_c -= __builtin_amdgcn_s_memtime();
    stamp_r -= __builtin_amdgcn_s_memrealtime();
  }

  // ---- one iteration ------------------------------------------------------------------
  auto body = [&](int it, int rs_dma, int us_cur, int us_nxt, int us_dma) {
    if (ABLATE & 2048) { const unsigned long long t = stamp(); if (it) st_comp += t - st_prev; st_prev = t; }
    if (!(ABLATE & 8)) {
      wait_vmem_all();
      __syncthreads();
    }
    if (ABLATE & 2048) { const unsigned long long t = stamp(); st_wait += t - st_prev; st_prev = t; }
    const bool dma_on = it + 2 < L;
    const char* ucur0 = smem + b_base[0] + us_cur * U_BYTES;
    const char* ucur1 = smem + b_base[1] + us_cur * U_BYTES;
    const char* unxt0 = smem + b_base[0] + us_nxt * U_BYTES;
    const char* unxt1 = smem + b_base[1] + us_nxt * U_BYTES;

    f32x2 bf[16][2];
#pragma unroll
    for (int e = 0; e < PF; e++) { bf[e][0] = bfn[e][0]; bf[e][1] = bfn[e][1]; }
    P2 d[2][16], v[2][16];   // v[tg][e] lives from step e-1 to step e
    v[0][0] = v0c[0];
    v[1][0] = v0c[1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) {
      // -- top of the step: its LDS requests and its LDS-DMA piece
      if (e + PF < 16) {
        bf[e + PF][0] = *(const f32x2*)(ucur0 + (e + PF) * 2048);
        bf[e + PF][1] = *(const f32x2*)(ucur1 + (e + PF) * 2048);
      } else {
        bfn[e + PF - 16][0] = *(const f32x2*)(unxt0 + (e + PF - 16) * 2048);
        bfn[e + PF - 16][1] = *(const f32x2*)(unxt1 + (e + PF - 16) * 2048);
      }
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int r = 3 * e + i;
        if (r < 32) d[raw_tg(r)][raw_px(r)] = ld2(smem + A_OFF4(raw_tg(r), raw_px(r)));
      }
      if (dma_on) {
        if (e < 8) issue_raw1(rs_dma, e);
        else issue_u1(us_dma, e - 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      wait_lds(wait_count4(e));
      __builtin_amdgcn_sched_barrier(0);
      const P2 a0 = v[0][e], a1 = v[1][e];
      const f32x2 b0 = bf[e][0], b1 = bf[e][1];
      if (ABLATE & 4) {
        asm volatile("" ::"v"(a0), "v"(a1), "v"(b0), "v"(b1));
      } else {
        // tied accumulators in AGPRs; dependent pairs four MFMAs apart
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][0][0]) : "v"(a0.x), "v"(b0.x));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][0][1]) : "v"(a0.x), "v"(b1.x));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][1][0]) : "v"(a1.x), "v"(b0.x));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][1][1]) : "v"(a1.x), "v"(b1.x));
      }
      // -- the transform arithmetic of this step, in the shadow of the MFMAs above
      if (e <= 10) {
#pragma unroll
        for (int tg = 0; tg < 2; tg++) { v[tg][e + 1] = v_of(tmpc[tg], e + 1); PIN2(v[tg][e + 1]); }
      }
      if (e == 11) {
#pragma unroll
        for (int pt = 12; pt < 16; pt++)
#pragma unroll
          for (int tg = 0; tg < 2; tg++) { v[tg][pt] = v_of(tmpc[tg], pt); PIN2(v[tg][pt]); }
      }
      if (e >= 12) {
#pragma unroll
        for (int tg = 0; tg < 2; tg++) {
          tmp_col(tmpc[tg], d[tg], e - 12);
#pragma unroll
          for (int i = 0; i < 4; i++) PIN2(tmpc[tg][i * 4 + e - 12]);
        }
      }
      if (e == 15) {
#pragma unroll
        for (int tg = 0; tg < 2; tg++) { v0c[tg] = v_of(tmpc[tg], 0); PIN2(v0c[tg]); }
      }
      if (!(ABLATE & 4)) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][0][0]) : "v"(a0.y), "v"(b0.y));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][0][1]) : "v"(a0.y), "v"(b1.y));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][1][0]) : "v"(a1.y), "v"(b0.y));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[e][1][1]) : "v"(a1.y), "v"(b1.y));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- per-wave epilogue (see the 8-wave kernel; 4 waves x 16 KiB here) -----------------------
  auto epilogue = [&](bool last_of_range, int rfree, int ufree) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int ln = lane, wv = w;
    asm volatile("" : "+v"(ln));
    asm volatile("" : "+s"(wv));
    const int e_t16 = ln & 15, e_h = ln >> 4, e_wt = wv >> 1, e_wk = wv & 1;
    char* wreg = smem + (wv < 2 ? rfree + wv * 16384 : ufree + (wv - 2) * 16384);
    KernargPtr kp = kernarg();
    const int N = kp->N, K = kp->K, relu = kp->relu, KBLK = K >> 6, totalTiles = N * WINO_TILES;
    const unsigned sk_q = kp->sk_q, sk_rem = kp->sk_rem;
    const int tail_item0 = kp->ndp * G;
    const float* bnBias = kp->bnBias;
    const float* bnScale = kp->bnScale;
    unsigned* tickets = kp->tickets;
    const auto rsrc_out = make_rsrc(kp->out, (unsigned)((size_t)N * WINO_HW * WINO_HW * K * sizeof(float)));
    const auto rsrc_slab = make_rsrc(kp->slabs, (unsigned)((size_t)2 * G * SLAB_BYTES));
    const unsigned slab_voff = (unsigned)((wv * 16 * 64 + ln) * 16);

    auto load_bn = [&](int item, float (&sc)[2], float (&bi)[2]) {
      const int kb = item % KBLK;
#pragma unroll
      for (int kt = 0; kt < 2; kt++) {
        const int kc = kb * KB + e_wk * 32 + kt * 16 + e_t16;
        sc[kt] = bnScale[kc];
        bi[kt] = bnBias[kc];
      }
    };
    auto draw_ticket = [&](int item) -> unsigned {
      unsigned old = 0;
      if (ln == 0)
        old = __hip_atomic_fetch_add(tickets + (size_t)item * 8 + wv, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return old;
    };

    float bn_sc[2], bn_bi[2];
    load_bn(c_item, bn_sc, bn_bi);
    unsigned pend_old = 0;
    if (pend_item >= 0) pend_old = draw_ticket(pend_item);
    // A^T m A: y[tg][kt][r] = the 2x2 output pixels of tile row 16tg+4h+r, out-channel 16kt+t16
    f32x4 y[2][2][4];
#pragma unroll
    for (int tg = 0; tg < 2; tg++)
#pragma unroll
      for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float t0[4], t1[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float m0 = acc[0 * 4 + j][tg][kt][r], m1 = acc[1 * 4 + j][tg][kt][r];
            const float m2 = acc[2 * 4 + j][tg][kt][r], m3 = acc[3 * 4 + j][tg][kt][r];
            t0[j] = m0 + m1 + m2;
            t1[j] = m1 - m2 - m3;
          }
          y[tg][kt][r][0] = t0[0] + t0[1] + t0[2];
          y[tg][kt][r][1] = t0[1] - t0[2] - t0[3];
          y[tg][kt][r][2] = t1[0] + t1[1] + t1[2];
          y[tg][kt][r][3] = t1[1] - t1[2] - t1[3];
        }
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[e][q >> 1][q & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool whole = seg_c0 == 0 && c_chunk == nchunks - 1;
    int job0 = -1, job1 = -1;
    unsigned old0 = 0;
    if (whole) {
      job0 = c_item;
    } else if (!(ABLATE & 1024)) {
      const unsigned my_slot = 2u * lg + (seg_c0 == 0 ? 1u : 0u);
#pragma unroll
      for (int q = 0; q < 16; q++)
        slab_store16(y[q >> 3][(q >> 2) & 1][q & 3], rsrc_slab, slab_voff + q * 1024, my_slot * SLAB_BYTES);
      if (!last_of_range) {
        pend_item = c_item;
      } else {
        wait_vmem_all();
        old0 = draw_ticket(c_item);
        job0 = c_item;
      }
    }
    if (pend_item >= 0 && pend_item != c_item) {
      job1 = pend_item;
      pend_item = -1;
    }
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
      const int item = j == 0 ? job0 : job1;
      if (item < 0) continue;
      if (!(j == 0 && whole)) {
        const unsigned x0 = (unsigned)(item - tail_item0) * (unsigned)nchunks, x1 = x0 + nchunks - 1;
        int gA = lg, gB = lg;
        while (sk_start(gA, sk_q, sk_rem, G) > x0) gA--;
        while (gB + 1 < G && sk_start(gB + 1, sk_q, sk_rem, G) <= x1) gB++;
        int nseg = 0;
        for (int g = gA; g <= gB; g++)
          nseg += sk_start(g + 1, sk_q, sk_rem, G) != sk_start(g, sk_q, sk_rem, G);
        const unsigned old = __builtin_amdgcn_readfirstlane(j == 0 ? old0 : pend_old);
        if (old != (unsigned)(nseg - 1)) continue;
        if (ln == 0)
          __hip_atomic_store(tickets + (size_t)item * 8 + wv, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool first = true;
#pragma unroll 1
        for (int g = gA; g <= gB; g++) {
          if (sk_start(g + 1, sk_q, sk_rem, G) == sk_start(g, sk_q, sk_rem, G)) continue;
          const unsigned slot = 2u * (unsigned)g + (first ? 1u : 0u);
#pragma unroll
          for (int half = 0; half < 2; half++) {   // 8 loads in flight at a time
            f32x4 t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = slab_load16(rsrc_slab, slab_voff + (half * 8 + q) * 1024, slot * SLAB_BYTES);
#pragma unroll
            for (int q = 0; q < 8; q++) {
              f32x4& yy = y[half][(q >> 2) & 1][q & 3];
              yy = first ? t[q] : yy + t[q];
            }
          }
          first = false;
        }
      }
      float sc2[2] = {bn_sc[0], bn_sc[1]}, bi2[2] = {bn_bi[0], bn_bi[1]};
      if (item != c_item) load_bn(item, sc2, bi2);
      if (ABLATE & 512) continue;

      // finalize: BN + ReLU, the wave's 32 tiles x 2x2 px x 32 k through its 16 KiB of LDS
      // ([tile 0..31][px 0..3][k 0..31] floats, 16-float group (2*px + kt) XORed with the MFMA row
      // group h = (tile>>2)&3), out as whole 128-byte runs
      const int tb = item / KBLK, kb = item - tb * KBLK;
      int ep_wbase[4], ep_rbase[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        ep_wbase[jj] = e_h * 2048 + ((jj ^ e_h) << 6) + e_t16 * 4;                       // + tg*8192 + r*512 + (g>>2)*256
        const int px = (ln >> 3) & 3, c = ln & 7;
        ep_rbase[jj] = (ln >> 5) * 512 + (((px * 2 + (c >> 2)) ^ jj) << 6) + (c & 3) * 16;   // + 2i*512
      }
#pragma unroll
      for (int tg = 0; tg < 2; tg++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {
              float v1 = sc2[kt] * y[tg][kt][r][pp] + bi2[kt];
              if (relu) v1 = fmaxf(v1, 0.f);
              const int g = pp * 2 + kt;
              *(float*)(wreg + ep_wbase[g & 3] + tg * 8192 + r * 512 + (g >> 2) * 256) = v1;
            }
      const int px = (ln >> 3) & 3, pa = px >> 1, pb = px & 1;
      const unsigned kbyte = (unsigned)((kb * KB + e_wk * 32 + (ln & 7) * 4) * sizeof(float));
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const f32x4 val = *(const f32x4*)(wreg + ep_rbase[(i >> 1) & 3] + i * 1024);
        const int g = tb * TB + e_wt * 32 + 2 * i + (ln >> 5);
        const bool live = g < totalTiles;
        const TileCoord tc = decode_tile(live ? g : 0);
        const int py = 1 + 2 * tc.ty + pa, pxx = 1 + 2 * tc.tx + pb;
        const unsigned img = (unsigned)(tc.n * WINO_HW * WINO_HW);
        if (live) buf_store16(val, rsrc_out, (unsigned)((img + py * WINO_HW + pxx) * K * sizeof(float)) + kbyte, 0);
      }
    }
  };

  // ================================ main loop =====================================
  {
    int us = 0;
    auto next = [](int s) { return s == 2 ? 0 : s + 1; };
    int it = 0;
#pragma unroll 1
    for (;;) {
      const int n = c_tail > 0 && c_tail < nchunks - c_chunk ? c_tail : nchunks - c_chunk;
      c_chunk += n - 1;
      int us_last = us;
#pragma unroll 1
      for (int k = 0; k < n; k++) {
        body(it, it & 1, us, next(us), next(next(us)));
#pragma unroll
        for (int p = 0; p < 8; p++) { a_lo[0][p] ^= RAW_BYTES; a_lo[1][p] ^= RAW_BYTES; }
        if (it + 2 < L) dma_advance();
        us_last = us;
        us = next(us);
        it++;
      }
      const bool last_of_range = it == L;
      if (ABLATE & 2048) { const unsigned long long t = stamp(); st_comp += t - st_prev; st_prev = t; }
      epilogue(last_of_range, (it & 1) * RAW_BYTES, N_RSTAGE * RAW_BYTES + us_last * U_BYTES);
      if (ABLATE & 2048) { const unsigned long long t = stamp(); st_epi += t - st_prev; st_prev = t; }
      if (last_of_range) break;
      if (c_tail > 0 && (c_tail -= n) == 0) c_item = lg;
      else c_item += c_tail > 0 ? 1 : G;
      c_chunk = 0;
      seg_c0 = 0;
    }
  }
#undef A_OFF4
#undef PIN2

  if (ABLATE & 2048) {
    if (lane == 0) {
      unsigned long long* dbg = prm.dbg +
                                ((size_t)lg * 8 + w) * 8;
      dbg[0] = st_wait;
      dbg[1] = st_comp;
      dbg[2] = st_epi;
      dbg[3] = dbg[4] = dbg[5] = dbg[6] = 0;
    }
  }
  if (ABLATE & 16) {
    stamp_c += __builtin_amdgcn_s_memtime();
    stamp_r += __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      unsigned long long* dbg = prm.dbg + (size_t)lg * 2;
      dbg[0] = stamp_c;
      dbg[1] = stamp_r;
    }
  }
}

}  // namespace fused4
}  // namespace wino
