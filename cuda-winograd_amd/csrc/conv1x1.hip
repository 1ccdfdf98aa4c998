// 1x1 convolution as an fp32 MFMA GEMM with fused folded-BN (+ReLU) epilogue, gfx950.
//
// Replaces kernel_512_one_128 / kernel_128_one_512 (Kernel128_one.cu:24-54,244-273) and
// kernel_1024_one_256 / kernel_256_one_1024 (Kernel256_one.cu:26-56,246-274):
//   C[m][k] = act( bnScale[k] * sum_c A[m][c] * B[c][k] + bnBias[k] )
// with A [M][Cin] (pixels x in-channels, the reference's HWC-flat activations),
// B [Cin][Kout] row-major exactly as the reference stores it (Kernel128_one.cu:40-42).
//
// Tiling: workgroup = NW waves (8, or 4: see four_waves() below) computes BM=112 x BN=16*NW of C; the
// K-loop runs over Cin in steps of BK = 32 through two LDS stages (60 / 44 KB, so 2-3 workgroups
// share a CU) filled by LDS-DMA through buffer descriptors (conv1x1_kernel.h).  BM = 7 MFMA row
// blocks because the reference's M = N*196 = 2^a * 49: 112-row tiles cover it exactly (25088 =
// 224 * 112).  Wave w owns columns [16w, 16w+16): 7 accumulator tiles of v_mfma_f32_16x16x4_f32
// (28 acc VGPRs); the A fragment of a step is shared by 4 MFMAs.  Shapes whose tile count does not
// fill the CUs evenly are launched in stream-K / split-K form (sk1_grid below).
//
// LDS images (16-byte units XOR-permuted on the DMA source side so that fragment reads are
// bank-conflict free):
//   A stage [112 rows][BK/4 units]: unit' = unit ^ f(row); fragments by ds_read_b128
//           f(row) = row & 15 (BK = 64: one 256-B bank row per A row), (row>>1) & 7 (BK = 32)
//   B stage [BK k][32 units]      : unit' = unit ^ (4*((k>>2)&1)); fragments by ds_read_b32
// The point loop is pinned (sched_barrier) with LDS requests two steps ahead of their use and
// counted lgkmcnt waits, and the LDS-DMA pieces are issued one per step: see the notes in
// wino_f2_fused_kernel.h, the same three hipcc behaviours apply here.
#include "conv1x1_kernel.h"
#include "conv1x1_small_kernel.h"

#include <atomic>

namespace wino {
namespace {

using namespace gemm1x1;

// Comparator: one thread per output element, plain fp32 FMA loop over Cin.
__global__ void conv1x1_direct_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                      const float* __restrict__ bnBias,
                                      const float* __restrict__ bnScale, float* __restrict__ Cout,
                                      long M, int Cin, int Kout, int relu) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * Kout) return;
  const long m = idx / Kout;
  const int k = (int)(idx - m * Kout);
  const float* a = A + m * Cin;
  float s = 0.f;
  for (int c = 0; c < Cin; c++) s = fmaf(a[c], B[(size_t)c * Kout + k], s);
  float y = bnScale[k] * s + bnBias[k];
  if (relu) y = fmaxf(y, 0.f);
  Cout[idx] = y;
}

}  // namespace
}  // namespace wino

using namespace wino;

// 4-wave workgroups (64 columns) or 8-wave (128 columns).  Four when Kout is small: twice the
// workgroups, so that at the reference's Kout = 128 every CU holds two of them; when Cin is small
// (few k-steps per workgroup: 128->512 measured 35.1 vs 37.1 us; otherwise equal to 8 waves within
// 1 %); and whenever Kout is not a multiple of 128 -- the API takes any multiple of 64, and a
// 128-column block would leave the last 64 columns uncomputed.
// Cin % 32 (the k-step), Kout % 64 (the narrowest column block); B is addressed through one buffer
// descriptor (32-bit byte offsets)
static inline bool bad_1x1_dims(int Cin, int Kout) {
  return Cin <= 0 || Kout <= 0 || (Cin % 32) != 0 || (Kout % 64) != 0 ||
         (unsigned long long)Cin * (unsigned long long)Kout * sizeof(float) >= (1ull << 32);
}
static inline bool four_waves(int Cin, int Kout) { return Kout <= 128 || Cin <= 128 || (Kout % 128) != 0; }

// Stream-K grid for `tiles` output tiles of nk k-steps on `cus` CUs, or 0 for the plain
// one-tile-per-workgroup launch.  A launch lasts as long as its busiest CU (conv1x1_kernel.h): the
// plain form puts r = ceil(tiles / cus) whole tiles on it, stream-K x = tiles * nk / cus k-steps
// plus a hand-over per range, at the end of the range, on the critical path.  Launch times in us
// fitted to measurements on MI355X (the four reference layers and 2048->512, 768->192, N = 1..256,
// K loops of 8..64 steps, 1..4 rounds, G = cus and 2 cus; reproduced to 1-3 %; refitted after the
// loop of this kernel lost a quarter of its instructions):
//                        8 waves (128 columns)               4 waves (64 columns)
//   plain                2.7 + r (1.555 nk + 1.75)           3.8 + r (0.778 nk + 0.7)
//   stream-K, G = cus    12 + x (1.60 + 1.75 / nk)           9.5 + x (0.89 + 0.7 / nk)
//   stream-K, G = 2 cus  13 + x (1.558 + 1.75 / nk)          11 + x (0.816 + 0.7 / nk)
// (the 1.75 / 0.7 is a whole tile's epilogue; two ranges per CU run the steps a little faster and pay
// a second hand-over).  The cheapest predicted form wins; stream-K must come in under 0.99 of plain.
// What that buys at the reference shapes: 1024->256 N = 128 106 -> 102 us, N = 80..120 105 -> 71..99,
// N = 160 156 -> 127; 512->128 N = 80 30 -> 26, N = 160 43 -> 40, N = 128 stays plain.
//
// Tiny problems -- too few k-steps to give every CU a range, the reference's own N = 1 protocol --
// leave CUs idle in any form; split-K over short ranges (a tile's segments gathered in k order by
// whoever arrives last): 1024->256 54 -> 19-25 us for N <= 16, 512->128 16-17 -> 11.5-13 us,
// 256->1024 17 -> 13.5 us at N = 1; 128->512 (4 steps) always loses.  See the model in sk1_grid.
//
// Developer overrides (wino_common.h Knobs: read once, wino_debug_reload_knobs() re-reads): WINO_1X1_SK=0 / 1
// forces the plain / stream-K form (1: whenever a legal grid exists), WINO_1X1_SK_GRID=G sets the
// number of ranges (rounded down to a multiple of 8 and of the column blocks, at most one range
// per k-step).
constexpr int SK1_MIN_STEPS = 4;
constexpr long long SK1_MAX_GRID = 16384;   // 2 * G slabs of <= 56 KB must stay below the 4 GiB a buffer descriptor spans
struct Sk1Model { double a_plain, t_plain, e_tile, a_sk1, t_sk1, a_sk2, t_sk2; };
constexpr Sk1Model SK1_MODEL_8W = {2.7, 1.555, 1.75, 12.0, 1.60, 13.0, 1.558};
constexpr Sk1Model SK1_MODEL_4W = {3.8, 0.778, 0.7, 9.5, 0.89, 11.0, 0.816};
static int sk1_grid(long long tiles, int nk, int cus, int nblk, bool four_wave_form, double* t_pred = nullptr) {
  const Knobs kn = knobs();
  const int force = kn.sk_1x1;
  {   // the plain form's predicted time: what stands when no stream-K / split-K grid is taken
    const Sk1Model& m0 = four_wave_form ? SK1_MODEL_4W : SK1_MODEL_8W;
    const long long rounds0 = cus > 0 ? (tiles + cus - 1) / cus : 1;
    if (t_pred) *t_pred = m0.a_plain + (double)rounds0 * (m0.t_plain * nk + m0.e_tile);
  }
  if (force == 0 || cus < 8 || tiles < 1) return 0;
  const long long U = tiles * nk;
  // G is a multiple of 8 (whole XCD groups) and of the column blocks per row tile (a range is
  // run by one workgroup per column block)
  long long step = 8;
  while (step % nblk) step += 8;
  if (kn.sk_1x1_grid) {
    long long G = kn.sk_1x1_grid;
    if (G > U) G = U;
    if (G > SK1_MAX_GRID) G = SK1_MAX_GRID;
    G -= G % step;
    return G >= step ? (int)G : 0;
  }
  const Sk1Model& m = four_wave_form ? SK1_MODEL_4W : SK1_MODEL_8W;
  {
    // Tiny problems: split-K over ranges of s k-steps.  A tile's nk / s segments are gathered one
    // after the other by whoever arrives last, so  T = a + s t + (nk / s) c  with, per segment,
    // c = 0.9 us (8 waves, 56 KB slabs) or 0.47 us (4 waves, 28 KB) and a = 6.7 / 7.2 us, fitted at
    // N = 1..4 (1024->256: s = 8 / 4 / 2 gives 23.1 / 19.1 / 21.9 us; 256->1024: s = 4 / 2 / 1 gives
    // 14.7 / 13.4 / 16.8; 2048->64: s = 16 / 4 gives 21.5 / 17.8).  Its optimum s* = sqrt(c nk / t) is
    // 4 steps for the 32-step layers, 2 for 256->1024's 8, 3-6 for the 4-wave layers.
    const double c_seg = four_wave_form ? 0.47 : 0.9, a_tiny = four_wave_form ? 7.2 : 6.7;
    int s_opt = (int)(__builtin_sqrt(c_seg * nk / m.t_plain) + 0.5);
    // (the model is flat near its optimum and has no term for the total slab traffic, which grows
    //  with the tile count: 512->128 at N = 4 measured 11.7 us with 4-step ranges, 12.6 with 3)
    if (s_opt < (nk >= 16 ? 4 : 2)) s_opt = nk >= 16 ? 4 : 2;
    if (s_opt > nk) s_opt = nk;
    if (U / s_opt < cus) {   // fewer such ranges than CUs: the general forms below do not apply
      long long G = U / s_opt;
      G -= G % step;
      if (G < step || G <= tiles) return 0;
      if (force == 1) return (int)G;
      const double s_eff = (double)U / (double)G;
      const double t_sk = a_tiny + s_eff * m.t_plain + ((double)nk / s_eff) * c_seg;
      const double t_plain = m.a_plain + m.t_plain * nk + m.e_tile;
      if (t_sk < 0.9 * t_plain) {
        if (t_pred) *t_pred = t_sk;
        return (int)G;
      }
      return 0;
    }
  }
  const long long rounds = (tiles + cus - 1) / cus;
  const double x = (double)U / (double)cus;
  const double t_plain = m.a_plain + (double)rounds * (m.t_plain * nk + m.e_tile);
  long long G1 = cus, G2 = 2ll * cus;
  G1 -= G1 % step;
  G2 -= G2 % step;
  if (G2 > SK1_MAX_GRID || U / G2 < SK1_MIN_STEPS) G2 = 0;
  if (G1 < step) return 0;
  const double t_sk1 = m.a_sk1 + x * (m.t_sk1 + m.e_tile / nk);
  const double t_sk2 = G2 ? m.a_sk2 + x * (m.t_sk2 + m.e_tile / nk) : 1e30;
  const long long G = t_sk2 < t_sk1 ? G2 : G1;
  if (force == 1) return (int)G;
  if ((t_sk2 < t_sk1 ? t_sk2 : t_sk1) < 0.99 * t_plain) {
    if (t_pred) *t_pred = t_sk2 < t_sk1 ? t_sk2 : t_sk1;
    return (int)G;
  }
  return 0;
}

// The latency form (conv1x1_small_kernel.h): blocks of (16 RT) x (16 CT) outputs held by one wave, 4 waves per
// workgroup, a block's K loop split over KS of them.  For layers with few pixel rows (plain or chained: padded operands,
// residual).  Every wave pulls its own operands through its CU's vector memory path and that path bounds the form.
// Per workgroup (its 4 / KS blocks sit side by side): A = (4 / KS) * RT * 16 * Cin * 4 bytes of pixel rows (shared by
// every column group, i.e. by all XCDs: fabric-side traffic), B = (4 / KS) * CT * 16 * Cin * 4 bytes of filter columns
// (one XCD per column group: L2 hits), C = the block's outputs.  Kernel times of every legal form at 1 .. 24 images of the
// four reference layers (tools/latency_cases.py explore1, profiles/r3/latency_explore_1x1_forms.json, 506 points) fit
//     T = 2.16 us + depth * (0.0281 us/KB * A + 0.0141 (CT = 4: 0.0171) us/KB * B + 0.098 us/KB * C + 0.43 us),
// depth = ceil(workgroups / CUs), to 13 % (least squares on the relative error); it underestimates launches several
// workgroups deep, so those must beat the tiled kernel's own launch model (sk1_grid) by 20 %, one-deep launches by 7 %.  Picking by it costs 1.1 % on average against the best measured choice of both kernels at each of those points.
// M = 196: 1024->256 5.4 us against the tiled kernel's 19.3, 512->128 3.8 / 10.5, 128->512 3.4 / 6.7, 256->1024 4.7 / 14.4;
// 8 images: 14.6 / 22.1, 6.7 / 12.6, 7.4 / 8.6, 14.7 / 19.1 (32 x 64 blocks with 16-byte filter loads).
// WINO_1X1_ALGO=big|small and WINO_1X1_SMALL_KS / _RT / _CT override.
struct Small1Plan {
  bool use;
  int ks, rt, ct;
  long long wgs;
  double t_us, t_big_us;   // the two launch models' times
};
static bool small1_legal(int Cin, int Kout, int ks, int rt, int ct) {
  (void)rt;
  return Cin % (16 * ks) == 0 && Kout % ((4 / ks) * ct * 16) == 0;
}
static double small1_time(long M, int Cin, int Kout, int cus, int ks, int rt, int ct, long long* wgs) {
  const long long rows = (M + 16 * rt - 1) / (16 * rt), cols = Kout / ((4 / ks) * ct * 16);
  *wgs = rows * cols;
  // per workgroup: pixel KB, filter KB, output KB (one wave-quarter of each per wave)
  const double a_kb = (double)(4 / ks) * rt * 16.0 * Cin * 4.0 / 1e3, b_kb = (double)(4 / ks) * ct * 16.0 * Cin * 4.0 / 1e3;
  const double c_kb = 16.0 * rt * (4 / ks) * ct * 16.0 * 4.0 / 1e3;
  const long long deep = (*wgs + cus - 1) / cus;
  return 2.16 + (double)deep * (0.0281 * a_kb + (ct == 4 ? 0.0171 : 0.0141) * b_kb + 0.098 * c_kb + 0.43);
}
static Small1Plan small1_plan(long M, int Cin, int Kout, int flags, int batch, int cus) {
  Small1Plan pl = {false, 1, 1, 1, 0, 0.0, 0.0};
  if (batch != 1 || M < 1) return pl;   // (every flag of the tiled kernel travels; batched launches do not)
  (void)flags;
  const Knobs kn = knobs();
  double best = 1e30;
  for (int rt = 1; rt <= 2; rt++)
    for (int ct = 1; ct <= 4; ct *= 2)
      for (int ks = 4; ks >= 1; ks >>= 1) {
        if (!small1_legal(Cin, Kout, ks, rt, ct)) continue;
        if (Cin / ks < 64 && ks > 1) continue;            // shorter K loops per wave are all overhead
        if ((kn.small_ks == 1 || kn.small_ks == 2 || kn.small_ks == 4) && ks != kn.small_ks) continue;
        if ((kn.small_rt == 1 || kn.small_rt == 2) && rt != kn.small_rt) continue;
        if ((kn.small_ct == 1 || kn.small_ct == 2 || kn.small_ct == 4) && ct != kn.small_ct) continue;
        long long wgs = 0;
        const double t = small1_time(M, Cin, Kout, cus, ks, rt, ct, &wgs);
        const long long rows = (M + 16 * rt - 1) / (16 * rt);
        if (rows > 65535) continue;                        // gridDim.y
        if (t < best) { best = t; pl.ks = ks; pl.rt = rt; pl.ct = ct; pl.wgs = wgs; pl.t_us = t; }
      }
  if (best > 1e29) return pl;
  {
    double t_big = 0.0;
    const bool four = four_waves(Cin, Kout);
    const int bn = four ? 64 : 128;
    (void)sk1_grid(((M + BM - 1) / BM) * (long long)(Kout / bn), Cin / 32, cus, Kout / bn, four, &t_big);
    pl.t_big_us = t_big;
    pl.use = best < (pl.wgs > cus ? 0.8 : 0.93) * t_big;   // (1024->256 at 40 images: modelled 38.5 / 41.0 us, measured 48.5 / 44.4)
  }
  if (kn.sk_1x1 != -1 || kn.sk_1x1_grid != 0) pl.use = false;   // a developer is forcing a form of the tiled kernel
  if (kn.algo_1x1 == 1) pl.use = false;
  if (kn.algo_1x1 == 2) pl.use = true;
  return pl;
}
template <int RT, int CT>
static void launch_1x1_small_ks(int ks, dim3 grid, hipStream_t s, const float* A, const float* B, const float* bnBias,
                                const float* bnScale, const float* R, float* C, long M, int Cin, int Kout, int flags, PadGeo pg) {
  const dim3 block(256);
  if (ks == 4) hipLaunchKernelGGL((conv1x1_small_kernel<4, RT, CT>), grid, block, 0, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  else if (ks == 2) hipLaunchKernelGGL((conv1x1_small_kernel<2, RT, CT>), grid, block, 0, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  else hipLaunchKernelGGL((conv1x1_small_kernel<1, RT, CT>), grid, block, 0, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
}
static int launch_1x1_small(const Small1Plan& pl, const float* A, const float* B, const float* bnBias,
                            const float* bnScale, const float* R, float* C, long M, int Cin, int Kout, int flags, PadGeo pg,
                            hipStream_t s) {
  // x = column group, y = row block: see the kernel
  const dim3 grid((unsigned)(Kout / ((4 / pl.ks) * pl.ct * 16)), (unsigned)((M + 16 * pl.rt - 1) / (16 * pl.rt)));
  if (pl.rt == 2 && pl.ct == 4) launch_1x1_small_ks<2, 4>(pl.ks, grid, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  else if (pl.ct == 4) launch_1x1_small_ks<1, 4>(pl.ks, grid, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  else if (pl.rt == 2 && pl.ct == 2) launch_1x1_small_ks<2, 2>(pl.ks, grid, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  else if (pl.rt == 2) launch_1x1_small_ks<2, 1>(pl.ks, grid, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  else if (pl.ct == 2) launch_1x1_small_ks<1, 2>(pl.ks, grid, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  else launch_1x1_small_ks<1, 1>(pl.ks, grid, s, A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, pg);
  return launch_status("conv1x1_small_kernel");
}

template <int BK, int NW, bool RES = false>
static int launch_1x1_res(const float* A, const float* B, const float* bnBias, const float* bnScale,
                          const float* R, float* C, long M, int Cin, int Kout, int flags, int nMB,
                          hipStream_t s, int batch, long batchA, long batchB, long batchC,
                          bool prepare_only, PadGeo pg);

template <int BK, int NW>
static int launch_1x1(const float* A, const float* B, const float* bnBias, const float* bnScale,
                      const float* R, float* C, long M, int Cin, int Kout, int flags, int nMB,
                      hipStream_t s, int batch = 1, long batchA = 0, long batchB = 0, long batchC = 0,
                      bool prepare_only = false, PadGeo pg = make_padgeo(WINO_PQ, WINO_PQ)) {
  if (flags & WINO_ADD_RESIDUAL)
    return launch_1x1_res<BK, NW, true>(A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, nMB, s, batch, batchA, batchB, batchC, prepare_only, pg);
  return launch_1x1_res<BK, NW, false>(A, B, bnBias, bnScale, R, C, M, Cin, Kout, flags, nMB, s, batch, batchA, batchB, batchC, prepare_only, pg);
}

template <int BK, int NW, bool RES>
static int launch_1x1_res(const float* A, const float* B, const float* bnBias, const float* bnScale,
                          const float* R, float* C, long M, int Cin, int Kout, int flags, int nMB,
                          hipStream_t s, int batch, long batchA, long batchB, long batchC,
                          bool prepare_only, PadGeo pg) {
  using G = Cfg<BK, NW>;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  WINO_HIP(hipGetDevice(&dev));
  if (!((attr_done.load() >> (dev & 63)) & 1ull)) {
    WINO_HIP(hipFuncSetAttribute((const void*)(conv1x1_bn_kernel<BK, NW, 0, false, RES>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    WINO_HIP(hipFuncSetAttribute((const void*)(conv1x1_bn_kernel<BK, NW, 0, true, RES>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_done.fetch_or(1ull << (dev & 63));
  }
  int cus = 0;
  if (int rc = device_cus(dev, &cus)) return rc;
  const long long tiles = (long long)nMB * (Kout / G::BN);
  const int Gsk = batch == 1 ? sk1_grid(tiles, Cin / BK, cus, Kout / G::BN, NW == 4) : 0;
  if (Gsk) {
    SkBufs bufs;
    if (int rc = sk_scratch(dev, s, (size_t)2 * Gsk * NW * RB * 1024, (size_t)tiles, &bufs)) return rc;
    if (prepare_only) return WINO_OK;
    const SkArgs sk{bufs.slabs, bufs.tickets, nullptr, bufs.err};
    hipLaunchKernelGGL((conv1x1_bn_kernel<BK, NW, 0, true, RES>), dim3(Gsk), dim3(G::NT), G::LDS_BYTES, s, A, B,
                       bnBias, bnScale, R, C, M, Cin, Kout, flags, nMB, 0L, 0L, 0L, sk, pg);
    const int rc = launch_status("conv1x1_bn_kernel (stream-K)");
    if (rc) sk_mark_failed(dev, s);   // the launch held the stream's scratch
    return rc;
  }
  if (prepare_only) return WINO_OK;
  const int grid = 8 * (Kout / G::BN) * ((nMB + 7) / 8);
  hipLaunchKernelGGL((conv1x1_bn_kernel<BK, NW, 0, false, RES>), dim3(grid, batch), dim3(G::NT), G::LDS_BYTES, s, A, B,
                     bnBias, bnScale, R, C, M, Cin, Kout, flags, nMB, batchA, batchB, batchC, SkArgs{nullptr, nullptr, nullptr, nullptr}, pg);
  return launch_status("conv1x1_bn_kernel");
}

namespace wino {
int last_clock_1x1(unsigned long long* stamps) {
  WINO_HIP(hipMemcpyFromSymbol(stamps, HIP_SYMBOL(wino::gemm1x1::wino_clk_slot_1x1), 4 * sizeof(unsigned long long)));
  return WINO_OK;
}
// Batched plain GEMM C_b = A_b . B_b (no BN) on the 1x1 kernel: used by the F(4x4) compatibility path.
int gemm_batched(const float* A, const float* B, float* C, long M, int Cin, int Kout, int batch,
                 long batchA, long batchB, long batchC, hipStream_t s) {
  if (bad_1x1_dims(Cin, Kout) || M < 1 || batch < 1 || batch > 65535) {
    set_error("unsupported batched GEMM shape");
    return WINO_E_SHAPE;
  }
  const int nMB = (int)((M + BM - 1) / BM);
  if (four_waves(Cin, Kout))
    return launch_1x1<32, 4>(A, B, nullptr, nullptr, nullptr, C, M, Cin, Kout, gemm1x1::WINO_INTERNAL_NO_BN, nMB, s,
                             batch, batchA, batchB, batchC);
  return launch_1x1<32, 8>(A, B, nullptr, nullptr, nullptr, C, M, Cin, Kout, gemm1x1::WINO_INTERNAL_NO_BN, nMB, s,
                           batch, batchA, batchB, batchC);
}
}  // namespace wino

// Shared body of wino_conv1x1_bn_ex (H = W = 14) and wino_conv1x1_bn_ex_hw.
static int conv1x1_ex(const float* A, const float* B, const float* bnBias, const float* bnScale,
                      const float* residual, float* C, long M, int H, int W, int Cin, int Kout, int flags,
                      wino_stream_t s) {
  if (!A || !B || !bnBias || !bnScale || !C) { set_error("NULL pointer"); return WINO_E_ARG; }
  if ((flags & WINO_ADD_RESIDUAL) && !residual) { set_error("WINO_ADD_RESIDUAL without residual"); return WINO_E_ARG; }
  if (misaligned16(A, B, C, residual)) { set_error("tensor pointers must be 16-byte aligned"); return WINO_E_ARG; }
  if (flags & ~(WINO_RELU | WINO_A_PADDED | WINO_C_PADDED | WINO_ADD_RESIDUAL)) { set_error("unknown flag bits 0x%x", flags); return WINO_E_ARG; }
  if (M < 1 || bad_1x1_dims(Cin, Kout)) {
    set_error("unsupported 1x1 shape M=%ld Cin=%d Kout=%d (need Cin %% 32 == 0, Kout %% 64 == 0)",
              M, Cin, Kout);
    return WINO_E_SHAPE;
  }
  const long nMBl = (M + BM - 1) / BM;
  if (nMBl > (1L << 24)) { set_error("M too large"); return WINO_E_SHAPE; }
  const int nMB = (int)nMBl;
  PadGeo pg = make_padgeo(WINO_PQ, WINO_PQ);
  if (flags & (WINO_A_PADDED | WINO_C_PADDED)) {
    if (H < 1 || W < 1 || H > 4094 || W > 4094) { set_error("unsupported feature map %dx%d", H, W); return WINO_E_SHAPE; }
    if ((M % ((long)H * W)) != 0) {
      set_error("padded layouts need M = N*%d*%d, got M=%ld", H, W, M);
      return WINO_E_SHAPE;
    }
    // the ring pass counts 16-byte units in 32 bits
    const unsigned long long ring_units = (unsigned long long)(M / ((long)H * W)) * (2ull * (W + 2) + 2ull * H) * (Kout / 4);
    if (ring_units >= (1ull << 32)) { set_error("padded output too large for one launch"); return WINO_E_SHAPE; }
    pg = make_padgeo(H, W);
  }
  {
    int dev = 0, cus = 0;
    WINO_HIP(hipGetDevice(&dev));
    if (int rc = device_cus(dev, &cus)) return rc;
    const Small1Plan sp = small1_plan(M, Cin, Kout, flags, 1, cus);
    if (sp.use) return launch_1x1_small(sp, A, B, bnBias, bnScale, residual, C, M, Cin, Kout, flags, pg, (hipStream_t)s);
  }
  // BK = 32 keeps a workgroup at 60 KB of LDS, so two workgroups share a CU (4 waves per SIMD)
  // and one's prologue / barrier bubbles / store tail hide under the other's MFMAs; measured
  // 3-14 % faster than BK = 64 (120 KB, one workgroup per CU) on the four reference shapes.
  if (four_waves(Cin, Kout))
    return launch_1x1<32, 4>(A, B, bnBias, bnScale, residual, C, M, Cin, Kout, flags, nMB, (hipStream_t)s, 1, 0, 0, 0, false, pg);
  return launch_1x1<32, 8>(A, B, bnBias, bnScale, residual, C, M, Cin, Kout, flags, nMB, (hipStream_t)s, 1, 0, 0, 0, false, pg);
}

extern "C" {

int wino_conv1x1_bn_ex(const float* A, const float* B, const float* bnBias, const float* bnScale,
                       const float* residual, float* C, long M, int Cin, int Kout, int flags,
                       wino_stream_t s) {
  return conv1x1_ex(A, B, bnBias, bnScale, residual, C, M, WINO_PQ, WINO_PQ, Cin, Kout, flags, s);
}

int wino_conv1x1_bn_ex_hw(const float* A, const float* B, const float* bnBias, const float* bnScale,
                          const float* residual, float* C, int N, int H, int W, int Cin, int Kout,
                          int flags, wino_stream_t s) {
  if (N < 1 || H < 1 || W < 1) { set_error("bad N=%d H=%d W=%d", N, H, W); return WINO_E_SHAPE; }
  return conv1x1_ex(A, B, bnBias, bnScale, residual, C, (long)N * H * W, H, W, Cin, Kout, flags, s);
}

// Host-side only: the launch form this shape takes on a device with `cus` compute units.
int wino_conv1x1_plan(long M, int Cin, int Kout, int cus, int* grid, int* row_tiles, int* col_blocks,
                      int* k_steps, int* stream_k) {
  if (!grid || !row_tiles || !col_blocks || !k_steps || !stream_k || cus < 1) { set_error("bad argument"); return WINO_E_ARG; }
  if (M < 1 || bad_1x1_dims(Cin, Kout)) {
    set_error("unsupported 1x1 shape M=%ld Cin=%d Kout=%d (need Cin %% 32 == 0, Kout %% 64 == 0)",
              M, Cin, Kout);
    return WINO_E_SHAPE;
  }
  const long nMBl = (M + BM - 1) / BM;
  if (nMBl > (1L << 24)) { set_error("M too large"); return WINO_E_SHAPE; }
  const int bn = four_waves(Cin, Kout) ? 64 : 128;   // as wino_conv1x1_bn_ex
  const int nblk = Kout / bn;
  const int G = sk1_grid((long long)nMBl * nblk, Cin / 32, cus, nblk, four_waves(Cin, Kout));
  *row_tiles = (int)nMBl;
  *col_blocks = nblk;
  *k_steps = Cin / 32;
  *stream_k = G != 0;
  *grid = G ? G : 8 * nblk * (int)((nMBl + 7) / 8);
  return WINO_OK;
}

// Host-side only: does a PLAIN layer of this shape take the latency form (conv1x1_small_kernel.h) on a device
// with `cus` CUs: *use, the K-split inside a workgroup and the number of workgroups.
int wino_conv1x1_small_plan(long M, int Cin, int Kout, int cus, int* use, int* k_split, int* workgroups) {
  int rt = 0, ct = 0;
  return wino_conv1x1_small_plan2(M, Cin, Kout, cus, use, k_split, &rt, &ct, workgroups);
}

int wino_conv1x1_small_plan2(long M, int Cin, int Kout, int cus, int* use, int* k_split, int* row_tiles, int* col_tiles,
                             int* workgroups) {
  if (!use || !k_split || !row_tiles || !col_tiles || !workgroups || cus < 1) { set_error("bad argument"); return WINO_E_ARG; }
  if (M < 1 || bad_1x1_dims(Cin, Kout)) {
    set_error("unsupported 1x1 shape M=%ld Cin=%d Kout=%d (need Cin %% 32 == 0, Kout %% 64 == 0)", M, Cin, Kout);
    return WINO_E_SHAPE;
  }
  const Small1Plan pl = small1_plan(M, Cin, Kout, 0, 1, cus);
  *use = pl.use;
  *k_split = pl.ks;
  *row_tiles = pl.rt;
  *col_tiles = pl.ct;
  *workgroups = pl.use ? (int)pl.wgs : 0;
  return WINO_OK;
}

int wino_debug_conv1x1_models(long M, int Cin, int Kout, int cus, double* t_latency_us, double* t_tiled_us) {
  if (!t_latency_us || !t_tiled_us || cus < 1 || M < 1 || bad_1x1_dims(Cin, Kout)) { set_error("bad argument"); return WINO_E_ARG; }
  const Small1Plan pl = small1_plan(M, Cin, Kout, 0, 1, cus);
  *t_latency_us = pl.t_us;
  *t_tiled_us = pl.t_big_us;
  return WINO_OK;
}

// Allocates the stream-K scratch this shape's launches on stream `s` will use (nothing for shapes
// that take the plain form): call it before capturing wino_conv1x1_bn(_ex) into a HIP graph.
int wino_conv1x1_prepare(long M, int Cin, int Kout, wino_stream_t s) {
  if (M < 1 || bad_1x1_dims(Cin, Kout)) {
    set_error("unsupported 1x1 shape M=%ld Cin=%d Kout=%d (need Cin %% 32 == 0, Kout %% 64 == 0)",
              M, Cin, Kout);
    return WINO_E_SHAPE;
  }
  const long nMBl = (M + BM - 1) / BM;
  if (nMBl > (1L << 24)) { set_error("M too large"); return WINO_E_SHAPE; }
  const int nMB = (int)nMBl;
  {   // (a layer small enough for the latency form uses no scratch; the tiled kernel's is allocated below all the same:
      //  a developer knob may still send the launch there)
    int dev = 0, cus = 0;
    WINO_HIP(hipGetDevice(&dev));
    if (int rc = device_cus(dev, &cus)) return rc;
    (void)cus;
  }
  if (four_waves(Cin, Kout))
    return launch_1x1<32, 4>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, Cin, Kout, 0, nMB, (hipStream_t)s, 1, 0, 0, 0, true);
  return launch_1x1<32, 8>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, Cin, Kout, 0, nMB, (hipStream_t)s, 1, 0, 0, 0, true);
}

int wino_conv1x1_bn(const float* A, const float* B, const float* bnBias, const float* bnScale,
                    float* C, long M, int Cin, int Kout, int relu, wino_stream_t s) {
  return wino_conv1x1_bn_ex(A, B, bnBias, bnScale, NULL, C, M, Cin, Kout, relu ? WINO_RELU : 0, s);
}

// 1x1 (C4 -> Cm) + BN + ReLU  ->  3x3 (Cm -> Cm) + BN + ReLU  ->  1x1 (Cm -> C4) + BN + skip + ReLU.
// Three launches on one stream; the two intermediates live in `workspace` in the padded
// [N][16][16][Cm] layout the 3x3 kernel reads and writes, so no repacking pass exists.
int wino_residual_block_hw(const float* x, const float* w1, const float* bn1Bias, const float* bn1Scale,
                           const float* U2, const float* bn2Bias, const float* bn2Scale,
                           const float* w3, const float* bn3Bias, const float* bn3Scale, float* out,
                           int N, int H, int W, int C4, int Cm, void* workspace, size_t workspace_bytes,
                           wino_stream_t s) {
  if (N < 1 || H < 1 || W < 1) { set_error("bad N=%d H=%d W=%d", N, H, W); return WINO_E_SHAPE; }
  if (!workspace || workspace_bytes < wino_residual_block_workspace_bytes_hw(N, H, W, Cm)) {
    set_error("workspace too small: need %zu bytes", wino_residual_block_workspace_bytes_hw(N, H, W, Cm));
    return WINO_E_ARG;
  }
  float* t1 = (float*)workspace;
  float* t2 = t1 + (size_t)N * (H + 2) * (W + 2) * Cm;
  int rc = wino_conv1x1_bn_ex_hw(x, w1, bn1Bias, bn1Scale, NULL, t1, N, H, W, C4, Cm, WINO_RELU | WINO_C_PADDED, s);
  if (rc) return rc;
  rc = wino_conv3x3_bn_relu_hw(t1, U2, bn2Bias, bn2Scale, t2, N, H, W, Cm, Cm, 1, s);
  if (rc) return rc;
  return wino_conv1x1_bn_ex_hw(t2, w3, bn3Bias, bn3Scale, x, out, N, H, W, Cm, C4,
                               WINO_RELU | WINO_A_PADDED | WINO_ADD_RESIDUAL, s);
}

int wino_residual_block_prepare_hw(int N, int H, int W, int C4, int Cm, wino_stream_t s) {
  if (N < 1 || H < 1 || W < 1) { set_error("bad N=%d H=%d W=%d", N, H, W); return WINO_E_SHAPE; }
  const long M = (long)N * H * W;
  if (int rc = wino_conv1x1_prepare(M, C4, Cm, s)) return rc;
  if (int rc = wino_conv3x3_prepare_hw(N, H, W, Cm, Cm, s)) return rc;
  return wino_conv1x1_prepare(M, Cm, C4, s);
}

int wino_residual_block_prepare(int N, int C4, int Cm, wino_stream_t s) {
  return wino_residual_block_prepare_hw(N, WINO_PQ, WINO_PQ, C4, Cm, s);
}

size_t wino_residual_block_workspace_bytes_hw(int N, int H, int W, int Cm) {
  return (size_t)2 * N * (H + 2) * (W + 2) * Cm * sizeof(float);
}

int wino_residual_block(const float* x, const float* w1, const float* bn1Bias, const float* bn1Scale,
                        const float* U2, const float* bn2Bias, const float* bn2Scale,
                        const float* w3, const float* bn3Bias, const float* bn3Scale, float* out,
                        int N, int C4, int Cm, void* workspace, size_t workspace_bytes,
                        wino_stream_t s) {
  return wino_residual_block_hw(x, w1, bn1Bias, bn1Scale, U2, bn2Bias, bn2Scale, w3, bn3Bias, bn3Scale, out,
                                N, WINO_PQ, WINO_PQ, C4, Cm, workspace, workspace_bytes, s);
}

size_t wino_residual_block_workspace_bytes(int N, int Cm) {
  return wino_residual_block_workspace_bytes_hw(N, WINO_PQ, WINO_PQ, Cm);
}

int wino_conv1x1_direct(const float* A, const float* B, const float* bnBias,
                        const float* bnScale, float* C, long M, int Cin, int Kout, int relu,
                        wino_stream_t s) {
  if (!A || !B || !bnBias || !bnScale || !C) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (M < 1 || Cin <= 0 || Kout <= 0) { set_error("bad 1x1 shape"); return WINO_E_SHAPE; }
  const long total = M * Kout;
  hipLaunchKernelGGL(conv1x1_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)s, A, B, bnBias, bnScale, C, M, Cin, Kout, relu);
  return launch_status("conv1x1_direct_kernel");
}

}  // extern "C"
