// Fused Winograd F(2x2,3x3) convolution + folded BN + ReLU for gfx950 (MI355X).
//
// Replaces the reference's three launches
//   kernel_{128,256}_winograd_BtdB -> kernel_*_OuterProduct_* -> kernel_*_winograd_AtIA
//   (Kernel128_winograd.cu:28-213, Kernel256_winograd.cu:27-218)
// with ONE kernel; the Winograd-domain tensors V (input) and M (products) never touch
// memory.  Design:
//
//   item       = TB=64 tiles x KB=64 out-channels x all 16 Winograd points; C is streamed in
//                chunks of BC=8 channels ("chunk iterations").  The launch is balanced by
//                iterations: whole-item rounds + a stream-K tail (wino_f2_fused_kernel.h).
//   workgroup  = 512 threads = 8 waves (2 per SIMD), all 160 KB of the CU's LDS.
//   wave (wt,ph) = 16 tiles x all 64 out-channels x 8 of the 16 points (rows 2 ph, 2 ph + 1 of the point grid)
//                = 32 accumulator tiles of v_mfma_f32_16x16x4_f32 (128 acc VGPRs); no two waves repeat a
//                  transform (round 1: 16 tiles x 32 out-channels x 16 points, B^T d B twice per tile block).
//   per chunk  : LDS-DMA (buffer_load_dwordx4 ... lds) stages
//                  raw[64 tiles][16 px][8 c]   (the 4x4 input patches, 32 KB, 2 stages)
//                  U  [16 pts][64 k][8 c]      (pre-packed filter chunk, 32 KB, 3 stages)
//                two iterations ahead of the MFMAs, continuously across items.
//   A operand  : each lane reads 3 of its tile's 4 patch rows for 2 channels (12 x ds_read_b64, addresses held
//                as absolute LDS pointers in registers) and applies its half of B^T d B in registers (16 packed
//                operations) -> V[8 pts]; no cross-lane traffic is needed because the MFMA A-fragment wants
//                exactly "one tile row, one channel" per lane.
//   B operand  : ds_read_b64 of the packed filter chunk.
//   loop       : two copies of the body, one per raw-stage parity (the stage is an immediate of the patch reads).
//   epilogue   : the wave's partial A^T m A (its two point rows) in-lane; the halves of a (tile, out-channel)
//                meet through LDS (wave pair w, w ^ 1), after which wave (wt, ph) owns 16 tiles x 32
//                out-channels: scale*y+bias, ReLU; the tiles go through the wave's own 8 KB of LDS and leave as
//                whole 128-byte runs of the padded NHWC output; the zero ring is written once per
//                launch by a flat ring pass.
// Small batches (the reference's N = 1) take wino_f2_small_kernel.h instead; feature maps other
// than 14x14 run the same kernel with the geometry in its arguments (GEN = true).
//
// LDS bank-conflict avoidance is done by XOR-permuting 16-byte units, applied on the DMA
// *source* address for the raw patches (the LDS destination of an LDS-DMA is lane-linear)
// and baked into the packed filter layout for U.
#include "wino_f2_small_kernel.h"

#include <atomic>
#include <mutex>
#include <vector>
#include <cstdlib>
#include <cstring>

namespace wino {
namespace {

using namespace fused;


// Position (in floats, 0..7) inside the 8-channel group of the packed filter at which
// channel `cl` (0..7) of out-channel `kl` (0..63 within the k-block) is stored: the
// 8-byte quarter index is XORed with 2*bit3(kl) so that the B-fragment ds_read_b64 of
// lanes (n, h) and (n+8, h) hit different bank groups.
__host__ __device__ constexpr int u_pos(int kl, int cl) {
  return ((((cl >> 1) ^ (((kl >> 3) & 1) << 1)) << 1) | (cl & 1));
}

__host__ __device__ inline size_t u_index(int C, int K, int e, int c, int k) {
  (void)C;
  const int it = c >> 3, cl = c & 7, kb = k >> 6, kl = k & 63;
  return ((((size_t)it * (K >> 6) + kb) * 16 + e) * 64 + kl) * 8 + u_pos(kl, cl);
}

// ---------------------------------------------------------------------------------
// Filter transforms (offline; reference: data_generator.py:63-78)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void f2_from_taps(const double g[3][3], double u[4][4]) {
  // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  u = G g G^T
  double t[4][3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    t[0][j] = g[0][j];
    t[1][j] = 0.5 * (g[0][j] + g[1][j] + g[2][j]);
    t[2][j] = 0.5 * (g[0][j] - g[1][j] + g[2][j]);
    t[3][j] = g[2][j];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u[i][0] = t[i][0];
    u[i][1] = 0.5 * (t[i][0] + t[i][1] + t[i][2]);
    u[i][2] = 0.5 * (t[i][0] - t[i][1] + t[i][2]);
    u[i][3] = t[i][2];
  }
}

__global__ void filter_transform_f2_kernel(const float* __restrict__ w, float* __restrict__ U,
                                           int C, int K) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * K) return;
  const int k = idx / C, c = idx - k * C;
  double g[3][3], u[4][4];
  const float* p = w + (size_t)idx * 9;  // [K][C][3][3]
#pragma unroll
  for (int i = 0; i < 9; i++) g[i / 3][i % 3] = (double)p[i];
  f2_from_taps(g, u);
#pragma unroll
  for (int e = 0; e < 16; e++) U[u_index(C, K, e, c, k)] = (float)u[e >> 2][e & 3];
}

// u36 [36][C][K] = G4 g G4^T  ->  g = L u36 L^T with L = [[4,0,0,0,0,0],[0,-3,3,0,0,0],[0,0,0,0,0,1]]
// (L G4 = I for the reference's G4, data_generator.py:65), then the F(2x2) transform.
__global__ void filter_import_f4_kernel(const float* __restrict__ u36, float* __restrict__ U,
                                        int C, int K) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // c*K + k  (k fastest: coalesced)
  if (idx >= C * K) return;
  const int c = idx / K, k = idx - c * K;
  double m[6][6];
#pragma unroll
  for (int e = 0; e < 36; e++) m[e / 6][e % 6] = (double)u36[(size_t)e * C * K + idx];
  double t[3][6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    t[0][j] = 4.0 * m[0][j];
    t[1][j] = 3.0 * (m[2][j] - m[1][j]);
    t[2][j] = m[5][j];
  }
  double g[3][3], u[4][4];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    g[i][0] = 4.0 * t[i][0];
    g[i][1] = 3.0 * (t[i][2] - t[i][1]);
    g[i][2] = t[i][5];
  }
  f2_from_taps(g, u);
#pragma unroll
  for (int e = 0; e < 16; e++) U[u_index(C, K, e, c, k)] = (float)u[e >> 2][e & 3];
}

}  // namespace
}  // namespace wino

using namespace wino;
namespace wino { int last_clock_1x1(unsigned long long* stamps); }   // conv1x1.hip

extern "C" {

size_t wino_filter_f2_elems(int C, int K) { return (size_t)16 * C * K; }

long wino_filter_f2_index(int C, int K, int e, int c, int k) {
  if (C <= 0 || K <= 0 || (C % 8) || (K % 64) || e < 0 || e >= 16 || c < 0 || c >= C || k < 0 || k >= K)
    return -1;
  return (long)u_index(C, K, e, c, k);
}

static int check_ck(int C, int K) {
  if (C <= 0 || K <= 0 || (C % 8) != 0 || (K % 64) != 0) {
    set_error("unsupported channels C=%d K=%d (need C %% 8 == 0, K %% 64 == 0)", C, K);
    return WINO_E_SHAPE;
  }
  return WINO_OK;
}

int wino_filter_transform_f2(const float* w_kcrs, float* U, int C, int K, wino_stream_t s) {
  if (!w_kcrs || !U) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (misaligned16(U)) { set_error("tensor pointers must be 16-byte aligned"); return WINO_E_ARG; }
  if (int rc = check_ck(C, K)) return rc;
  const int n = C * K;
  hipLaunchKernelGGL(filter_transform_f2_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)s, w_kcrs, U, C, K);
  return launch_status("filter_transform_f2_kernel");
}

int wino_filter_import_f4(const float* u36, float* U, int C, int K, wino_stream_t s) {
  if (!u36 || !U) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (misaligned16(U)) { set_error("tensor pointers must be 16-byte aligned"); return WINO_E_ARG; }
  if (int rc = check_ck(C, K)) return rc;
  const int n = C * K;
  hipLaunchKernelGGL(filter_import_f4_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)s, u36, U, C, K);
  return launch_status("filter_import_f4_kernel");
}

}  // extern "C"

// ---------------------------------------------------------------------------------
// Stream-K scratch of this kernel: 2 slabs of SLAB_BYTES per logical workgroup, one ticket counter
// per (item, wave); the per-(device, stream) set lives in wino_runtime.hip (sk_scratch).
// ---------------------------------------------------------------------------------
static int sk_cus(int dev, int* cus) { return device_cus(dev, cus); }

static int sk_workspace(int dev, hipStream_t s, int G, size_t items, SkBufs* bufs) {
  const size_t wgs = G < 256 ? 256 : (size_t)G;
  return sk_scratch(dev, s, 2 * wgs * SLAB_BYTES, items * 8, bufs);
}

// Launch geometry of the throughput kernel (see the header of wino_f2_fused_kernel.h): G logical
// workgroups run items / G whole-item rounds and share the remaining items % G items as a
// stream-K tail.  Two candidates are priced with a small cost model, in units of one chunk
// iteration (~2.2 us at 2.3 GHz):
//   * G = items when that fits the CUs: whole items only, no tail, no hand-off;
//   * G = CUs (capped so that a workgroup keeps >= SK_MIN_ITERS iterations);
//   * when all items fit the CUs at once: item-aligned ranges, G = T / s for the shortest divisor
//     s >= 4 of the chunk count whose grid fits -- every workgroup then holds one segment instead
//     of straddling two items (one epilogue, not two).
// Measured on MI355X (N = 128): 256 channels 392 items -> G = 256 (1 round + 17-iteration tail)
// 125 us vs 151 us for two whole-item rounds; 128 channels 196 items -> G = 196, 43 us vs 51 us
// for G = 256 (all tail).  An epilogue costs ~2.3 iterations, the tail's hand-off ~3.2.
constexpr int SK_MIN_ITERS = 8, SK_ALIGNED_MIN_ITERS = 4;
// (hand-off refitted after the loop changes of this round: whole items vs the all-tail grid at 128 / 192 /
//  256 / 384 channels, N = 46..100, solve to 2.7-3.6 iterations; it was 4.8)
constexpr double SK_EPILOGUE_ITERS = 2.3, SK_HANDOFF_ITERS = 3.2, SK_HANDOFF_ALLTAIL_ITERS = 5.0;
static double sk_cost(long long items, int nchunks, long long G) {
  const long long ndp = items / G, tail_items = items % G;
  double c = (double)ndp * (nchunks + SK_EPILOGUE_ITERS);
  if (tail_items) {
    const long long tail_T = tail_items * nchunks, per = (tail_T + G - 1) / G;
    // A range usually straddles two items (two segments, two epilogues).  When the range length
    // divides an item's chunk count and the ranges are all equal, every workgroup holds exactly one
    // segment: one epilogue.  Measured (N = 12..40, 128 channels): 20.6-23.1 us with 4-iteration
    // aligned ranges against 27.3-28.1 with 8-iteration unaligned ones.
    const bool aligned = tail_T % G == 0 && per < nchunks && nchunks % per == 0;
    const double segments = aligned ? 1.0 : (double)((per + nchunks - 1) / nchunks) + 1.0;
    // (an all-tail grid whose ranges straddle items gathers everything at the very end of the launch, on every
    //  workgroup's critical path: re-measured after the ticket rule of round 2 -- 256 channels N = 24 / 32 / 48 / 64
    //  and 128 channels N = 96 solve to 4.1-6.2 iterations, against 2.7-3.2 for aligned ranges and 1.5-3 behind
    //  whole-item rounds; with 5.0 the policy takes the aligned halves at 256 channels N = 32 (45.1 against 48.0 us)
    //  and whole items at N = 64 (72.7 against 75.2) and at 128 channels N = 96 (39.8 against 40.8))
    const double handoff = (!aligned && ndp == 0) ? SK_HANDOFF_ALLTAIL_ITERS : SK_HANDOFF_ITERS;
    c += (double)per + segments * SK_EPILOGUE_ITERS + handoff;
    // (Not modelled: ranges of one or two iterations cut an item into many segments, and whoever gathers it reads
    //  their slabs one after the other.  192 channels, N = 112 -- 258 items on 256 CUs, a 48-iteration tail, 24
    //  segments per item -- takes 119.6 us against 107.1 for two even rounds on 129 workgroups; a per-segment term
    //  that catches this case (0.9-1.0 iterations per segment) mis-prices 128 channels N = 176 and 384 channels
    //  N = 112, where the same kind of tail costs 0.4 per segment and the even rounds lose 16-17 %.  Left as is.)
  }
  return c;
}
static int sk_grid_for(int cus, long long items, int nchunks, int* G) {
  const Knobs kn = knobs();   // developer overrides (cached; tests sweep the decomposition through them)
  const int min_iters = kn.sk_min_iters > 0 ? kn.sk_min_iters : SK_MIN_ITERS;
  const long long T = items * nchunks;
  long long g = cus;
  if (g > T / min_iters) g = T / min_iters;
  if (g < 1) g = 1;
  if (items <= cus && sk_cost(items, nchunks, items) <= sk_cost(items, nchunks, g)) g = items;
  if (items < cus) {
    // item-aligned ranges: the shortest divisor of the chunk count (at least SK_ALIGNED_MIN_ITERS:
    // below that the serial gather of an item's segments outweighs the shorter ranges -- 256 channels
    // N = 5: 2-iteration ranges 29.7 us, 4-iteration ranges 24.5) whose grid fits the CUs
    for (int sl = SK_ALIGNED_MIN_ITERS; sl < nchunks; sl++) {
      if (nchunks % sl) continue;
      const long long ga = T / sl;
      if (ga > cus) continue;
      if (sk_cost(items, nchunks, ga) < sk_cost(items, nchunks, g)) g = ga;
      break;
    }
  }
  if (items > cus) {
    // several rounds: a grid a little below the CU count can make the rounds come out even -- items = r * G
    // exactly: whole items only, no tail, no hand-off (256 channels, N = 160: 492 items = 2 x 246: 138.6 us
    // against 144.2 for G = 256 with a 236-item tail).  Priced with the same model, exact divisions only.
    for (long long gg = cus - 1; gg >= cus - cus / 4 && gg >= 1; gg--)
      if (items % gg == 0 && sk_cost(items, nchunks, gg) < sk_cost(items, nchunks, g)) g = gg;
  }
  if (kn.sk_grid >= 1) g = kn.sk_grid;
  if (g > 16384) g = 16384;   // 2 * G slabs of 64 KB must stay below the 4 GiB a buffer descriptor spans
  *G = (int)g;
  return WINO_OK;
}

// k-groups of the stream-K tail (wino_f2_fused_kernel.h): K/64 when the grid is a multiple of it -- the items are
// K/64 per tile block, so the tail's item count is then a multiple too -- else 1 (one item-major list, round 2's
// scheme).  WINO_SK_KP=0 forces round 2's scheme altogether (A/B measurements).
static int tail_groups(int G, int K) {
  const int kblk = K / KB;
  if (knobs().sk_kp == 0 || knobs().sk_kp == 3 || kblk <= 1 || (G % kblk) != 0) return 1;
  return kblk;
}

// Phase order of a k-group's tail ranges (tail_range_of, wino_f2_fused_kernel.h): for Gp equal ranges of q iterations
// the period P = nchunks / gcd(q, nchunks) of their channel phases, the inverse of q / gcd modulo P, and Gp / P.
struct PhaseOrder { int P, inv, copies; };
static PhaseOrder phase_order(long long q, long long rem, int nchunks, long long Gp) {
  PhaseOrder id = {1, 0, (int)Gp};
  if (rem != 0 || q <= 0 || knobs().sk_kp == 0 || knobs().sk_kp == 2) return id;   // (WINO_SK_KP=0: round 2's scheme; 2: k-groups without the phase order; 3: phase order without the groups -- A/B)
  long long a = q % nchunks, b = nchunks;
  while (a) { const long long t = b % a; b = a; a = t; }
  const long long g = b, P = nchunks / g;
  if (P <= 1 || Gp % P != 0) return id;
  const long long qq = (q / g) % P;
  long long inv = 0;
  for (long long x = 1; x < P; x++)
    if ((qq * x) % P == 1) { inv = x; break; }
  if (!inv) return id;
  return PhaseOrder{(int)P, (int)inv, (int)(Gp / P)};
}

// The largest batch one launch takes: the kernels address the tensors with 32-bit byte offsets
// (both tensors must stay below 4 GiB) and the stream-K bookkeeping counts chunk iterations in
// 32 bits.  Larger batches are split by the launcher (images are independent).
static long long conv3x3_batch_limit(int H, int W, int C, int K) {
  const unsigned long long per_image = (unsigned long long)(H + 2) * (W + 2) * (unsigned long long)(C > K ? C : K) * sizeof(float);
  long long n = (long long)(((1ull << 32) - 1) / per_image);
  const long long tiles = (long long)((H + 1) / 2) * ((W + 1) / 2);
  const long long per_tb = (long long)(K / KB) * (C / BC);               // chunk iterations per 64-tile block
  const long long max_tb = ((1ll << 31) - 1) / per_tb - 1;
  const long long n_iter = max_tb * TB / tiles;
  if (n > n_iter) n = n_iter;
  if (n > (1ll << 30)) n = 1ll << 30;
  return n;
}

static int check_conv3x3_dims(int H, int W, int C, int K) {
  if (int rc = check_ck(C, K)) return rc;
  if (H < 1 || W < 1 || H > 4094 || W > 4094) {
    set_error("unsupported feature map %dx%d", H, W);
    return WINO_E_SHAPE;
  }
  if (conv3x3_batch_limit(H, W, C, K) < 1) {
    set_error("%dx%d C=%d K=%d: one image does not fit a launch (tensors must stay below 4 GiB)", H, W, C, K);
    return WINO_E_SHAPE;
  }
  return WINO_OK;
}

// one launch
static int check_conv3x3(int N, int H, int W, int C, int K) {
  if (int rc = check_conv3x3_dims(H, W, C, K)) return rc;
  if (N < 1 || N > conv3x3_batch_limit(H, W, C, K)) {
    set_error("bad batch N=%d (one launch takes 1..%lld images of this shape: input/output below 4 GiB)", N,
              conv3x3_batch_limit(H, W, C, K));
    return WINO_E_SHAPE;
  }
  return WINO_OK;
}

// Two kernels, same arithmetic: the throughput kernel (64-tile x 64-out-channel items, 8-wave
// workgroups, whole-item rounds + stream-K tail) and the one-wave-per-SIMD latency kernel (blocks of 16 tiles
// x 16 CT out-channels, CT = 1, 2 or 4; any feature map; wino_f2_small_kernel.h), which wins while its blocks fit
// ONE round of the CUs (a second round of blocks doubles the latency kernel's time at once).
// While the blocks leave CUs idle the latency kernel also splits a block's contraction over S workgroups
// (C-split): S as large as the idle CUs allow (at most 8, and every wave of the S workgroups gets a task in the
// first round: 4 S <= 2 C / 16).
// Among the block widths that fit, the one with the shortest modelled time (small_form below; least squares over the
// forms measured by tools/latency_cases.py explore3, profiles/r3/), and only while that beats the throughput kernel's
// fitted time in this regime, 18.8 us + 0.0174 us x C + 1.94 us x (chunk iterations per CU) (54 points at 64 ... 512
// channels, within 3 us): the latency kernel's price per task grows with C, and from 384 channels on a full round of
// its blocks is the slower launch.
// WINO_3X3_ALGO=big|small, WINO_SMALL_SPLIT, WINO_SMALL_CT override.
struct SmallPlan {
  bool use;
  int split, nT16;       // nT16: blocks of 16 tiles
  size_t blocks;
  int ct;                // MFMA tiles per wave, side by side
  double t_us;           // the model's time
};
// T = 5.70 us + rounds x (0.487 + (0.509 + 0.229 fill) CT) us + split cost: a round is 8 whole-line pixel loads per wave
// (the workgroup's 32, shared through LDS) and 8 CT filter-fragment loads; fill = workgroups / CUs; the split cost (slab
// round trip, growing with the block) 0.4 / 1.2 / 4.5 us at CT = 1 / 2 / 4.  148 measured forms, rms 0.85 us.
constexpr double SMALL_T0 = 5.70, SMALL_ROUND = 0.487, SMALL_FLT = 0.509, SMALL_FLT_FILL = 0.2285;
// The block width ct on `cus` CUs: its blocks and the C-split.  false if its blocks do not fit one round.
static bool small_form(int N, int tiles, int C, int K, int cus, int ct, SmallPlan* pl) {
  if ((K % (16 * ct)) != 0) return false;
  pl->ct = ct;
  pl->nT16 = (int)(((long long)N * tiles + 15) / 16);
  const long long blocks = (long long)pl->nT16 * (K / (16 * ct));
  pl->blocks = (size_t)blocks;
  pl->split = 1;
  pl->t_us = 0;
  if (blocks > cus) return false;
  const int ntask = (C / 16) * 2;
  int sp = (int)(cus / blocks);
  if (sp > SMALL_MAX_SPLIT) sp = SMALL_MAX_SPLIT;
  while (sp > 1 && SMALL_WAVES * sp > ntask) sp--;
  pl->split = sp;
  const int waves = SMALL_WAVES * sp;
  const double fill = (double)(blocks * sp) / cus;
  const double split_us = sp > 1 ? (ct == 1 ? 0.38 : ct == 2 ? 1.21 : 4.45) : 0.0;
  pl->t_us = SMALL_T0 + (double)((ntask + waves - 1) / waves) * (SMALL_ROUND + (SMALL_FLT + SMALL_FLT_FILL * fill) * ct) + split_us;
  return true;
}
static SmallPlan small_plan(int N, int H, int W, int C, int K, int cus) {
  SmallPlan pl = {false, 1, 0, 0, 1, 0.0};
  if ((C % 16) != 0 || H < 1 || W < 1) return pl;
  const long long tiles_ll = (long long)((H + 1) / 2) * ((W + 1) / 2);
  if (tiles_ll * N > (1ll << 30)) return pl;
  const int tiles = (int)tiles_ll;
  const Knobs kn = knobs();
  if (kn.small3_ct == 1 || kn.small3_ct == 2 || kn.small3_ct == 4) {
    pl.use = small_form(N, tiles, C, K, cus, kn.small3_ct, &pl);
  } else {
    SmallPlan best = pl;
    for (int ct = 1; ct <= 4; ct *= 2) {
      SmallPlan f = pl;
      if (small_form(N, tiles, C, K, cus, ct, &f) && (!best.use || f.t_us < best.t_us)) { best = f; best.use = true; }
    }
    const double items = (double)(((long long)N * tiles + TB - 1) / TB) * (K / KB);
    const double t_big = 18.8 + 0.0174 * C + 1.94 * items * (C / BC) / cus;
    if (best.use && 1.08 * best.t_us < t_big) pl = best;   // (the margin: the model is 2-3 us short on a full round of wide blocks)
    else small_form(N, tiles, C, K, cus, 1, &pl), pl.use = false;   // (the counts a forced launch would use)
  }
  if (kn.algo_3x3 == 1) pl.use = false;
  if (kn.algo_3x3 == 2) pl.use = pl.nT16 <= 65535 && (K % (16 * pl.ct)) == 0;
  if (!pl.use) return pl;
  if (kn.small_split >= 1 && kn.small_split <= SMALL_MAX_SPLIT) pl.split = kn.small_split;
  return pl;
}
static int small_scratch(int dev, hipStream_t s, const SmallPlan& pl, SkBufs* bufs) {
  bufs->slabs = nullptr; bufs->tickets = nullptr; bufs->err = nullptr;
  if (pl.split <= 1) return WINO_OK;
  return sk_scratch(dev, s, pl.blocks * pl.split * pl.ct * SMALL_SLAB_BYTES, pl.blocks, bufs);
}

static int conv3x3_prepare(int N, int H, int W, int C, int K, hipStream_t s) {
  if (int rc = check_conv3x3_dims(H, W, C, K)) return rc;
  if (N < 1) { set_error("bad batch N=%d", N); return WINO_E_SHAPE; }
  {   // a batch the launcher splits: its largest launch decides the scratch
    long long step = conv3x3_batch_limit(H, W, C, K);
    if (N > step) N = (int)(step > 64 ? step - step % 64 : step);
  }
  if (int rc = check_conv3x3(N, H, W, C, K)) return rc;
  int dev = 0, cus = 0;
  WINO_HIP(hipGetDevice(&dev));
  if (int rc = sk_cus(dev, &cus)) return rc;
  SkBufs bufs;
  const SmallPlan sp = small_plan(N, H, W, C, K, cus);
  if (sp.use) return small_scratch(dev, s, sp, &bufs);
  const int nTB = (int)(((long long)N * ((H + 1) / 2) * ((W + 1) / 2) + TB - 1) / TB);
  const size_t items = (size_t)nTB * (K / KB);
  int G = 0;
  if (int rc = sk_grid_for(cus, (long long)items, C / BC, &G)) return rc;
  return sk_workspace(dev, s, G, items, &bufs);
}

template <bool GEN, bool TAIL>
static int launch_fused(const FusedParams& prm, int G, int dev, hipStream_t s) {
  // raise the dynamic-LDS cap (all 160 KB of the CU) once per device
  static std::atomic<unsigned long long> attr_done{0};
  if (!((attr_done.load() >> (dev & 63)) & 1ull)) {
    WINO_HIP(hipFuncSetAttribute((const void*)(wino_f2_fused_kernel<0, GEN, TAIL>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_done.fetch_or(1ull << (dev & 63));
  }
  hipLaunchKernelGGL((wino_f2_fused_kernel<0, GEN, TAIL>), dim3(G), dim3(NTHREADS), LDS_BYTES, s, prm);
  const int rc = launch_status("wino_f2_fused_kernel");
  if (rc) sk_mark_failed(dev, s);   // the launch held the stream's scratch
  return rc;
}

static int conv3x3_launch_one(const float* in, const float* U, const float* bnBias, const float* bnScale,
                              float* out, int N, int H, int W, int C, int K, int relu, hipStream_t s);

// Any batch: batches whose tensors would reach 4 GiB go out as several launches of whole images
// (a multiple of 64 images each, so that every launch but the last fills its 64-tile blocks).
static int conv3x3_launch(const float* in, const float* U, const float* bnBias, const float* bnScale,
                          float* out, int N, int H, int W, int C, int K, int relu, hipStream_t s) {
  if (!in || !U || !bnBias || !bnScale || !out) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (misaligned16(in, U, out)) { set_error("tensor pointers must be 16-byte aligned"); return WINO_E_ARG; }
  if (int rc = check_conv3x3_dims(H, W, C, K)) return rc;
  if (N < 1) { set_error("bad batch N=%d", N); return WINO_E_SHAPE; }
  long long step = conv3x3_batch_limit(H, W, C, K);
  if (N <= step) return conv3x3_launch_one(in, U, bnBias, bnScale, out, N, H, W, C, K, relu, s);
  if (step > 64) step -= step % 64;
  const size_t in_img = (size_t)(H + 2) * (W + 2) * C, out_img = (size_t)(H + 2) * (W + 2) * K;
  for (long long n0 = 0; n0 < N; n0 += step) {
    const int n = (int)(N - n0 < step ? N - n0 : step);
    if (int rc = conv3x3_launch_one(in + (size_t)n0 * in_img, U, bnBias, bnScale, out + (size_t)n0 * out_img,
                                    n, H, W, C, K, relu, s)) return rc;
  }
  return WINO_OK;
}

static int conv3x3_launch_one(const float* in, const float* U, const float* bnBias, const float* bnScale,
                              float* out, int N, int H, int W, int C, int K, int relu, hipStream_t s) {
  if (int rc = check_conv3x3(N, H, W, C, K)) return rc;
  const bool fixed14 = H == WINO_PQ && W == WINO_PQ;
  int dev = 0, cus = 0;   // one device query per launch; the CU count is cached per device
  WINO_HIP(hipGetDevice(&dev));
  if (int rc = sk_cus(dev, &cus)) return rc;
  const SmallPlan sp = small_plan(N, H, W, C, K, cus);
  if (sp.use) {
    SkBufs bufs;
    if (int rc = small_scratch(dev, s, sp, &bufs)) return rc;
    const unsigned stx = (unsigned)((W + 1) / 2), st = (unsigned)((H + 1) / 2) * stx;
    const Geo sgeo = {H + 2, W + 2, st, stx, make_fastdiv(st), make_fastdiv(stx)};
    const SmallParams prm = {in, U, bnBias, bnScale, out, N, C, K, relu, bufs.slabs, bufs.tickets, bufs.err, nullptr, sgeo};
    const dim3 grid(K / (16 * sp.ct), sp.nT16, sp.split), block(64 * SMALL_WAVES);   // x = out-channel block: see the kernel
    if (fixed14) {
      if (sp.ct == 4) hipLaunchKernelGGL((wino_f2_small_kernel<4, false>), grid, block, 0, s, prm);
      else if (sp.ct == 2) hipLaunchKernelGGL((wino_f2_small_kernel<2, false>), grid, block, 0, s, prm);
      else hipLaunchKernelGGL((wino_f2_small_kernel<1, false>), grid, block, 0, s, prm);
    } else {
      if (sp.ct == 4) hipLaunchKernelGGL((wino_f2_small_kernel<4, true>), grid, block, 0, s, prm);
      else if (sp.ct == 2) hipLaunchKernelGGL((wino_f2_small_kernel<2, true>), grid, block, 0, s, prm);
      else hipLaunchKernelGGL((wino_f2_small_kernel<1, true>), grid, block, 0, s, prm);
    }
    const int rc = launch_status("wino_f2_small_kernel");
    if (rc && sp.split > 1) sk_mark_failed(dev, s);
    return rc;
  }
  const unsigned tiles_x = (unsigned)((W + 1) / 2), tiles = (unsigned)((H + 1) / 2) * tiles_x;
  const int nTB = (int)(((long long)N * tiles + TB - 1) / TB);
  const size_t items = (size_t)nTB * (K / KB);
  int G = 0;
  if (int rc = sk_grid_for(cus, (long long)items, C / BC, &G)) return rc;
  SkBufs bufs;
  if (int rc = sk_workspace(dev, s, G, items, &bufs)) return rc;
  const long long Tt = (long long)(items % (size_t)G) * (C / BC);   // the stream-K tail's iterations
  const int kp = tail_groups(G, K);                                   // ... cut per k-block when the grid allows it
  const long long Tg = Tt / kp, Gp = G / kp;
  const Geo geo = {H + 2, W + 2, tiles, tiles_x, make_fastdiv(tiles), make_fastdiv(tiles_x)};
  const PhaseOrder po = phase_order(Tg / Gp, Tg % Gp, C / BC, Gp);
  const FusedParams prm = {in, U, N, C, K, relu, nTB, (int)(items / (size_t)G), (unsigned)(Tg / Gp), (unsigned)(Tg % Gp), kp, (int)Gp,
                           po.P, po.inv, po.copies, make_fastdiv((unsigned)kp), make_fastdiv((unsigned)po.P), make_fastdiv((unsigned)po.copies), geo, bnBias, bnScale, out, bufs.slabs, bufs.tickets, bufs.err, nullptr};
  // whole items only (no stream-K tail): the kernel variant without the hand-off in its epilogue
  if (Tt == 0) return fixed14 ? launch_fused<false, false>(prm, G, dev, s) : launch_fused<true, false>(prm, G, dev, s);
  return fixed14 ? launch_fused<false, true>(prm, G, dev, s) : launch_fused<true, true>(prm, G, dev, s);
}

// Diagnostic: the throughput kernel's stamped build (ABLATE = 16: s_memtime / s_memrealtime at the start
// and at the end of every workgroup's main loop, each pair stored at once; the outputs are the product kernel's).  bench.py runs it right
// after its timed region to report the clock the chip holds inside THIS kernel under sustained load.
static int conv3x3_clock_probe(const float* in, const float* U, const float* bnBias, const float* bnScale,
                               float* out, int N, int C, int K, unsigned long long* stamps, int* workgroups,
                               hipStream_t s) {
  if (!in || !U || !bnBias || !bnScale || !out || !stamps || !workgroups) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (int rc = check_conv3x3(N, WINO_PQ, WINO_PQ, C, K)) return rc;
  int dev = 0, cus = 0;
  WINO_HIP(hipGetDevice(&dev));
  if (int rc = sk_cus(dev, &cus)) return rc;
  const int nTB = (int)(((long long)N * WINO_TILES + TB - 1) / TB);
  const size_t items = (size_t)nTB * (K / KB);
  int G = 0;
  if (int rc = sk_grid_for(cus, (long long)items, C / BC, &G)) return rc;
  if (G > 2048) { set_error("clock probe: grid %d exceeds the stamp buffer", G); return WINO_E_SHAPE; }
  SkBufs bufs;
  if (int rc = sk_workspace(dev, s, G, items, &bufs)) return rc;
  const long long Tt = (long long)(items % (size_t)G) * (C / BC);
  const int kp = tail_groups(G, K);
  const long long Tg = Tt / kp, Gp = G / kp;
  const Geo geo = {WINO_HW, WINO_HW, WINO_TILES, 7, make_fastdiv(WINO_TILES), make_fastdiv(7)};
  const PhaseOrder po = phase_order(Tg / Gp, Tg % Gp, C / BC, Gp);
  const FusedParams prm = {in, U, N, C, K, 1, nTB, (int)(items / (size_t)G), (unsigned)(Tg / Gp), (unsigned)(Tg % Gp), kp, (int)Gp,
                           po.P, po.inv, po.copies, make_fastdiv((unsigned)kp), make_fastdiv((unsigned)po.P), make_fastdiv((unsigned)po.copies), geo, bnBias, bnScale, out, bufs.slabs, bufs.tickets, bufs.err, stamps};
  static std::atomic<unsigned long long> attr_done{0};
  if (!((attr_done.load() >> (dev & 63)) & 1ull)) {
    WINO_HIP(hipFuncSetAttribute((const void*)(wino_f2_fused_kernel<16, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_done.fetch_or(1ull << (dev & 63));
  }
  hipLaunchKernelGGL((wino_f2_fused_kernel<16, false>), dim3(G), dim3(NTHREADS), LDS_BYTES, s, prm);
  *workgroups = G;
  return launch_status("wino_f2_fused_kernel (stamped)");
}

extern "C" {

int wino_diag_conv3x3_clock(const float* in, const float* U, const float* bnBias, const float* bnScale,
                            float* out, int N, int C, int K, unsigned long long* stamps_dev,
                            int* workgroups, wino_stream_t s) {
  return conv3x3_clock_probe(in, U, bnBias, bnScale, out, N, C, K, stamps_dev, workgroups, (hipStream_t)s);
}

int wino_conv3x3_plan(int N, int H, int W, int C, int K, int cus, int* grid, int* rounds, long* tail_iters,
                      int* iters_per_item) {
  if (!grid || !rounds || !tail_iters || !iters_per_item || cus < 1) { set_error("bad argument"); return WINO_E_ARG; }
  if (int rc = check_conv3x3(N, H, W, C, K)) return rc;
  const long long nTB = ((long long)N * ((H + 1) / 2) * ((W + 1) / 2) + TB - 1) / TB;
  const long long items = nTB * (K / KB);
  int G = 0;
  if (int rc = sk_grid_for(cus, items, C / BC, &G)) return rc;
  *grid = G;
  *rounds = (int)(items / G);
  *tail_iters = (long)((items % G) * (C / BC));
  *iters_per_item = C / BC;
  return WINO_OK;
}

// Host-side only: does this shape take the latency kernel on a device with `cus` CUs, and in which form.
int wino_conv3x3_small_plan(int N, int H, int W, int C, int K, int cus, int* use, int* point_rows, int* split,
                            int* workgroups) {
  if (!use || !point_rows || !split || !workgroups || cus < 1) { set_error("bad argument"); return WINO_E_ARG; }
  if (int rc = check_conv3x3(N, H, W, C, K)) return rc;
  const SmallPlan pl = small_plan(N, H, W, C, K, cus);
  *use = pl.use;
  *point_rows = 2;
  *split = pl.split;
  *workgroups = pl.use ? (int)(pl.blocks * pl.split) : 0;
  return WINO_OK;
}

int wino_conv3x3_small_plan2(int N, int H, int W, int C, int K, int cus, int* use, int* point_rows, int* split,
                             int* col_tiles, int* workgroups) {
  if (!col_tiles) { set_error("bad argument"); return WINO_E_ARG; }
  if (int rc = wino_conv3x3_small_plan(N, H, W, C, K, cus, use, point_rows, split, workgroups)) return rc;
  *col_tiles = small_plan(N, H, W, C, K, cus).ct;
  return WINO_OK;
}

int wino_diag_last_clock(int kernel, wino_stream_t s, unsigned long long stamps[4]) {
  if (!stamps || (kernel != 0 && kernel != 1)) { set_error("bad argument"); return WINO_E_ARG; }
  WINO_HIP(hipStreamSynchronize((hipStream_t)s));
  if (kernel == 1) return wino::last_clock_1x1(stamps);
  WINO_HIP(hipMemcpyFromSymbol(stamps, HIP_SYMBOL(wino::fused::wino_clk_slot_3x3), 4 * sizeof(unsigned long long)));
  return WINO_OK;
}

// Host-side only: into how many k-groups the launch wino_conv3x3_plan describes cuts its tail (1 = one item-major
// list).  Group k owns the tail items of out-channel block k, item rounds*grid + groups*j + k for j = 0, 1, ...;
// its tail_iters / groups iterations are cut into Gp = grid / groups equal ranges [r*T/Gp, (r+1)*T/Gp), and the
// workgroup l = groups*j + k at position j runs range r = (phase_inv * (j / phase_copies)) % phase_period +
// phase_period * (j % phase_copies) -- the ranges of one XCD's workgroups start at consecutive channel phases.
int wino_conv3x3_plan_groups(int N, int H, int W, int C, int K, int cus, int* groups, int* phase_period, int* phase_inv,
                             int* phase_copies) {
  if (!groups || !phase_period || !phase_inv || !phase_copies || cus < 1) { set_error("bad argument"); return WINO_E_ARG; }
  if (int rc = check_conv3x3(N, H, W, C, K)) return rc;
  const long long nTB = ((long long)N * ((H + 1) / 2) * ((W + 1) / 2) + TB - 1) / TB;
  const long long items = nTB * (K / KB);
  int G = 0;
  if (int rc = sk_grid_for(cus, items, C / BC, &G)) return rc;
  const int kp = tail_groups(G, K);
  const long long Tg = (items % G) * (C / BC) / kp, Gp = G / kp;
  const PhaseOrder po = phase_order(Tg / Gp, Tg % Gp, C / BC, Gp);
  *groups = kp;
  *phase_period = po.P;
  *phase_inv = po.inv;
  *phase_copies = po.copies;
  return WINO_OK;
}

int wino_conv3x3_prepare(int N, int C, int K, wino_stream_t s) {
  return conv3x3_prepare(N, WINO_PQ, WINO_PQ, C, K, (hipStream_t)s);
}

int wino_conv3x3_prepare_hw(int N, int H, int W, int C, int K, wino_stream_t s) {
  return conv3x3_prepare(N, H, W, C, K, (hipStream_t)s);
}

int wino_conv3x3_bn_relu(const float* in, const float* U, const float* bnBias,
                         const float* bnScale, float* out, int N, int C, int K, int relu,
                         wino_stream_t s) {
  return conv3x3_launch(in, U, bnBias, bnScale, out, N, WINO_PQ, WINO_PQ, C, K, relu, (hipStream_t)s);
}

int wino_conv3x3_bn_relu_hw(const float* in, const float* U, const float* bnBias,
                            const float* bnScale, float* out, int N, int H, int W, int C, int K,
                            int relu, wino_stream_t s) {
  return conv3x3_launch(in, U, bnBias, bnScale, out, N, H, W, C, K, relu, (hipStream_t)s);
}

}  // extern "C"
