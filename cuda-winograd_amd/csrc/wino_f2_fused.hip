// Fused Winograd F(2x2,3x3) convolution + folded BN + ReLU for gfx950 (MI355X).
//
// Replaces the reference's three launches
//   kernel_{128,256}_winograd_BtdB -> kernel_*_OuterProduct_* -> kernel_*_winograd_AtIA
//   (Kernel128_winograd.cu:28-213, Kernel256_winograd.cu:27-218)
// with ONE kernel; the Winograd-domain tensors V (input) and M (products) never touch
// memory.  Design:
//
//   workgroup  = 512 threads = 8 waves (2 per SIMD), owns TB=64 tiles x KB=64 out-channels
//                x all 16 Winograd points; C is streamed in chunks of BC=8 channels.
//   wave (wt,wk) = 16 tiles x 32 out-channels x 16 points
//                = 32 accumulator tiles of v_mfma_f32_16x16x4_f32 (128 acc VGPRs).
//   per chunk  : LDS-DMA (global_load_lds_dwordx4) stages
//                  raw[64 tiles][16 px][8 c]   (the 4x4 input patches, 32 KB)
//                  U  [16 pts][64 k][8 c]      (pre-packed filter chunk, 32 KB)
//                into one of two 64 KB LDS stages while the other one is consumed.
//   A operand  : each lane reads its tile's 4x4 patch for 2 channels (16 x ds_read_b64),
//                applies B^T d B in registers (32 VALU adds per channel) -> V[16 pts];
//                no cross-lane traffic is needed because the MFMA A-fragment wants
//                exactly "one tile row, one channel" per lane.
//   B operand  : ds_read_b64 of the packed filter chunk.
//   epilogue   : the 16 accumulators of one (tile, k) sit in the SAME lane/register slot
//                of 16 different MFMA tiles, so A^T m A is 24 in-lane adds; then
//                scale*y+bias, ReLU, and 64-byte-segment stores into the padded NHWC
//                output, plus the zero ring.
//
// LDS bank-conflict avoidance is done by XOR-permuting 16-byte units, applied on the DMA
// *source* address for the raw patches (the LDS destination of an LDS-DMA is lane-linear)
// and baked into the packed filter layout for U.
#include "wino_common.h"

#include <atomic>

namespace wino {
namespace {

constexpr int TB = 64;                       // tiles per workgroup
constexpr int KB = 64;                       // out-channels per workgroup
constexpr int BC = 8;                        // in-channels per pipeline stage
constexpr int NTHREADS = 512;
constexpr int RAW_BYTES = TB * 16 * BC * 4;  // 32768
constexpr int U_BYTES = 16 * KB * BC * 4;    // 32768
constexpr int STAGE_BYTES = RAW_BYTES + U_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // 131072
constexpr int U_CHUNK_FLOATS = 16 * KB * BC; // 8192 floats per (c-chunk, k-block)

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

// ---------------------------------------------------------------------------------
// The fused kernel
// ---------------------------------------------------------------------------------
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

__global__ void __launch_bounds__(NTHREADS, 2)
wino_f2_fused_kernel(const float* __restrict__ in, const float* __restrict__ Uq,
                     const float* __restrict__ bnBias, const float* __restrict__ bnScale,
                     float* __restrict__ out, int N, int C, int K, int relu, int nTB) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware block -> (tile block, k block): blocks b and b+8 share an XCD (its L2), so the
  // K/64 k-blocks that read the same input tiles are placed on the same XCD back to back.
  const int KBLK = K >> 6;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int kb = slot % KBLK;
  const int tb = (slot / KBLK) * 8 + xcd;
  if (tb >= nTB) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = w >> 1;  // which 16-tile block of the 64
  const int wk = w & 1;   // which 32-channel half of the 64
  const int totalTiles = N * WINO_TILES;

  // ---- DMA source offsets (loop invariant) ------------------------------------
  // raw stage layout: [tile 0..63][unit' 0..31] of 16 B; unit' = px'*2 + half'.
  // LDS unit (t, px', half') holds pixel px = px' ^ (t&7), channel half = half' ^ bit3(t).
  // wave-instruction q = 8*j + w (j = 0..3) covers tiles 2q, 2q+1.
  const float* raw_src[4];
  {
    const int up = lane & 31;
    const int pxp = up >> 1, halfp = up & 1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int tl = 16 * j + 2 * w + (lane >> 5);
      const int px = pxp ^ (tl & 7);
      const int half = halfp ^ ((tl >> 3) & 1);
      int g = tb * TB + tl;
      g = g < totalTiles ? g : totalTiles - 1;  // clamp: padded rows read a valid tile
      const TileCoord tc = decode_tile(g);
      const int y = 2 * tc.ty + (px >> 2), x = 2 * tc.tx + (px & 3);
      raw_src[j] = in + ((size_t)(tc.n * WINO_HW + y) * WINO_HW + x) * C + half * 4;
    }
  }
  const float* u_src = Uq + (size_t)kb * U_CHUNK_FLOATS + w * 256 + lane * 4;
  const size_t u_chunk_stride = (size_t)KBLK * U_CHUNK_FLOATS;

  auto issue = [&](int stage, int chunk) {
    char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; j++) dma16(raw_src[j] + chunk * BC, sbase + (8 * j + w) * 1024);
    const float* us = u_src + (size_t)chunk * u_chunk_stride;
#pragma unroll
    for (int j = 0; j < 4; j++) dma16(us + j * 2048, sbase + RAW_BYTES + (8 * j + w) * 1024);
  };

  // ---- fragment read addresses (loop invariant) ---------------------------------
  const int t16 = lane & 15, h = lane >> 4;
  // A: tile row tl = wt*16 + t16; 8-byte quarter h holds channels 2h, 2h+1 of the chunk
  const int a_base = (wt * 16 + t16) * 512 + ((h ^ (((t16 >> 3) & 1) << 1)) << 3);
  const int a_sw = t16 & 7;
  // B: k_local = wk*32 + cb*16 + t16
  int b_base[2];
#pragma unroll
  for (int cb = 0; cb < 2; cb++) {
    const int kl = wk * 32 + cb * 16 + t16;
    b_base[cb] = RAW_BYTES + kl * 32 + ((h ^ (((kl >> 3) & 1) << 1)) << 3);
  }

  f32x4 acc[16][2];
#pragma unroll
  for (int e = 0; e < 16; e++) {
    acc[e][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[e][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int nchunks = C / BC;
  issue(0, 0);
  for (int it = 0; it < nchunks; ++it) {
    wait_vmem_all();   // my own DMA pieces of chunk `it` have landed
    __syncthreads();   // everyone's have; everyone is also done reading the other stage
    if (it + 1 < nchunks) issue((it + 1) & 1, it + 1);

    const char* st = smem + (it & 1) * STAGE_BYTES;

    // ---- A operand: load the 4x4 patch (2 channels per lane) and transform ------
    f32x2 d[16];
#pragma unroll
    for (int px = 0; px < 16; px++)
      d[px] = *(const f32x2*)(st + a_base + (((px & 7) ^ a_sw) << 5) + ((px >> 3) << 8));
    f32x2 v[16];
    {
      f32x2 tmp[16];
#pragma unroll
      for (int j = 0; j < 4; j++) {  // B^T d : combine patch rows
        tmp[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
        tmp[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
        tmp[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
        tmp[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {  // (B^T d) B : combine patch columns
        v[i * 4 + 0] = tmp[i * 4 + 0] - tmp[i * 4 + 2];
        v[i * 4 + 1] = tmp[i * 4 + 1] + tmp[i * 4 + 2];
        v[i * 4 + 2] = tmp[i * 4 + 2] - tmp[i * 4 + 1];
        v[i * 4 + 3] = tmp[i * 4 + 1] - tmp[i * 4 + 3];
      }
    }

    // ---- 16 points x 2 k-blocks x 2 k-steps of v_mfma_f32_16x16x4_f32 ------------
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const f32x2 b0 = *(const f32x2*)(st + b_base[0] + e * 2048);
      const f32x2 b1 = *(const f32x2*)(st + b_base[1] + e * 2048);
      acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e].x, b0.x, acc[e][0], 0, 0, 0);
      acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e].x, b1.x, acc[e][1], 0, 0, 0);
      acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e].y, b0.y, acc[e][0], 0, 0, 0);
      acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e].y, b1.y, acc[e][1], 0, 0, 0);
    }
  }

  // ---- epilogue: A^T m A, BN, ReLU, store (C/D layout: col = lane&15, row = 4*(lane>>4)+r)
  float sc[2], bi[2];
  int kcol[2];
#pragma unroll
  for (int cb = 0; cb < 2; cb++) {
    kcol[cb] = kb * KB + wk * 32 + cb * 16 + t16;
    sc[cb] = bnScale[kcol[cb]];
    bi[cb] = bnBias[kcol[cb]];
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int g = tb * TB + wt * 16 + 4 * h + r;
    if (g >= totalTiles) continue;
    const TileCoord tc = decode_tile(g);
    float* img = out + (size_t)tc.n * WINO_HW * WINO_HW * K;
    const int oy = 1 + 2 * tc.ty, ox = 1 + 2 * tc.tx;
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
      float t0[4], t1[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float m0 = acc[0 * 4 + j][cb][r], m1 = acc[1 * 4 + j][cb][r];
        const float m2 = acc[2 * 4 + j][cb][r], m3 = acc[3 * 4 + j][cb][r];
        t0[j] = m0 + m1 + m2;
        t1[j] = m1 - m2 - m3;
      }
      float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
      float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
      y00 = sc[cb] * y00 + bi[cb];
      y01 = sc[cb] * y01 + bi[cb];
      y10 = sc[cb] * y10 + bi[cb];
      y11 = sc[cb] * y11 + bi[cb];
      if (relu) {
        y00 = fmaxf(y00, 0.f);
        y01 = fmaxf(y01, 0.f);
        y10 = fmaxf(y10, 0.f);
        y11 = fmaxf(y11, 0.f);
      }
      float* o = img + kcol[cb];
      o[((oy)*WINO_HW + ox) * K] = y00;
      o[((oy)*WINO_HW + ox + 1) * K] = y01;
      o[((oy + 1) * WINO_HW + ox) * K] = y10;
      o[((oy + 1) * WINO_HW + ox + 1) * K] = y11;
      // zero ring (the next 3x3 layer's padding, Kernel128_winograd.cu:163,243)
      if (tc.ty == 0) {
        o[(ox)*K] = 0.f;
        o[(ox + 1) * K] = 0.f;
        if (tc.tx == 0) o[0] = 0.f;
        if (tc.tx == 6) o[15 * K] = 0.f;
      }
      if (tc.ty == 6) {
        o[(15 * WINO_HW + ox) * K] = 0.f;
        o[(15 * WINO_HW + ox + 1) * K] = 0.f;
        if (tc.tx == 0) o[(15 * WINO_HW) * K] = 0.f;
        if (tc.tx == 6) o[(15 * WINO_HW + 15) * K] = 0.f;
      }
      if (tc.tx == 0) {
        o[((oy)*WINO_HW) * K] = 0.f;
        o[((oy + 1) * WINO_HW) * K] = 0.f;
      }
      if (tc.tx == 6) {
        o[((oy)*WINO_HW + 15) * K] = 0.f;
        o[((oy + 1) * WINO_HW + 15) * K] = 0.f;
      }
    }
  }
}

}  // namespace
}  // namespace wino

using namespace wino;

extern "C" {

size_t wino_filter_f2_elems(int C, int K) { return (size_t)16 * C * K; }

static int check_ck(int C, int K) {
  if (C <= 0 || K <= 0 || (C % 8) != 0 || (K % 64) != 0) {
    set_error("unsupported channels C=%d K=%d (need C %% 8 == 0, K %% 64 == 0)", C, K);
    return WINO_E_SHAPE;
  }
  return WINO_OK;
}

int wino_filter_transform_f2(const float* w_kcrs, float* U, int C, int K, wino_stream_t s) {
  if (!w_kcrs || !U) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (int rc = check_ck(C, K)) return rc;
  const int n = C * K;
  hipLaunchKernelGGL(filter_transform_f2_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)s, w_kcrs, U, C, K);
  return launch_status("filter_transform_f2_kernel");
}

int wino_filter_import_f4(const float* u36, float* U, int C, int K, wino_stream_t s) {
  if (!u36 || !U) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (int rc = check_ck(C, K)) return rc;
  const int n = C * K;
  hipLaunchKernelGGL(filter_import_f4_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)s, u36, U, C, K);
  return launch_status("filter_import_f4_kernel");
}

int wino_conv3x3_bn_relu(const float* in, const float* U, const float* bnBias,
                         const float* bnScale, float* out, int N, int C, int K, int relu,
                         wino_stream_t s) {
  if (!in || !U || !bnBias || !bnScale || !out) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (int rc = check_ck(C, K)) return rc;
  if (N < 1 || (long)N * WINO_TILES > (1L << 30)) { set_error("bad batch N=%d", N); return WINO_E_SHAPE; }
  // raise the dynamic-LDS cap (128 KB of the CU's 160 KB) once per device
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  WINO_HIP(hipGetDevice(&dev));
  if (!((attr_done.load() >> (dev & 63)) & 1ull)) {
    WINO_HIP(hipFuncSetAttribute((const void*)wino_f2_fused_kernel,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_done.fetch_or(1ull << (dev & 63));
  }
  const int nTB = (N * WINO_TILES + TB - 1) / TB;
  const int grid = 8 * (K / KB) * ((nTB + 7) / 8);
  hipLaunchKernelGGL(wino_f2_fused_kernel, dim3(grid), dim3(NTHREADS), LDS_BYTES, (hipStream_t)s,
                     in, U, bnBias, bnScale, out, N, C, K, relu, nTB);
  return launch_status("wino_f2_fused_kernel");
}

}  // extern "C"
