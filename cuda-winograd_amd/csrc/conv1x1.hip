// 1x1 convolution as an fp32 MFMA GEMM with fused folded-BN (+ReLU) epilogue, gfx950.
//
// Replaces kernel_512_one_128 / kernel_128_one_512 (Kernel128_one.cu:24-54,244-273) and
// kernel_1024_one_256 / kernel_256_one_1024 (Kernel256_one.cu:26-56,246-274):
//   C[m][k] = act( bnScale[k] * sum_c A[m][c] * B[c][k] + bnBias[k] )
// with A [M][Cin] (pixels x in-channels, the reference's HWC-flat activations),
// B [Cin][Kout] row-major exactly as the reference stores it (Kernel128_one.cu:40-42).
//
// Tiling: workgroup = 256 threads (4 waves) computes BM=112 x BN=128 of C; K-loop over Cin
// in steps of BK=32, two LDS stages filled by LDS-DMA.  BM = 7 MFMA row blocks because the
// reference's M = N*196 = 2^a * 49: 112-row tiles cover it exactly (25088 = 224 * 112) and
// 224 tiles fill one round of 256 CUs at Kout = 128.  Wave w owns columns [32w, 32w+32):
// 7 x 2 accumulator tiles of v_mfma_f32_16x16x4_f32 (56 acc VGPRs).
//
// LDS images (16-byte units XOR-permuted on the DMA source side so that fragment reads are
// bank-conflict free):
//   A stage [112 rows][8 units]: unit' = unit ^ ((row>>1)&7); A fragments by ds_read_b128
//   B stage [32 k][32 units]   : unit' = unit ^ (4*((k>>2)&1)); B fragments by ds_read_b32
#include "wino_common.h"

namespace wino {
namespace {

constexpr int BM = 112, BN = 128, BK = 32, NT = 256;
constexpr int RB = BM / 16;              // 7 row blocks
constexpr int A_BYTES = BM * BK * 4;     // 14336
constexpr int B_BYTES = BK * BN * 4;     // 16384
constexpr int STAGE = A_BYTES + B_BYTES; // 30720
constexpr int LDS_BYTES = 2 * STAGE;     // 61440
constexpr int A_WAVE_INSTR = A_BYTES / 1024;  // 14

__global__ void __launch_bounds__(NT, 2)
conv1x1_bn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                  const float* __restrict__ bnBias, const float* __restrict__ bnScale,
                  float* __restrict__ Cout, long M, int Cin, int Kout, int relu, int nMB) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // blocks that share a row tile (same A rows) are adjacent in `slot` on one XCD
  const int NBLK = Kout / BN;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nb = slot % NBLK;
  const int mb = (slot / NBLK) * 8 + xcd;
  if (mb >= nMB) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long m0 = (long)mb * BM;
  const int n0 = nb * BN;

  // ---- DMA sources --------------------------------------------------------------
  // A: wave-instruction q (0..13) covers rows 8q..8q+7; lane -> row 8q + lane/8, unit' lane%8
  // wave w issues q = w, w+4, w+8, (w+12 if < 14)
  const float* a_src[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int q = w + 4 * j;
    const int row = 8 * q + (lane >> 3);
    const int unit = (lane & 7) ^ ((row >> 1) & 7);
    long gr = m0 + row;
    gr = gr < M ? gr : M - 1;  // clamp: padded rows read a valid row
    a_src[j] = A + gr * Cin + unit * 4;
  }
  // B: wave-instruction q (0..15) covers k rows 2q, 2q+1; lane -> k = 2q + lane/32, unit' lane%32
  const float* b_src[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int q = w + 4 * j;
    const int k = 2 * q + (lane >> 5);
    const int unit = (lane & 31) ^ (((k >> 2) & 1) << 2);
    b_src[j] = B + (size_t)k * Kout + n0 + unit * 4;
  }
  auto issue = [&](int stage, int kc) {
    char* sb = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int q = w + 4 * j;
      if (q < A_WAVE_INSTR) dma16(a_src[j] + kc * BK, sb + q * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int q = w + 4 * j;
      dma16(b_src[j] + (size_t)kc * BK * Kout, sb + A_BYTES + q * 1024);
    }
  };

  // ---- fragment addresses ---------------------------------------------------------
  const int r16 = lane & 15, h = lane >> 4;
  const int a_sw = (r16 >> 1) & 7;
  // A row rb*16 + r16, k sub-chunk s (16 wide), lane reads unit (4s + h) -> unit' = (4s+h)^a_sw
  int a_off[2];
#pragma unroll
  for (int s = 0; s < 2; s++) a_off[s] = r16 * 128 + (((4 * s + h) ^ a_sw) << 4);
  // B element (k = 16s + 4h + j, col = 32w + 16cb + r16): float index k*128 + (col ^ 16*(h&1))
  int b_off[2];
#pragma unroll
  for (int cb = 0; cb < 2; cb++)
    b_off[cb] = A_BYTES + ((4 * h) * BN + ((32 * w + 16 * cb + r16) ^ ((h & 1) << 4))) * 4;

  f32x4 acc[RB][2];
#pragma unroll
  for (int i = 0; i < RB; i++) {
    acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int nk = Cin / BK;
  issue(0, 0);
  for (int it = 0; it < nk; ++it) {
    wait_vmem_all();
    __syncthreads();
    if (it + 1 < nk) issue((it + 1) & 1, it + 1);
    const char* st = smem + (it & 1) * STAGE;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      float b[2][4];
#pragma unroll
      for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          b[cb][j] = *(const float*)(st + b_off[cb] + (16 * s + j) * BN * 4);
#pragma unroll
      for (int rb = 0; rb < RB; rb++) {
        const f32x4 a = *(const f32x4*)(st + rb * 2048 + a_off[s]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[0][j], acc[rb][0], 0, 0, 0);
          acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[1][j], acc[rb][1], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: BN (+ReLU), C/D layout col = lane&15, row = 4*(lane>>4)+i ---------
#pragma unroll
  for (int cb = 0; cb < 2; cb++) {
    const int col = n0 + 32 * w + 16 * cb + r16;
    const float sc = bnScale[col], bi = bnBias[col];
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const long row = m0 + rb * 16 + 4 * h + i;
        if (row < M) {
          float y = sc * acc[rb][cb][i] + bi;
          if (relu) y = fmaxf(y, 0.f);
          Cout[row * Kout + col] = y;
        }
      }
    }
  }
}

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

extern "C" {

int wino_conv1x1_bn(const float* A, const float* B, const float* bnBias, const float* bnScale,
                    float* C, long M, int Cin, int Kout, int relu, wino_stream_t s) {
  if (!A || !B || !bnBias || !bnScale || !C) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (M < 1 || Cin <= 0 || Kout <= 0 || (Cin % BK) != 0 || (Kout % BN) != 0) {
    set_error("unsupported 1x1 shape M=%ld Cin=%d Kout=%d (need Cin %% %d == 0, Kout %% %d == 0)",
              M, Cin, Kout, BK, BN);
    return WINO_E_SHAPE;
  }
  const long nMBl = (M + BM - 1) / BM;
  if (nMBl > (1L << 24)) { set_error("M too large"); return WINO_E_SHAPE; }
  const int nMB = (int)nMBl;
  const int grid = 8 * (Kout / BN) * ((nMB + 7) / 8);
  hipLaunchKernelGGL(conv1x1_bn_kernel, dim3(grid), dim3(NT), LDS_BYTES, (hipStream_t)s, A, B,
                     bnBias, bnScale, C, M, Cin, Kout, relu, nMB);
  return launch_status("conv1x1_bn_kernel");
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
