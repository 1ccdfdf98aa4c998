// Independent GPU comparator for the 3x3 layers: direct (non-Winograd) cross-correlation
// + folded BN + ReLU, one thread per output element, plain fp32 FMA over (c, r, s).
// It fills the role cuDNN plays in the reference's host drivers
// (Kernel128_winograd.cu:382-404): a second implementation the custom path is diffed
// against at run time.  Deliberately simple; it is not the product path.
#include "wino_common.h"

namespace wino {
namespace {

__global__ void conv3x3_direct_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                      const float* __restrict__ bnBias,
                                      const float* __restrict__ bnScale, float* __restrict__ out,
                                      int N, int Hp, int Wp, int C, int K, int relu) {
  // thread -> (n, oy, ox, k) over the full padded Hp x Wp output; ring threads store 0
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)N * Hp * Wp * K;
  if (idx >= total) return;
  const int k = (int)(idx % K);
  const long p = idx / K;
  const int ox = (int)(p % Wp);
  const int oy = (int)((p / Wp) % Hp);
  const int n = (int)(p / ((long)Hp * Wp));
  if (ox < 1 || ox > Wp - 2 || oy < 1 || oy > Hp - 2) {
    out[idx] = 0.f;
    return;
  }
  // output (oy,ox) in padded coords = valid-conv pixel (oy-1, ox-1): taps in[oy-1+r][ox-1+s]
  const float* ip = in + ((size_t)(n * Hp + oy - 1) * Wp + ox - 1) * C;
  const float* wp = w + (size_t)k * C * 9;  // [K][C][3][3]
  float s = 0.f;
  for (int c = 0; c < C; c++) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int q = 0; q < 3; q++)
        s = fmaf(ip[(size_t)(r * Wp + q) * C + c], wp[c * 9 + r * 3 + q], s);
  }
  float y = bnScale[k] * s + bnBias[k];
  if (relu) y = fmaxf(y, 0.f);
  out[idx] = y;
}

}  // namespace
}  // namespace wino

using namespace wino;

static int direct_launch(const float* in, const float* w_kcrs, const float* bnBias, const float* bnScale,
                         float* out, int N, int H, int W, int C, int K, int relu, wino_stream_t s) {
  if (!in || !w_kcrs || !bnBias || !bnScale || !out) { set_error("NULL pointer"); return WINO_E_ARG; }
  if (N < 1 || C < 1 || K < 1 || H < 1 || W < 1) { set_error("bad shape"); return WINO_E_SHAPE; }
  const long total = (long)N * (H + 2) * (W + 2) * K;
  if ((total + 255) / 256 > 0x7fffffffL) { set_error("too large"); return WINO_E_SHAPE; }
  hipLaunchKernelGGL(conv3x3_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)s, in, w_kcrs, bnBias, bnScale, out, N, H + 2, W + 2, C, K, relu);
  return launch_status("conv3x3_direct_kernel");
}

extern "C" int wino_conv3x3_direct(const float* in, const float* w_kcrs, const float* bnBias,
                                   const float* bnScale, float* out, int N, int C, int K, int relu,
                                   wino_stream_t s) {
  return direct_launch(in, w_kcrs, bnBias, bnScale, out, N, WINO_PQ, WINO_PQ, C, K, relu, s);
}

extern "C" int wino_conv3x3_direct_hw(const float* in, const float* w_kcrs, const float* bnBias,
                                      const float* bnScale, float* out, int N, int H, int W, int C, int K,
                                      int relu, wino_stream_t s) {
  return direct_launch(in, w_kcrs, bnBias, bnScale, out, N, H, W, C, K, relu, s);
}
