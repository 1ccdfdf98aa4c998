// Developer tool: prices the parts of the 1x1-conv GEMM kernel (ABLATE in csrc/conv1x1_kernel.h)
// on the four reference shapes at N = 128.  Not part of the library.   make tools
#include "conv1x1_kernel.h"

#include <cstdlib>
#include <vector>

namespace wino { void set_error(const char*, ...) {} int hip_fail(hipError_t, const char*) { return -1; } }
using namespace wino::gemm1x1;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NW, int AB>
float run(const float* A, const float* B, const float* b, const float* s, float* C, long M, int Cin, int Kout) {
  using G = Cfg<32, NW>;
  CK(hipFuncSetAttribute((const void*)(conv1x1_bn_kernel<32, NW, AB>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  const int nMB = (int)((M + BM - 1) / BM);
  const int grid = 8 * (Kout / G::BN) * ((nMB + 7) / 8);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&] { hipLaunchKernelGGL((conv1x1_bn_kernel<32, NW, AB>), dim3(grid), dim3(G::NT), G::LDS_BYTES, 0, A, B, b, s, (const float*)nullptr, C, M, Cin, Kout, 1, nMB, 0L, 0L, 0L, wino::gemm1x1::SkArgs{nullptr, nullptr}, wino::gemm1x1::make_padgeo(14, 14)); };
  for (int i = 0; i < 300; i++) launch();   // clock ramp: a burst from an idle chip runs at 2.05 GHz
  CK(hipEventRecord(e0));
  for (int i = 0; i < 100; i++) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 10.f;
}

template <int NW>
void sweep(const float* A, const float* B, const float* b, const float* s, float* C, long M, int Cin, int Kout) {
  const double ideal = 2.0 * M * Cin * Kout / 157.3e12 * 1e6;
  printf("%5d->%-5d NW=%d | %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f | mfma floor %.1f us\n", Cin, Kout, NW,
         run<NW, 0>(A, B, b, s, C, M, Cin, Kout), run<NW, 1>(A, B, b, s, C, M, Cin, Kout), run<NW, 2>(A, B, b, s, C, M, Cin, Kout),
         run<NW, 3>(A, B, b, s, C, M, Cin, Kout), run<NW, 8>(A, B, b, s, C, M, Cin, Kout), run<NW, 512>(A, B, b, s, C, M, Cin, Kout),
         run<NW, 4>(A, B, b, s, C, M, Cin, Kout), ideal);
}

// the bottleneck block's last layer: 256 -> 1024 + BN + skip + ReLU, A read from the padded 3x3 output
template <int NW>
float run_block_tail(const float* Apad, const float* B, const float* b, const float* s, const float* R, float* C, long M, int Cin, int Kout, int reps) {
  using G = Cfg<32, NW>;
  CK(hipFuncSetAttribute((const void*)(conv1x1_bn_kernel<32, NW, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  const int nMB = (int)((M + BM - 1) / BM);
  const int grid = 8 * (Kout / G::BN) * ((nMB + 7) / 8);
  const int flags = WINO_RELU | WINO_A_PADDED | WINO_ADD_RESIDUAL;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&] { hipLaunchKernelGGL((conv1x1_bn_kernel<32, NW, 0>), dim3(grid), dim3(G::NT), G::LDS_BYTES, 0, Apad, B, b, s, R, C, M, Cin, Kout, flags, nMB, 0L, 0L, 0L, wino::gemm1x1::SkArgs{nullptr, nullptr}, wino::gemm1x1::make_padgeo(14, 14)); };
  for (int i = 0; i < 5; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const long M = 128 * 196;
  if (argc > 1 && argv[1][0] == 'r') {   // residual mode: the block's last layer, after a clock ramp
    float *A, *B, *b, *s, *C, *R;
    CK(hipMalloc(&A, (size_t)128 * 256 * 256 * 4)); CK(hipMalloc(&B, 256 * 1024 * 4)); CK(hipMalloc(&C, M * 1024 * 4)); CK(hipMalloc(&R, M * 1024 * 4));
    CK(hipMalloc(&b, 4096)); CK(hipMalloc(&s, 4096));
    std::vector<float> h(M * 1024);
    for (auto& x : h) x = (float)(rand() & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(A, h.data(), (size_t)128 * 256 * 256 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(R, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), 256 * 1024 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(s, h.data() + 1024, 4096, hipMemcpyHostToDevice));
    run_block_tail<8>(A, B, b, s, R, C, M, 256, 1024, 3000);
    float t[3];
    for (int i = 0; i < 3; i++) t[i] = run_block_tail<8>(A, B, b, s, R, C, M, 256, 1024, 200);
    printf("%s 256->1024 + skip + ReLU (A padded), 8 waves, N=128: %.1f / %.1f / %.1f us\n", argv[0], t[0], t[1], t[2]);
    return 0;
  }
  float *A, *B, *b, *s, *C;
  CK(hipMalloc(&A, M * 1024 * 4)); CK(hipMalloc(&B, 1024 * 1024 * 4)); CK(hipMalloc(&C, M * 1024 * 4));
  CK(hipMalloc(&b, 4096)); CK(hipMalloc(&s, 4096));
  std::vector<float> h(M * 1024);
  for (auto& x : h) x = (float)(rand() & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), 1024 * 1024 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(s, h.data() + 1024, 4096, hipMemcpyHostToDevice));
  printf("us per launch, N=128 (M=25088)  |    full noA-DMA noB-DMA   noDMA  noSync noStore  noMFMA\n");
  sweep<4>(A, B, b, s, C, M, 512, 128);  sweep<8>(A, B, b, s, C, M, 512, 128);
  sweep<4>(A, B, b, s, C, M, 128, 512);  sweep<8>(A, B, b, s, C, M, 128, 512);
  sweep<4>(A, B, b, s, C, M, 1024, 256); sweep<8>(A, B, b, s, C, M, 1024, 256);
  sweep<4>(A, B, b, s, C, M, 256, 1024); sweep<8>(A, B, b, s, C, M, 256, 1024);
  return 0;
}
