// Developer tool: prices the parts of the 1x1-conv GEMM kernel (ABLATE in csrc/conv1x1_kernel.h)
// on the four reference shapes at N = 128.  Not part of the library.   make tools
#include "conv1x1_kernel.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
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
  auto launch = [&] { hipLaunchKernelGGL((conv1x1_bn_kernel<32, NW, AB>), dim3(grid), dim3(G::NT), G::LDS_BYTES, 0, A, B, b, s, (const float*)nullptr, C, M, Cin, Kout, 1, nMB, 0L, 0L, 0L, wino::gemm1x1::SkArgs{nullptr, nullptr, nullptr}, wino::gemm1x1::make_padgeo(14, 14)); };
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
  CK(hipFuncSetAttribute((const void*)(conv1x1_bn_kernel<32, NW, 0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  const int nMB = (int)((M + BM - 1) / BM);
  const int grid = 8 * (Kout / G::BN) * ((nMB + 7) / 8);
  const int flags = WINO_RELU | WINO_A_PADDED | WINO_ADD_RESIDUAL;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&] { hipLaunchKernelGGL((conv1x1_bn_kernel<32, NW, 0, false, true>), dim3(grid), dim3(G::NT), G::LDS_BYTES, 0, Apad, B, b, s, R, C, M, Cin, Kout, flags, nMB, 0L, 0L, 0L, wino::gemm1x1::SkArgs{nullptr, nullptr, nullptr}, wino::gemm1x1::make_padgeo(14, 14)); };
  for (int i = 0; i < 5; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

// timeline of one launch (plain or stream-K form): chip-wide 100 MHz stamps per workgroup
template <int NW, bool SKF>
void timeline(const float* A, const float* B, const float* b, const float* s, float* C, long M, int Cin, int Kout, int Gsk) {
  using G = Cfg<32, NW>;
  CK(hipFuncSetAttribute((const void*)(conv1x1_bn_kernel<32, NW, 32768, SKF>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)(conv1x1_bn_kernel<32, NW, 0, SKF>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  const int nMB = (int)((M + BM - 1) / BM);
  const int grid = SKF ? Gsk : 8 * (Kout / G::BN) * ((nMB + 7) / 8);
  wino::gemm1x1::SkArgs sk{nullptr, nullptr, nullptr};
  CK(hipMalloc(&sk.dbg, (size_t)grid * 64));
  if (SKF) {
    CK(hipMalloc(&sk.slabs, (size_t)2 * grid * NW * RB * 1024));
    CK(hipMalloc(&sk.tickets, (size_t)nMB * (Kout / G::BN) * 4));
    CK(hipMemset(sk.tickets, 0, (size_t)nMB * (Kout / G::BN) * 4));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](auto ab) { hipLaunchKernelGGL((conv1x1_bn_kernel<32, NW, decltype(ab)::value, SKF>), dim3(grid), dim3(G::NT), G::LDS_BYTES, 0, A, B, b, s, (const float*)nullptr, C, M, Cin, Kout, 1, nMB, 0L, 0L, 0L, sk, wino::gemm1x1::make_padgeo(14, 14)); };
  for (int i = 0; i < 3000; i++) launch(std::integral_constant<int, 0>{});
  CK(hipEventRecord(e0));
  for (int i = 0; i < 200; i++) launch(std::integral_constant<int, 0>{});
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const float us_prod = ms * 5.f;
  for (int i = 0; i < 50; i++) launch(std::integral_constant<int, 32768>{});
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st((size_t)grid * 8);
  CK(hipMemcpy(st.data(), sk.dbg, st.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, tend = 0;
  for (int l = 0; l < grid; l++) { t0 = std::min(t0, st[8 * l]); tend = std::max(tend, st[8 * l + 3]); }
  std::vector<double> entry, first, lastep, exitt, epi;
  for (int l = 0; l < grid; l++) {
    entry.push_back((st[8 * l] - t0) * 0.01); first.push_back((st[8 * l + 1] - st[8 * l]) * 0.01);
    lastep.push_back((st[8 * l + 2] - t0) * 0.01); exitt.push_back((st[8 * l + 3] - t0) * 0.01); epi.push_back((st[8 * l + 3] - st[8 * l + 2]) * 0.01);
  }
  auto pr = [&](const char* name, std::vector<double> v) {
    std::sort(v.begin(), v.end());
    printf("  %-34s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us\n", name, v[0], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  };
  double cyc = 0, rt = 0;
  for (int l = 0; l < grid; l++) { cyc += (double)(st[8 * l + 6] - st[8 * l + 5]); rt += (double)(st[8 * l + 3] - st[8 * l]); }
  printf("%d->%d NW=%d %s grid=%d: product %.2f us per launch; first entry -> last exit %.2f us; in-kernel clock %.3f GHz\n", Cin, Kout, NW, SKF ? "stream-K" : "plain", grid, us_prod, (tend - t0) * 0.01, cyc / rt * 0.1);
  pr("entry (after the first entry)", entry); pr("entry -> first stage issued", first); pr("start of the last epilogue", lastep); pr("exit", exitt); pr("last epilogue -> exit", epi);
  if (getenv("TL_VERBOSE"))
    for (int l = 0; l < grid; l++) {
      const unsigned hw = (unsigned)st[8 * l + 4], xcc = (unsigned)(st[8 * l + 4] >> 32);
      printf("wg %d cu %u.%u.%u entry %.2f last_epi %.2f exit %.2f\n", l, xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15, entry[l], lastep[l], exitt[l]);
    }
}

int main(int argc, char** argv) {
  const long M = 128 * 196;
  if (argc > 1 && argv[1][0] == 't') {
    float *A, *B, *b, *s, *C;
    CK(hipMalloc(&A, M * 1024 * 4)); CK(hipMalloc(&B, 1024 * 1024 * 4)); CK(hipMalloc(&C, M * 1024 * 4));
    CK(hipMalloc(&b, 4096)); CK(hipMalloc(&s, 4096));
    std::vector<float> h(M * 1024);
    for (auto& x : h) x = (float)(rand() & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), 1024 * 1024 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(s, h.data() + 1024, 4096, hipMemcpyHostToDevice));
    timeline<8, true>(A, B, b, s, C, M, 1024, 256, 512);
    timeline<8, false>(A, B, b, s, C, M, 1024, 256, 0);
    timeline<8, false>(A, B, b, s, C, M, 256, 1024, 0);
    timeline<4, false>(A, B, b, s, C, M, 128, 512, 0);
    timeline<4, false>(A, B, b, s, C, M, 512, 128, 0);
    return 0;
  }
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
