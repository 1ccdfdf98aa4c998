// Developer tool: fp32 MFMA issue-rate ceiling on this device, and the clock it holds.
// 512-thread workgroups (2 waves/SIMD) or 256-thread (1 wave/SIMD), 32 independent 16x16x4
// accumulators per wave, operands in registers, random-ish data. Reports TFLOP/s and the
// in-kernel clock (s_memtime cycles / s_memrealtime 100 MHz ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// PAT: 1 = each accumulator twice in a row (dependent back-to-back), 2 = a0 a1 a0 a1 (dependence at
// distance 2, the fused kernel's order), 4 = a0 a1 a2 a3 a0 a1 a2 a3 (distance 4)
template <int NT, int PAT = 1>
__global__ void __launch_bounds__(NT) mfma_loop(const float* __restrict__ in, float* __restrict__ out,
                                                unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(tid * 8 + i + 77) & 4095]; }
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; i++) acc[i] = (f32x4){0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 32; g += PAT) {
#pragma unroll
      for (int i = g; i < g + PAT; i++)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b[(i >> 2) & 7], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = g; i < g + PAT; i++)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i & 7], a[(i >> 2) & 7], acc[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 32; i++) s += acc[i];
  out[blockIdx.x * NT + tid] = s[0] + s[1] + s[2] + s[3];
  if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NT, int PAT = 1>
void run(const float* in, float* out, unsigned long long* st, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((mfma_loop<NT, PAT>), dim3(blocks), dim3(NT), 0, 0, in, out, st, iters);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((mfma_loop<NT, PAT>), dim3(blocks), dim3(NT), 0, 0, in, out, st, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps;
  const double flops = (double)blocks * (NT / 64) * iters * 64.0 * 2048.0;
  std::vector<unsigned long long> h(blocks * 2);
  CK(hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks; i++) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  printf("pattern %d  threads/WG %4d blocks %5d iters %5d : %8.1f us  %7.1f TFLOP/s  in-kernel clock %.3f GHz  cycles/MFMA/SIMD %.1f\n",
         PAT, NT, blocks, iters, us, flops / us / 1e6, cyc / rt * 0.1, (cyc / blocks) / (iters * 64.0 * (NT / 256)));
}

int main() {
  float *in, *out; unsigned long long* st;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 2048 * 512 * 4)); CK(hipMalloc(&st, 2048 * 16));
  std::vector<float> h(4096);
  for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  run<256>(in, out, st, 256, 100);
  run<256>(in, out, st, 256, 1000);
  run<512>(in, out, st, 256, 50);
  run<512>(in, out, st, 256, 500);
  run<512>(in, out, st, 256, 2000);
  run<512>(in, out, st, 392, 32);
  run<256, 2>(in, out, st, 256, 1000);
  run<512, 2>(in, out, st, 256, 500);
  run<256, 4>(in, out, st, 256, 1000);
  run<512, 4>(in, out, st, 256, 500);
  return 0;
}
