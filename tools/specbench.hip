// Developer tool: prices a wave-SPECIALISED shape of the fused 3x3 loop before any kernel is written.
// 512-thread workgroups, one per CU, two waves per SIMD with different jobs:
//   waves 0-3 (consumers, one per SIMD): per iteration 64 v_mfma_f32_16x16x4_f32 fed by 16 A-fragment and
//     32 B-fragment ds_read_b64 (pinned, prefetched two steps ahead), nothing else;
//   waves 4-7 (producers): per iteration P LDS-DMA pieces of 1 KiB, 8 patch ds_read_b64, 16 packed adds,
//     8 ds_write_b64 of the transformed operand;
//   one vmcnt(0) (producers) + s_barrier per iteration for everybody.
// Reports cycles per iteration of the consumer waves; the matrix pipe's floor is 64 x 32 = 2048.
// MODE: 0 as above   1 producers idle (consumers alone)   2 no LDS-DMA   3 consumers also wait lgkmcnt(0) per step
//   hipcc --offload-arch=gfx950 -O3 tools/specbench.hip -o tools/specbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define DMA16(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), \
                                                          (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)

template <int MODE, int PIECES>
__global__ void __launch_bounds__(512, 2) spec(const float* __restrict__ src, float* __restrict__ out,
                                               unsigned long long* __restrict__ stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 40960; i += 512) ((float*)smem)[i] = src[i & 8191];
  __syncthreads();
  if (w < 4) {   // ---------------- consumer
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = (f32x4){0, 0, 0, 0};
    const int rd = lane * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      __builtin_amdgcn_s_barrier();
      const char* vst = smem + (it & 1) * 16384 + w * 4096;           // V stage: [16 pts][64 lanes] x 8 B per wave
      const char* ust = smem + 65536 + (it % 3) * 32768;              // U stage
      f32x2 a[16], b0[16], b1[16];
      a[0] = *(const f32x2*)(vst + rd); b0[0] = *(const f32x2*)(ust + rd); b1[0] = *(const f32x2*)(ust + 512 + rd);
      a[1] = *(const f32x2*)(vst + 512 + rd); b0[1] = *(const f32x2*)(ust + 2048 + rd); b1[1] = *(const f32x2*)(ust + 2048 + 512 + rd);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        if (e + 2 < 16) {
          a[e + 2] = *(const f32x2*)(vst + (e + 2) * 512 + rd);
          b0[e + 2] = *(const f32x2*)(ust + (e + 2) * 2048 + rd);
          b1[e + 2] = *(const f32x2*)(ust + (e + 2) * 2048 + 512 + rd);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) LGKM(0);
        else if (e + 2 < 16) LGKM(6);
        else if (e == 14) LGKM(3);
        else LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        MFMA(acc[2 * e], a[e].x, b0[e].x);
        MFMA(acc[2 * e + 1], a[e].x, b1[e].x);
        MFMA(acc[2 * e], a[e].y, b0[e].y);
        MFMA(acc[2 * e + 1], a[e].y, b1[e].y);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 32; i++) s += acc[i];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) stamps[blockIdx.x] = t1 - t0;
  } else {       // ---------------- producer
    const int p = w - 4;
    const char* gsrc = (const char*)src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
    f32x2 keep = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      if (MODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (MODE == 1) continue;
      if (MODE != 2) {
#pragma unroll
        for (int j = 0; j < PIECES; j++)
          DMA16(gsrc + (it & 31) * 1024 + j * 64, smem + 32768 + ((it + j) & 1) * 16384 + ((PIECES * p + j) & 15) * 1024);
      }
      f32x2 d[8];
#pragma unroll
      for (int i = 0; i < 8; i++) d[i] = *(const f32x2*)(smem + 32768 + (it & 1) * 16384 + p * 4096 + i * 512 + lane * 8);
      f32x2 v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = d[i] - d[(i + 2) & 7];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = v[i] + d[(i + 5) & 7];
#pragma unroll
      for (int i = 0; i < 8; i++) *(f32x2*)(smem + ((it + 1) & 1) * 16384 + p * 4096 + i * 512 + lane * 8) = v[i];
      keep += v[3];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + tid] = keep.x + keep.y;
  }
}

// The point-split 32x32x2 shape (PS32), all 8 waves alike (2 per SIMD): wave = 32 tiles x 32 out-channels x
// 8 of the 16 Winograd points = 8 accumulator tiles of v_mfma_f32_32x32x2_f32 (128 registers).  Per
// 8-channel iteration and wave: 32 MFMAs (64 cycles each), 12 patch reads + 8 filter reads (ds_read_b128: a
// lane holds 4 channels of one pixel / one filter tap), 32 packed adds (B^T d B for 4 channels x 8 points),
// 8 LDS-DMA pieces, one vmcnt(0) + barrier.  Floor: 32 x 64 x 2 waves = 4096 cycles per iteration per SIMD.
// V: 1 no LDS-DMA   2 no transform adds   4 filter reads only (no patch reads, no adds)
#define MFMA32(ACC, A, B) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ void __launch_bounds__(512, 2) ps32(const float* __restrict__ src, float* __restrict__ out,
                                               unsigned long long* __restrict__ stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 40960; i += 512) ((float*)smem)[i] = src[i & 8191];
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
  f32x4 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = (f32x4){src[(tid + i) & 8191], src[(tid + 2 * i) & 8191], src[(tid + 3 * i) & 8191], src[(tid + 5 * i) & 8191]};
  const char* gsrc = (const char*)src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  const int rd16 = lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* rst = smem + (it & 1) * 32768;
    const char* ust = smem + 65536 + (it % 3) * 32768;
    f32x4 d[12], b[8];
    b[0] = *(const f32x4*)(ust + rd16);
    b[1] = *(const f32x4*)(ust + 4096 + rd16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (e + 2 < 8) b[e + 2] = *(const f32x4*)(ust + (e + 2) * 4096 + rd16);
      if (!(V & 4) && e < 6) {
        d[2 * e] = *(const f32x4*)(rst + (2 * e) * 1024 + rd16);
        d[2 * e + 1] = *(const f32x4*)(rst + (2 * e + 1) * 1024 + rd16);
      }
      if (!(V & 1)) DMA16(gsrc + (it & 31) * 1024 + e * 64, smem + 98304 + (it & 1) * 32768 + ((e * 8 + w) % 32) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      if (e + 2 < 8) { if (!(V & 4) && e < 6) LGKM(3); else LGKM(1); } else LGKM(0);
      __builtin_amdgcn_sched_barrier(0);
      MFMA32(acc[e], v[e].x, b[e].x);
      MFMA32(acc[e], v[e].y, b[e].y);
      MFMA32(acc[e], v[e].z, b[e].z);
      MFMA32(acc[e], v[e].w, b[e].w);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(V & 4)) {
      if (!(V & 2)) {
        f32x4 t[8];
#pragma unroll
        for (int i = 0; i < 4; i++) { t[i] = d[i] - d[i + 8]; t[i + 4] = d[i + 4] + d[i + 8]; }
#pragma unroll
        for (int i = 0; i < 2; i++) {
          v[4 * i + 0] = t[4 * i + 0] - t[4 * i + 2];
          v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
          v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
          v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 12; i++) asm volatile("" :: "v"(d[i]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) s += acc[i][j];
  out[blockIdx.x * 512 + tid] = s + v[0].x;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}
template <int V>
void run32(const char* what, const float* in, float* out, unsigned long long* st, int iters) {
  CK(hipFuncSetAttribute((const void*)(ps32<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((ps32<V>), dim3(256), dim3(512), 163840, 0, in, out, st, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto c : h) cyc += c;
  printf("%-64s %6.0f cycles per iteration = %5.1f per 16x16x4-equivalent MFMA per SIMD (floor 32)\n", what, cyc / 256 / iters, cyc / 256 / iters / 128.0);
}

template <int MODE, int PIECES>
void run(const char* what, const float* in, float* out, unsigned long long* st, int iters) {
  CK(hipFuncSetAttribute((const void*)(spec<MODE, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((spec<MODE, PIECES>), dim3(256), dim3(512), 163840, 0, in, out, st, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto c : h) cyc += c;
  printf("%-64s %6.0f cycles per iteration = %5.1f per MFMA (floor 32)\n", what, cyc / 256 / iters, cyc / 256 / iters / 64.0);
}

int main() {
  float *in, *out; unsigned long long* st;
  CK(hipMalloc(&in, 64 * 65536 + 65536)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&st, 256 * 8));
  std::vector<float> h((64 * 65536 + 65536) / 4);
  for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; rep++) {
    run<1, 12>("consumers alone (producers only meet the barrier)", in, out, st, 400);
    run<2, 12>("producers: patch reads + adds + V writes, no LDS-DMA", in, out, st, 400);
    run<0, 8>("producers: 8 LDS-DMA pieces + transform", in, out, st, 400);
    run<0, 12>("producers: 12 LDS-DMA pieces + transform", in, out, st, 400);
    run<0, 16>("producers: 16 LDS-DMA pieces + transform", in, out, st, 400);
    run<3, 12>("12 pieces, consumers drain lgkmcnt every step", in, out, st, 400);
    run32<0>("PS32: point-split 32x32x2, the full mix", in, out, st, 400);
    run32<1>("PS32: no LDS-DMA", in, out, st, 400);
    run32<2>("PS32: no transform adds", in, out, st, 400);
    run32<4>("PS32: MFMA + filter reads + DMA + barrier only", in, out, st, 400);
    run32<5>("PS32: MFMA + filter reads + barrier only", in, out, st, 400);
  }
  return 0;
}
