// Developer tool: a synthetic loop with the fused 3x3 kernel's per-iteration instruction mix (8 waves,
// 2 per SIMD: 64 MFMAs, 32 B-operand LDS reads prefetched one step ahead, 16 patch LDS reads,
// 8 LDS-DMA KiB, 32 packed adds, one vmcnt(0)+barrier), reproducing its ~5150 cycles per iteration,
// plus variants that change one ingredient at a time.  Reports cycles per (16x16x4-equivalent) MFMA
// per SIMD; the pipe's floor is 32.
//   V bits: 1 B reads as one b128 per step   2 no LDS-DMA      4 LDS-DMA in 4-byte pieces (4x the count)
//           8 no patch reads                16 no packed adds  32 32x32x2 MFMAs (half the B reads)
//          64 16 waves x 64 accumulators (a row of 4 points per wave; 4 waves per SIMD)
// Use it to rule shapes out, not to predict gains: it matches the kernel's total but not its
// sensitivities (the kernel prefetches two points ahead with counted waits; here it is one).  The
// b128 variant's -5.6 % was -0.9 % of loop cycles and nothing in launch time in the real kernel,
// and cost the latency kernel 11-15 %.
//   make tools/mixbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define DMA_N(SRC, DST, BYTES) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), \
                                   (__attribute__((address_space(3))) void*)(DST), BYTES, 0, 0)
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define MFMA32(ACC, A, B) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

template <int V>
__global__ void __launch_bounds__((V & 64) ? 1024 : 512)
mix(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool W16 = V & 64, B128 = V & 1, NODMA = V & 2, DMA4 = V & 4, NORAW = V & 8, NOVALU = V & 16, M32 = V & 32;
  constexpr int NT = W16 ? 1024 : 512, STEPS = W16 ? 4 : 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 40960; i += NT) ((float*)smem)[i] = src[i & 8191];
  __syncthreads();
  f32x4 acc[W16 ? 16 : 32];
  f32x16 acc32[M32 ? 8 : 1];
#pragma unroll
  for (int i = 0; i < (W16 ? 16 : 32); i++) acc[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < (M32 ? 8 : 1); i++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc32[i][j] = 0.f;
  f32x2 v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = (f32x2){src[(tid + i) & 8191], src[(tid + 2 * i) & 8191]};
  const char* gsrc = (const char*)src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  const int rd = lane * 8, rd16 = lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int stage = (it & 1) * 32768;
    f32x2 d[16];
    f32x4 b[2];   // {kt0.x, kt0.y, kt1.x, kt1.y} for a step (8-wave mixes); two of them per step for W16
    f32x4 b2[2];
    auto load_b = [&](int e, f32x4& dst, f32x4& dst2) {
      if (W16) {
        dst = *(const f32x4*)(smem + 65536 + e * 4096 + rd16);
        dst2 = *(const f32x4*)(smem + 65536 + e * 4096 + 2048 + rd16);
      } else if (B128 || M32) {
        if (M32) { if (!(e & 1)) dst = *(const f32x4*)(smem + 65536 + e * 1024 + rd16); }
        else dst = *(const f32x4*)(smem + 65536 + e * 2048 + rd16);
      } else {
        f32x2 lo = *(const f32x2*)(smem + 65536 + e * 2048 + rd);
        f32x2 hi = *(const f32x2*)(smem + 65536 + e * 2048 + 512 + rd);
        dst = (f32x4){lo.x, lo.y, hi.x, hi.y};
      }
    };
    load_b(0, b[0], b2[0]);
#pragma unroll
    for (int e = 0; e < STEPS; e++) {
      if (e + 1 < STEPS) load_b(e + 1, b[(M32 ? ((e + 1) >> 1) : (e + 1)) & 1], b2[(e + 1) & 1]);
      if (!NORAW && (W16 || e < 8)) {
        d[2 * e] = *(const f32x2*)(smem + stage + e * 1024 + rd);
        d[2 * e + 1] = *(const f32x2*)(smem + stage + e * 1024 + 512 + rd);
      }
      if (!NODMA && (W16 || (e >= 4 && e < 12))) {
        const int pc = W16 ? e * 16 + w : (e - 4) * 8 + w;
        char* dst = smem + 98304 + (it & 1) * 32768 + (pc % 32) * 1024;
        const char* s = gsrc + (it & 31) * 1024 + e * 64;
        if (DMA4) { DMA_N(s, dst, 4); DMA_N(s + 4, dst + 256, 4); DMA_N(s + 8, dst + 512, 4); DMA_N(s + 12, dst + 768, 4); }
        else DMA_N(s, dst, 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      // wait for the B operand requested one step ago: leave this step's requests in flight
      if (e == 0 || e + 1 == STEPS) { if (e == 0) LGKM(0); else if (W16 && !NORAW) LGKM(2); else LGKM(0); }
      else if (W16) { if (NORAW) LGKM(2); else LGKM(4); }
      else {
        const int mine = ((B128 || M32) ? ((M32 && (e & 1) == 0) ? 0 : 1) : 2) + ((!NORAW && e < 8) ? 2 : 0);
        if (mine == 0) LGKM(0); else if (mine == 1) LGKM(1); else if (mine == 2) LGKM(2); else if (mine == 3) LGKM(3); else LGKM(4);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (W16) {
        MFMA(acc[4 * e], v[e].x, b[e & 1].x); MFMA(acc[4 * e + 1], v[e].x, b[e & 1].z);
        MFMA(acc[4 * e + 2], v[e].x, b2[e & 1].x); MFMA(acc[4 * e + 3], v[e].x, b2[e & 1].z);
        MFMA(acc[4 * e], v[e].y, b[e & 1].y); MFMA(acc[4 * e + 1], v[e].y, b[e & 1].w);
        MFMA(acc[4 * e + 2], v[e].y, b2[e & 1].y); MFMA(acc[4 * e + 3], v[e].y, b2[e & 1].w);
      } else if (M32) {
        const f32x4 bb = b[(e >> 1) & 1];
        MFMA32(acc32[(2 * e) & 7], v[e].x, (e & 1) ? bb.z : bb.x);
        MFMA32(acc32[(2 * e + 1) & 7], v[e].y, (e & 1) ? bb.w : bb.y);
      } else {
        MFMA(acc[2 * e], v[e].x, b[e & 1].x);
        MFMA(acc[2 * e + 1], v[e].x, b[e & 1].z);
        MFMA(acc[2 * e], v[e].y, b[e & 1].y);
        MFMA(acc[2 * e + 1], v[e].y, b[e & 1].w);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    LGKM(0);
    if (!NOVALU && !NORAW) {
      if (!W16) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = d[i] + d[(i + 5) & 15];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = v[i] - d[(i + 3) & 15];
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = d[i] + d[i + 4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = v[i] - v[(i + 1) & 3];
      }
    } else if (!NORAW) {
#pragma unroll
      for (int i = 0; i < (W16 ? 8 : 16); i++) asm volatile("" :: "v"(d[i]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < (W16 ? 16 : 32); i++) s += acc[i];
  float s32 = 0;
#pragma unroll
  for (int i = 0; i < (M32 ? 8 : 1); i++)
#pragma unroll
    for (int j = 0; j < 16; j++) s32 += acc32[i][j];
  out[blockIdx.x * NT + tid] = s[0] + s[1] + s[2] + s[3] + v[0].x + s32;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* what, const float* in, float* out, unsigned long long* st, int iters) {
  CK(hipFuncSetAttribute((const void*)(mix<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((mix<V>), dim3(256), dim3((V & 64) ? 1024 : 512), 163840, 0, in, out, st, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto c : h) cyc += c;
  printf("%-52s %5.1f cycles/MFMA/SIMD  (%4.0f per iteration)\n", what, cyc / 256 / (iters * 128.0), cyc / 256 / iters);
}

int main() {
  float *in, *out; unsigned long long* st;
  CK(hipMalloc(&in, 64 * 65536 + 65536)); CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&st, 256 * 8));
  std::vector<float> h((64 * 65536 + 65536) / 4);
  for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<0>("the fused kernel's mix", in, out, st, 200);
  run<1>("B operand as one b128 read per step", in, out, st, 200);
  run<2>("no LDS-DMA", in, out, st, 200);
  run<4>("LDS-DMA in 4-byte pieces (4x the instructions)", in, out, st, 200);
  run<8>("no patch reads, no adds", in, out, st, 200);
  run<16>("no packed adds", in, out, st, 200);
  run<2 | 8>("MFMA + B reads + barrier only", in, out, st, 200);
  run<1 | 2 | 8>("MFMA + b128 B reads + barrier only", in, out, st, 200);
  run<32>("32x32x2 MFMAs, half the B reads", in, out, st, 200);
  run<32 | 2>("32x32x2 MFMAs, no LDS-DMA", in, out, st, 200);
  run<64>("16 waves x 64 accumulators", in, out, st, 200);
  run<64 | 2>("16 waves, no LDS-DMA", in, out, st, 200);
  return 0;
}
