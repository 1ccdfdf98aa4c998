// Developer tool: does a second wave's non-MFMA work slow a SIMD's MFMA stream?  512-thread
// workgroups: waves 0-3 (one per SIMD) issue back-to-back fp32 MFMAs; waves 4-7 (their SIMD
// mates) run MODE: 0 idle (exit), 1 packed VALU adds, 2 ds_read_b64 stream, 3 s_nop spin (SALU),
// 4 MFMAs too (the usual two-stream case), 5 ds_read_b64 + VALU mix.  Reports the MFMA waves'
// cycles per MFMA.   hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k(const float* __restrict__ in, float* __restrict__ out,
                                         unsigned long long* __restrict__ stamps, int iters) {
  __shared__ float lds[16384];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < 16384; i += 512) lds[i] = in[i & 4095];
  __syncthreads();
  if (w < 4 || MODE == 4) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(tid * 8 + i + 77) & 4095]; }
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = (f32x4){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; i++) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b[(i >> 2) & 7], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i & 7], a[(i >> 2) & 7], acc[i], 0, 0, 0);
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 32; i++) s += acc[i];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) stamps[blockIdx.x] = t1 - t0;
  } else if (MODE == 1) {
    f32x2 x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = (f32x2){in[tid & 4095], in[(tid + i) & 4095]};
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = x[i] + x[(i + 1) & 7];
    }
    out[blockIdx.x * 512 + tid] = x[0].x + x[3].y;
  } else if (MODE == 2 || MODE == 5) {
    f32x2 s = {0.f, 0.f};
    const f32x2* p = (const f32x2*)lds + (tid & 63);
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        f32x2 t = p[r * 64 + ((it & 7) << 10) / 8];
        if (MODE == 5) t = t + s;
        s += t;
      }
    }
    out[blockIdx.x * 512 + tid] = s.x + s.y;
  } else if (MODE == 3) {
    for (int it = 0; it < iters * 16; ++it) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
  }
}

template <int MODE>
void run(const char* what, const float* in, float* out, unsigned long long* st, int iters) {
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, in, out, st, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto c : h) cyc += c;
  printf("%-44s MFMA waves: %.1f cycles per MFMA\n", what, cyc / 256 / (iters * 64.0));
}

int main() {
  float *in, *out; unsigned long long* st;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&st, 256 * 8));
  std::vector<float> h(4096);
  for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  run<0>("mate idle", in, out, st, 400);
  run<1>("mate: packed VALU adds", in, out, st, 400);
  run<2>("mate: ds_read_b64 stream", in, out, st, 400);
  run<5>("mate: ds_read_b64 + VALU", in, out, st, 400);
  run<3>("mate: s_nop spin", in, out, st, 400);
  run<4>("mate: MFMAs too (per-wave figure)", in, out, st, 400);
  return 0;
}
