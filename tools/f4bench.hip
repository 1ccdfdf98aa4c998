// Developer tool: prices the main loop of a FUSED F(4x4,3x3) kernel -- the reference's own algorithm
// (Kernel128_winograd.cu:41-72 B^T, :136-147 A^T, :186-213 the 36 point GEMMs) -- before any kernel is written,
// next to the loop of the product's fused F(2x2,3x3) kernel, in the same units.  Loop only: no epilogue, synthetic
// conflict-free LDS addresses, same instruction mix per iteration as a real kernel would have.
//
// Shape priced (the one the register file allows at two waves per SIMD):
//   workgroup = 8 waves, item = 64 tiles (4 whole 16x16 images: 16 tiles of 4x4 outputs each) x 32 out-channels
//   wave (wt, wk) = 16 tiles x 16 out-channels x ALL 36 points = 36 accumulator tiles (144 VGPRs)
//   per 8-channel iteration and wave: 36 patch reads (ds_read_b64: 6x6 pixels x 2 channels), B^T d B on 6x6
//   (12 vectors of 6, 12 operations each in the factored form = 144 packed or 288 scalar VALU ops), 36 filter
//   fragment reads (ds_read_b64), 72 v_mfma_f32_16x16x4_f32, 9 LDS-DMA pieces (raw images de-duplicated:
//   4 images x 256 px x 8 ch = 32 KB, filters 36 x 32 x 8 = 36 KB per stage), one vmcnt(0) + barrier.
//   V cannot be double-buffered across iterations (72 + 72 + 144 registers): a wave transforms, then multiplies;
//   the SIMD's other wave is what overlaps.  STAGGER = 1 runs waves 4-7 half an iteration out of phase (they
//   multiply with the V of the previous iteration while waves 0-3 transform, then transform the next raw stage).
// Cost per output, for the comparison with F(2x2): a 14x14 image is 16 F(4x4) tiles (23 % clipped away) or 49
// F(2x2) tiles.  Per image, out-channel and 8 input channels: F4 = 16 tiles x 36 points x 2 / 16 = 72 MFMA lane-
// slots ... in workgroup terms: F4 576 MFMAs per 4 images x 32 k, F2 512 MFMAs per 64/49 images x 64 k:
//   MFMAs per (image, 64 out-channels, 8 channels):  F4 = 576 / 4 * 2 = 288,   F2 = 512 * 49 / 64 = 392   (1.36 x)
// so F(4x4) wins per output when its cycles per MFMA stay below 1.36 x F(2x2)'s.
//   hipcc --offload-arch=gfx950 -O3 tools/f4bench.hip -o tools/f4bench && tools/f4bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define DMA16(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), \
                                                          (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
__device__ __forceinline__ void lgkm(int n) {
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0xC07F); break;
    case 1: __builtin_amdgcn_s_waitcnt(0xC17F); break;
    case 2: __builtin_amdgcn_s_waitcnt(0xC27F); break;
    case 3: __builtin_amdgcn_s_waitcnt(0xC37F); break;
    case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
    case 5: __builtin_amdgcn_s_waitcnt(0xC57F); break;
    case 6: __builtin_amdgcn_s_waitcnt(0xC67F); break;
    default: __builtin_amdgcn_s_waitcnt(0xCF7F); break;
  }
}

// B^T x on a vector of 6 (the reference's B^T, Kernel128_winograd.cu:44-70), factored: 12 operations
//   r0 = 4 x0 - 5 x2 + x4          a = x4 - 4 x2, b = x3 - 4 x1:  r1 = a + b, r2 = a - b
//   c = x4 - x2, e = x3 - x1:  r3 = c + 2 e, r4 = c - 2 e       r5 = 4 x1 - 5 x3 + x5
template <bool SCALAR>
__device__ __forceinline__ void bt6(f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3, f32x2& x4, f32x2& x5, float five) {
  if (!SCALAR) {
    const f32x2 m5 = {-five, -five};
    const f32x2 t = m5 * x2 + x4, r0 = 4.f * x0 + t;
    const f32x2 a = -4.f * x2 + x4, b = -4.f * x1 + x3;
    const f32x2 c = x4 - x2, e = x3 - x1;
    const f32x2 u = m5 * x3 + x5, r5 = 4.f * x1 + u;
    x0 = r0; x1 = a + b; x2 = a - b; x3 = 2.f * e + c; x4 = -2.f * e + c; x5 = r5;
  } else {
    // the same twelve operations per channel as plain v_fma_f32 / v_add_f32 / v_sub_f32 (asm: the SLP vectoriser
    // would fuse C++ scalars back into packed ops)
    f32x2 r0, r1, r2, r3, r4, r5;
#define FMA(D, A, B, C) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(D) : "v"(A), "v"(B), "v"(C))
#define FMAI(D, IMM, B, C) asm("v_fma_f32 %0, " IMM ", %1, %2" : "=v"(D) : "v"(B), "v"(C))
#define FMAS(D, S, B, C) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(D) : "s"(S), "v"(B), "v"(C))
#define ADD(D, A, B) asm("v_add_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define SUB(D, A, B) asm("v_sub_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
    const float m5 = -five;
#define ONE(L)                                                                     \
    {                                                                              \
      float t, a, b, c, e, u;                                                      \
      FMAS(t, m5, x2.L, x4.L); FMAI(r0.L, "4.0", x0.L, t);                         \
      FMAI(a, "-4.0", x2.L, x4.L); FMAI(b, "-4.0", x1.L, x3.L);                    \
      ADD(r1.L, a, b); SUB(r2.L, a, b);                                            \
      SUB(c, x4.L, x2.L); SUB(e, x3.L, x1.L);                                      \
      FMAI(r3.L, "2.0", e, c); FMAI(r4.L, "-2.0", e, c);                           \
      FMAS(u, m5, x3.L, x5.L); FMAI(r5.L, "4.0", x1.L, u);                         \
    }
    ONE(x) ONE(y)
#undef ONE
    x0 = r0; x1 = r1; x2 = r2; x3 = r3; x4 = r4; x5 = r5;
  }
}

// FLAGS: 1 no LDS-DMA   2 no transform arithmetic (reads kept)   4 no patch reads and no transform   8 no filter reads
template <int STAGGER, bool SCALAR, int FLAGS>
__global__ void __launch_bounds__(512, 2) f4loop(const float* __restrict__ src, float* __restrict__ out,
                                                 unsigned long long* __restrict__ stamps, int iters, float five) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: raw stages R0, R1 (32 KB each, de-duplicated images), filter stages U0, U1 (36 KB each) = 136 KB
  constexpr int RAW = 32768, UB = 36864;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (2 * RAW + 2 * UB) / 4; i += 512) ((float*)smem)[i] = src[i & 8191] * 0.01f;
  __syncthreads();
  f32x4 acc[36];
#pragma unroll
  for (int i = 0; i < 36; i++) acc[i] = (f32x4){0, 0, 0, 0};
  f32x2 v[36];
#pragma unroll
  for (int i = 0; i < 36; i++) v[i] = (f32x2){src[(tid + i) & 8191], src[(tid + 3 * i) & 8191]};
  const char* gsrc = (const char*)src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  const int rd = lane * 8;                       // conflict-free synthetic fragment address
  const bool late = STAGGER && w >= 4;           // waves 4-7: multiply first (V of the previous iteration), then transform
  auto transform = [&](int it) {                 // raw stage (it & 1) -> v[36]   (6 columns, then 6 rows, in place)
    if (FLAGS & 4) return;
    const char* rst = smem + (it & 1) * RAW + (w & 3) * 8192;
    f32x2 d[36];
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
      for (int i = 0; i < 6; i++) d[i * 6 + j] = *(const f32x2*)(rst + ((i * 6 + j) % 16) * 512 + rd);
    if (FLAGS & 2) {
#pragma unroll
      for (int i = 0; i < 36; i++) asm volatile("" :: "v"(d[i]));
      return;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) bt6<SCALAR>(d[0 * 6 + j], d[1 * 6 + j], d[2 * 6 + j], d[3 * 6 + j], d[4 * 6 + j], d[5 * 6 + j], five);
#pragma unroll
    for (int i = 0; i < 6; i++) bt6<SCALAR>(d[i * 6 + 0], d[i * 6 + 1], d[i * 6 + 2], d[i * 6 + 3], d[i * 6 + 4], d[i * 6 + 5], five);
#pragma unroll
    for (int i = 0; i < 36; i++) v[i] = d[i];
  };
  auto multiply = [&](int it) {
    const char* ust = smem + 2 * RAW + (it & 1) * UB + (w >> 2) * 18432;   // this wave's 16 out-channels: 36 x 512 B
    f32x2 b[36];
    if (!(FLAGS & 8)) {
      b[0] = *(const f32x2*)(ust + rd);
      b[1] = *(const f32x2*)(ust + 512 + rd);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 36; e++) {
      if (FLAGS & 8) b[e] = v[(e + 7) % 36];
      else if (e + 2 < 36) b[e + 2] = *(const f32x2*)(ust + (e + 2) * 512 + rd);
      // this wave's LDS-DMA pieces of the NEXT iteration's stages, one per step from step 4 on: 9 per wave
      if (!(FLAGS & 1) && e >= 4 && e < 13)
        DMA16(gsrc + ((it + e) & 31) * 1024, smem + ((it + 1) & 1) * (e < 8 ? RAW : UB) + (e < 8 ? 0 : 2 * RAW) + (((e - 4) * 8 + w) % 32) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      if (!(FLAGS & 8)) lgkm(e + 2 < 36 ? 1 : e == 34 ? 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      MFMA(acc[e], v[e].x, b[e].x);
      MFMA(acc[e], v[e].y, b[e].y);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // One code path for both halves: the sequence of every wave is  T M T M ...; waves 0-3 meet the iteration's
  // barrier before T, waves 4-7 (STAGGER) before M -- half an iteration out of phase.  (Two alternative copies of
  // the body under an if / else spill hundreds of registers.)  A real kernel would need a third raw stage for the
  // late waves' T; the instruction mix and the LDS / DMA traffic are the same.
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (!late) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    transform(it);
    if (late) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    multiply(it);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 36; i++) s += acc[i];
  out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3] + v[5].x;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------
// The shape a real kernel could take (f4loop2): POINT-SPLIT waves, tile-major raw stage.
//   wave (wt, ph) = 16 tiles x BOTH 16-out-channel blocks x 18 of the 36 points (point rows 3 ph .. 3 ph + 2)
//                 = 36 accumulator tiles as before, but no two waves repeat a transform: 30 patch reads (5 of the
//                   6 patch rows), B^T d for 3 rows x 6 columns (36 ops) + 3 rows of (.) B (36 ops) = 72 packed ops,
//                   36 filter reads (18 points x 2 blocks), 72 MFMAs (a V value feeds 4).
//   LDS: ONE raw stage [64 tiles][73 units of 16 B] = 73 KB (tile-major 6x6 patches: every patch read is an
//        immediate offset off one base register; 73 is odd, so 16 tiles hit 16 distinct bank groups) + TWO filter
//        stages of 36 KB.  The raw stage is single: T (read + transform) and M (multiply) are separated by a second
//        barrier per iteration, after which the LDS-DMA for the next iteration's patches may overwrite the stage.
//   per wave and iteration: 10 raw + 5 filter LDS-DMA pieces, issued one per step during M.
// FLAGS as above.
template <int FLAGS>
__global__ void __launch_bounds__(512, 2) f4loop2(const float* __restrict__ src, float* __restrict__ out,
                                                  unsigned long long* __restrict__ stamps, int iters, float five) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RAW = 64 * 73 * 16, UB = 36864;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (RAW + 2 * UB) / 4; i += 512) ((float*)smem)[i] = src[i & 8191] * 0.01f;
  __syncthreads();
  f32x4 acc[18][2];
#pragma unroll
  for (int i = 0; i < 18; i++) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
  f32x2 v[18];
#pragma unroll
  for (int i = 0; i < 18; i++) v[i] = (f32x2){src[(tid + i) & 8191], src[(tid + 3 * i) & 8191]};
  const char* gsrc = (const char*)src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  const int t16 = lane & 15, h = lane >> 4, wt = w >> 1, ph = w & 1;
  // patch pixel (r, c) of this lane's tile: base + (6 r + c) * 32, channel pair h (16-byte half h >> 1 swizzled by
  // the tile's bit 3 so that lanes 0-31 / 32-63 spread over all 64 banks)
  const char* rbase = smem + (wt * 16 + t16) * (73 * 16) + (((h >> 1) ^ ((t16 >> 3) & 1)) << 4) + ((h & 1) << 3) + ph * (6 * 32);
  const char* ubase0 = smem + RAW + ph * (18 * 1024) + t16 * 32 + ((h ^ (((t16 >> 3) & 1) << 1)) << 3);
  const char* ubase1 = ubase0 + 512;
  asm volatile("" : "+v"(ubase1));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- T: 30 patch reads (5 rows x 6 columns), the wave's 3 rows of B^T d, then (.) B per row
    if (!(FLAGS & 4)) {
      f32x2 d[5][6];
#pragma unroll
      for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) d[r][c] = *(const f32x2*)(rbase + (r * 6 + c) * 32);
      if (FLAGS & 2) {
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
          for (int c = 0; c < 6; c++) asm volatile("" :: "v"(d[r][c]));
      } else {
        const f32x2 m5 = {-five, -five};
        const f32x2 sg = {ph ? -1.f : 1.f, ph ? -1.f : 1.f};
        f32x2 tmp[3][6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          // rows (0,1,2) of B^T d for ph = 0 from patch rows 0..4; rows (5,4,3) for ph = 1 from rows 5..1 (read
          // here as 0..4 of the shifted base): the same three formulas with one sign (one code path)
          const f32x2 a = -4.f * d[2][c] + d[4][c], b = -4.f * d[1][c] + d[3][c];
          tmp[0][c] = 4.f * d[0][c] + (m5 * d[2][c] + d[4][c]);
          tmp[1][c] = a + sg * b;
          tmp[2][c] = a - sg * b;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
          f32x2 x0 = tmp[i][0], x1 = tmp[i][1], x2 = tmp[i][2], x3 = tmp[i][3], x4 = tmp[i][4], x5 = tmp[i][5];
          bt6<false>(x0, x1, x2, x3, x4, x5, five);
          v[i * 6 + 0] = x0; v[i * 6 + 1] = x1; v[i * 6 + 2] = x2; v[i * 6 + 3] = x3; v[i * 6 + 4] = x4; v[i * 6 + 5] = x5;
        }
      }
    }
    // ---- everyone has read the raw stage: the next iteration's patches may land in it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- M: 18 points x 2 column blocks x 2 channel steps = 72 MFMAs; filter fragments two steps ahead
    const int uoff = (it & 1) * UB;
    f32x2 b0[18], b1[18];
    if (!(FLAGS & 8)) {
      b0[0] = *(const f32x2*)(ubase0 + uoff); b1[0] = *(const f32x2*)(ubase1 + uoff);
      b0[1] = *(const f32x2*)(ubase0 + uoff + 1024); b1[1] = *(const f32x2*)(ubase1 + uoff + 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 18; e++) {
      if (FLAGS & 8) { b0[e] = v[(e + 7) % 18]; b1[e] = v[(e + 11) % 18]; }
      else if (e + 2 < 18) {
        b0[e + 2] = *(const f32x2*)(ubase0 + uoff + (e + 2) * 1024);
        b1[e + 2] = *(const f32x2*)(ubase1 + uoff + (e + 2) * 1024);
      }
      if (!(FLAGS & 1) && e >= 2 && e < 17)   // 15 pieces per wave: 10 raw, 5 filter
        DMA16(gsrc + ((it + e) & 31) * 1024, smem + (e < 12 ? ((e - 2) * 8 + w) % 73 * 1024 : RAW + ((it + 1) & 1) * UB + ((e - 12) * 8 + w) % 36 * 1024));
      __builtin_amdgcn_sched_barrier(0);
      if (!(FLAGS & 8)) lgkm(e + 2 < 18 ? 4 : e == 16 ? 2 : 0);   // the fragments of step e: all but the two younger pairs
      __builtin_amdgcn_sched_barrier(0);
      MFMA(acc[e][0], v[e].x, b0[e].x);
      MFMA(acc[e][1], v[e].x, b1[e].x);
      MFMA(acc[e][0], v[e].y, b0[e].y);
      MFMA(acc[e][1], v[e].y, b1[e].y);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  f32x4 s = acc[0][0];
#pragma unroll
  for (int i = 1; i < 18; i++) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3] + v[5].x;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int FLAGS>
void run2(const char* what, const float* in, float* out, unsigned long long* st, int iters) {
  CK(hipFuncSetAttribute((const void*)(f4loop2<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((f4loop2<FLAGS>), dim3(256), dim3(512), 163840, 0, in, out, st, iters, 5.0f);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto c : h) cyc += c;
  const double per_iter = cyc / 256 / iters, per_mfma = per_iter / 144.0;
  printf("%-78s %6.0f cycles / iteration = %5.1f per MFMA per SIMD (floor 32) = %5.1f F(2x2)-equivalent\n", what, per_iter,
         per_mfma, per_mfma / 1.361);
}

template <int STAGGER, bool SCALAR, int FLAGS>
void run(const char* what, const float* in, float* out, unsigned long long* st, int iters) {
  CK(hipFuncSetAttribute((const void*)(f4loop<STAGGER, SCALAR, FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((f4loop<STAGGER, SCALAR, FLAGS>), dim3(256), dim3(512), 163840, 0, in, out, st, iters, 5.0f);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), st, 256 * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto c : h) cyc += c;
  const double per_iter = cyc / 256 / iters, per_mfma = per_iter / 144.0;   // 2 waves x 72 MFMAs per SIMD and iteration
  printf("%-78s %6.0f cycles / iteration = %5.1f per MFMA per SIMD (floor 32) = %5.1f F(2x2)-equivalent\n", what, per_iter,
         per_mfma, per_mfma / 1.361);
}

int main() {
  float *in, *out; unsigned long long* st;
  CK(hipMalloc(&in, 64 * 65536 + 65536)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&st, 256 * 8));
  std::vector<float> h((64 * 65536 + 65536) / 4);
  for (auto& x : h) x = (float)rand() / (float)RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("F(4x4,3x3) fused loop, 8 waves x (16 tiles x 16 k x 36 points), per 8-channel iteration; the last column divides by 1.361\n"
         "(MFMAs per output, F(2x2) / F(4x4) at 14x14) and compares with the product F(2x2) loop's 38.5 cycles per MFMA (40.6 with epilogues)\n");
  for (int rep = 0; rep < 2; rep++) {
    run<0, false, 0>("lockstep, packed transform, full mix", in, out, st, 300);
    run<1, false, 0>("staggered halves, packed transform, full mix", in, out, st, 300);
    run<0, true, 0>("lockstep, scalar transform, full mix", in, out, st, 300);
    run<1, true, 0>("staggered halves, scalar transform, full mix", in, out, st, 300);
    run<1, true, 1>("staggered, scalar, no LDS-DMA", in, out, st, 300);
    run<1, true, 2>("staggered, no transform arithmetic (patch reads kept)", in, out, st, 300);
    run<1, true, 4>("staggered, no patch reads, no transform", in, out, st, 300);
    run<1, true, 4 | 1>("MFMAs + filter reads + barrier only", in, out, st, 300);
    run<1, true, 4 | 1 | 8>("MFMAs + barrier only", in, out, st, 300);
    run2<0>("POINT-SPLIT, tile-major single raw stage, two barriers: full mix", in, out, st, 300);
    run2<1>("point-split: no LDS-DMA", in, out, st, 300);
    run2<2>("point-split: no transform arithmetic (patch reads kept)", in, out, st, 300);
    run2<4>("point-split: no patch reads, no transform", in, out, st, 300);
    run2<4 | 1>("point-split: MFMAs + filter reads + barriers only", in, out, st, 300);
  }
  return 0;
}
