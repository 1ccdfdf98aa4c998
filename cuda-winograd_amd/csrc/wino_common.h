// Internal helpers shared by the HIP translation units of libwinograd_mi355x.so.
// gfx950 (CDNA4, wave64) only.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "winograd_mi355x.h"

namespace wino {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Thread-local last-error text behind wino_last_error_string().
void set_error(const char* fmt, ...);
// Every tensor pointer of the C-ABI must be 16-byte aligned (the kernels move 16 bytes per lane; hipMalloc gives 256):
// true if any of the given pointers is not.  BN vectors are read four bytes at a time and need no more than that.
inline bool misaligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15u) != 0;
}
int hip_fail(hipError_t e, const char* what);
// Stream-K scratch (wino_runtime.hip): write-through slabs for partial segments and ticket
// counters, owned by the library, one set per (device, stream) so that launches on different
// streams never share it; grown on demand (a synchronous hipMalloc: call wino_conv3x3_prepare /
// wino_conv1x1_prepare first when the launch is going to be captured into a graph).  Launches on
// one stream run one after the other, so the 3x3 and the 1x1 kernels share a stream's set.
// Counters are zero at allocation and returned to zero by every launch's last arrivers.
// `err` is one host-visible word per (device, stream) that a kernel sets when it draws a ticket on a counter
// that cannot have been zero when the launch began (a launch that died mid-way before it): sk_scratch() then
// fails with WINO_E_STATE until wino_stream_reset_scratch() has zeroed the counters again -- the reference
// holds no state between calls and fails fast (Kernel128_winograd.cu:16-22,236-256); this is the same
// contract for the one piece of state the library keeps.
struct SkBufs {
  float* slabs;
  unsigned* tickets;
  unsigned* err;
};
int sk_scratch(int dev, hipStream_t s, size_t slab_bytes, size_t n_tickets, SkBufs* out);
int sk_scratch_release(hipStream_t s);   // wino_stream_destroy: the stream's scratch, on every device
// A launch that had been handed the scratch failed: the stream's counters can no longer be trusted.
void sk_mark_failed(int dev, hipStream_t s);
int device_cus(int dev, int* cus);
// Developer knobs (WINO_* environment variables), read once per process at first use and cached;
// wino_debug_reload_knobs() re-reads them (tests sweep the launch decompositions that way).
struct Knobs {
  int sk_grid;        // WINO_SK_GRID: logical workgroups of the 3x3 throughput kernel (0 = cost model)
  int sk_min_iters;   // WINO_SK_MIN_ITERS: shortest stream-K range of the 3x3 kernel (0 = default)
  int algo_3x3;       // WINO_3X3_ALGO: 0 automatic, 1 "big" (throughput kernel), 2 "small" (latency kernel)
  int sk_1x1;         // WINO_1X1_SK: -1 automatic, 0 plain form, 1 stream-K whenever a legal grid exists
  int sk_1x1_grid;    // WINO_1X1_SK_GRID: number of ranges (0 = model)
  int sk_kp;          // WINO_SK_KP: 1 (default) the 3x3 stream-K tail per k-block, ranges placed in phase order; 0 round 2's
                      // item-major list in launch order; 2 / 3 only the groups / only the phase order (A/B measurements)
  int small_split;    // WINO_SMALL_SPLIT: C-split S of the 3x3 latency kernel (0 = policy)
  int small3_ct;      // WINO_SMALL_CT: MFMA tiles per wave (block width / 16) of the 3x3 latency kernel, 1 / 2 / 4 (0 = policy)
  int algo_1x1;       // WINO_1X1_ALGO: 0 automatic, 1 "big" (LDS-staged kernel), 2 "small" (latency kernel)
  int small_ks;       // WINO_1X1_SMALL_KS: K-split of the 1x1 latency kernel, 1 / 2 / 4 (0 = policy)
  int small_rt, small_ct;   // WINO_1X1_SMALL_RT / _CT: MFMA row / column tiles per wave, 1 / 2 (0 = policy)
};
Knobs knobs();
#define WINO_HIP(call)                                          \
  do {                                                          \
    hipError_t e_ = (call);                                     \
    if (e_ != hipSuccess) return ::wino::hip_fail(e_, #call);   \
  } while (0)

// Launch-error check that does not synchronise (safe inside graph capture).
static inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, what);
  return WINO_OK;
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4): each lane fetches 16 B from its
// own `src`; the wave's 64 pieces land at `lds_wave_base + lane*16` (wave-uniform base).
__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)src,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same 16-byte LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds):
// per-lane 32-bit byte offset `voff` + wave-uniform byte offset `soff` (an SGPR), so that walking
// a loop-uniform stride costs no VALU address arithmetic at all.  The buffer must be < 4 GiB.
#if defined(__HIP_DEVICE_COMPILE__)
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)0, (short)0, 0, 0)) buffer_rsrc_t;
__device__ __forceinline__ buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma16_buf(buffer_rsrc_t rsrc, unsigned voff, unsigned soff,
                                          void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base,
                                           16, voff, soff, 0, 0);
}
// 16-byte write-through (sc1) store / L1-bypassing (sc1) load through a buffer descriptor: the
// pair used for data handed from one workgroup to another inside a launch (per-XCD L2s are not
// coherent with each other; sc1 stores leave the L2, sc1 loads never hit a stale L1 line).
typedef unsigned u32x4_vs __attribute__((__vector_size__(16)));
__device__ __forceinline__ void slab_store16(f32x4 v, buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_vs, v), rsrc, voff, soff, 16);
}
__device__ __forceinline__ void buf_store16(f32x4 v, buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_vs, v), rsrc, voff, soff, 0);
}
// the same, non-temporal (nt): write-once data that should not displace what others re-read from L2
__device__ __forceinline__ void buf_store16_nt(f32x4 v, buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_vs, v), rsrc, voff, soff, 2);
}
__device__ __forceinline__ f32x4 slab_load16(buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16));
}
#else  // host pass: the kernels are only parsed, these builtins do not exist there
typedef int buffer_rsrc_t;
__device__ inline buffer_rsrc_t make_rsrc(const void*, unsigned) { return 0; }
__device__ inline void dma16_buf(buffer_rsrc_t, unsigned, unsigned, void*) {}
__device__ inline void slab_store16(f32x4, buffer_rsrc_t, unsigned, unsigned) {}
__device__ inline void buf_store16(f32x4, buffer_rsrc_t, unsigned, unsigned) {}
__device__ inline void buf_store16_nt(f32x4, buffer_rsrc_t, unsigned, unsigned) {}
__device__ inline f32x4 slab_load16(buffer_rsrc_t, unsigned, unsigned) { return f32x4{0.f, 0.f, 0.f, 0.f}; }
#endif

// Division of a 32-bit unsigned by a launch-invariant divisor (Granlund-Montgomery): the
// multiplier travels in the kernel arguments, q = (t + ((n - t) >> 1)) >> (l - 1), t = umulhi(m, n).
struct FastDiv {
  unsigned m, l;   // l = 0: divisor 1
};
__host__ __device__ inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f = {0u, 0u};
  if (d <= 1) return f;
  unsigned l = 0;
  while ((1ull << l) < d) l++;
  f.l = l;
  f.m = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1ull);
  return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, FastDiv f) {
  if (f.l == 0) return n;
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> 1)) >> (f.l - 1);
}

__device__ __forceinline__ void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace wino
