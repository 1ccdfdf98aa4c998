// Internal helpers shared by the HIP translation units of libwinograd_mi355x.so.
// gfx950 (CDNA4, wave64) only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "winograd_mi355x.h"

namespace wino {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Thread-local last-error text behind wino_last_error_string().
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

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

__device__ __forceinline__ void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace wino
