/* cpu_baseline.h -- internal to the host driver (layer_driver.c <-> cpu_baseline.c). */
#ifndef WINO_CPU_BASELINE_H
#define WINO_CPU_BASELINE_H

/* cores this process may run on (sched_getaffinity, else _SC_NPROCESSORS_ONLN) */
__attribute__((visibility("hidden"))) int wino_host_cores(void);

/* kind 3: in [N][16][16][C], w [K][C][3][3], out [N][16][16][K] (ring 0), valid 3x3 conv + BN (+ReLU)
 * kind 1: in [N*196][C],     w [C][K],       out [N*196][K],               GEMM + BN (+ReLU)
 * naive im2col + three-loop SGEMM on up to `threads` host threads (fewer when the layer has fewer blocks
 * of output pixels); 0 on success; *threads_used (may be NULL) receives the number actually started */
__attribute__((visibility("hidden"))) int wino_cpu_conv(int kind, const float* in, const float* w, const float* bias, const float* scale,
                  float* out, int N, int C, int K, int relu, int threads, int* threads_used);

#endif
