/*
 * util.h -- host helpers of the ./Test driver (C ABI, exported by libwinograd_mi355x.so).
 *
 * Same four entry points, argument meaning and error behaviour as the reference's
 * util.h:19-25 / util.c:5-63, re-implemented from scratch:
 *   get_parameter          malloc + read `size` little-endian fp32 values from `filename`;
 *                          on a missing file or failed malloc prints a message and exit(0)
 *                          like the reference (util.c:28-44); additionally a SHORT file is
 *                          an error here (the reference ignores fread's result).
 *   transpose              out[j*w + i] = in[i*h + j] for i<w, j<h; frees its input
 *                          (util.c:15-26): turns [w][h] row-major into [h][w].
 *   getTimeMicroseconds64  wall clock in microseconds (util.c:5-13).
 *   output_checker         max |A-B| and count of |A-B| > 1e-5 over a len x len x channel
 *                          tensor, A optionally padded by `shift` pixels per side
 *                          (util.c:46-63); prints "[max_error: %f][error_cnt: %d]" and
 *                          -- unlike the reference, which falls off the end of a non-void
 *                          function -- returns max_error.
 */
#ifndef WINO_UTIL_H
#define WINO_UTIL_H

/* The reference's util.h:9-19 pulls these in for every translation unit that includes it; its own
 * util.c:1-3 gets printf / malloc / exit / uint64_t from nowhere else, so a reference-side unit may
 * rely on util.h alone for them.  Same set here, minus the two x86 intrinsics headers (<immintrin.h>,
 * <xmmintrin.h>), which no caller uses and a non-x86 host does not have. */
#include <assert.h>
#include <errno.h>
#include <float.h>
#include <inttypes.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

float* get_parameter(const char* filename, int size);
float* transpose(float* weight, int h, int w);
uint64_t getTimeMicroseconds64(void);
float output_checker(float* A, float* B, int len, int channel, int shift);

/* extension used by the batched drivers: same comparison, silent, accumulating into
 * caller-owned statistics; returns the largest |B| seen (for the relative metric) */
float output_checker_accumulate(const float* A, const float* B, int len, int channel, int shift,
                                float* max_error, long* error_cnt);

#ifdef __cplusplus
}
#endif
#endif
