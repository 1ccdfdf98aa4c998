/* Host helpers of the ./Test driver -- see include/util.h for the contract and the
 * reference lines each function mirrors (util.c:5-63 of bssrdf/CUDA-Winograd). */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime under strict -std=c11 */
#include "util.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

uint64_t getTimeMicroseconds64(void) {
  struct timespec now;
  clock_gettime(CLOCK_REALTIME, &now);
  return (uint64_t)now.tv_sec * 1000000ull + (uint64_t)(now.tv_nsec / 1000);
}

float* get_parameter(const char* filename, int size) {
  const size_t want = (size_t)size * sizeof(float);
  float* buf = (float*)malloc(want ? want : sizeof(float));
  if (buf == NULL) {
    printf("Bad Malloc\n");
    exit(0);
  }
  FILE* f = fopen(filename, "rb");
  if (f == NULL) {
    printf("Bad file path: %s, %s\n", filename, strerror(errno));
    exit(0);
  }
  const size_t got = fread(buf, 1, want, f);
  fclose(f);
  if (got != want) {
    printf("Short file: %s holds %zu of %zu bytes\n", filename, got, want);
    exit(0);
  }
  return buf;
}

float* transpose(float* weight, int h, int w) {
  /* input is [w][h] row-major, result is [h][w] row-major */
  float* t = (float*)malloc((size_t)w * h * sizeof(float));
  if (t == NULL) {
    printf("Bad Malloc\n");
    exit(0);
  }
  for (int row = 0; row < h; ++row)
    for (int col = 0; col < w; ++col) t[(size_t)row * w + col] = weight[(size_t)col * h + row];
  free(weight);
  return t;
}

float output_checker_accumulate(const float* A, const float* B, int len, int channel, int shift,
                                float* max_error, long* error_cnt) {
  const int stride = len + 2 * shift;
  float big = 0.f;
  for (int y = 0; y < len; ++y) {
    const float* arow = A + ((size_t)(y + shift) * stride + shift) * channel;
    const float* brow = B + (size_t)y * len * channel;
    for (int i = 0; i < len * channel; ++i) {
      const float d = fabsf(arow[i] - brow[i]);
      if (d > 1e-5f) ++*error_cnt;
      if (d > *max_error || d != d) *max_error = d;
      const float m = fabsf(brow[i]);
      if (m > big) big = m;
    }
  }
  return big;
}

float output_checker(float* A, float* B, int len, int channel, int shift) {
  float max_error = 0.f;
  long error_cnt = 0;
  output_checker_accumulate(A, B, len, channel, shift, &max_error, &error_cnt);
  printf("[max_error: %f][error_cnt: %d]\n", max_error, (int)error_cnt);
  return max_error;
}
