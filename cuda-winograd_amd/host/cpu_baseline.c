/*
 * cpu_baseline.c -- the host-core baseline ./Test prints beside its GPU numbers.
 *
 * A deliberately naive convolution of the SAME layer on the SAME inputs: explicit im2col of a
 * block of output pixels, then a plain three-loop SGEMM against the [9C][K] (or [Cin][Kout])
 * weight matrix, then the folded BN (+ReLU).  Worker threads (one per online core) pull blocks of
 * up to 28 output pixels from a shared counter.  This is a reported baseline (SURVEY.md section 8d,
 * BASELINE.md section 4): it is timed and diffed against the GPU output, it never produces a
 * result the library returns.  It plays the part cuDNN's GEMM algorithm plays in the reference's
 * tables (README.md:25), on the box's own host cores.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cpu_baseline.h"

enum { MAX_ROWS_PER_BLOCK = 28 };

typedef struct {
  int kind;                 /* 3 or 1 */
  int N, C, K, relu;
  const float *in, *wmat, *bias, *scale;
  float* out;
  long rows;                /* N * 196 output pixels */
  int rows_per_block;       /* 28, fewer when the layer has too few rows to give every thread some */
  atomic_long next_block;
} work_t;

/* gather the 9C (3x3) or C (1x1) inputs of output pixel `r` into `col` */
static void im2col_row(const work_t* w, long r, float* col) {
  const int C = w->C;
  if (w->kind == 1) {
    memcpy(col, w->in + r * C, (size_t)C * sizeof(float));
    return;
  }
  const long n = r / 196;
  const int p = (int)(r % 196), y = p / 14, x = p % 14;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx)
      memcpy(col + (size_t)(dy * 3 + dx) * C, w->in + (((n * 16 + y + dy) * 16) + x + dx) * C,
             (size_t)C * sizeof(float));
}

static void* worker(void* arg) {
  work_t* w = (work_t*)arg;
  const int C = w->C, K = w->K, depth = w->kind == 3 ? 9 * C : C;
  const int ROWS_PER_BLOCK = w->rows_per_block;
  float* cols = (float*)malloc((size_t)ROWS_PER_BLOCK * depth * sizeof(float));
  float* acc = (float*)malloc((size_t)ROWS_PER_BLOCK * K * sizeof(float));
  if (!cols || !acc) { free(cols); free(acc); return (void*)1; }
  const long nblocks = (w->rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  for (;;) {
    const long b = atomic_fetch_add(&w->next_block, 1);
    if (b >= nblocks) break;
    const long r0 = b * ROWS_PER_BLOCK;
    const int m = (int)(w->rows - r0 < ROWS_PER_BLOCK ? w->rows - r0 : ROWS_PER_BLOCK);
    for (int i = 0; i < m; ++i) im2col_row(w, r0 + i, cols + (size_t)i * depth);
    /* SGEMM: acc[m][K] = cols[m][depth] * wmat[depth][K], three plain loops */
    memset(acc, 0, (size_t)m * K * sizeof(float));
    for (int i = 0; i < m; ++i)
      for (int d = 0; d < depth; ++d) {
        const float a = cols[(size_t)i * depth + d];
        const float* wrow = w->wmat + (size_t)d * K;
        float* o = acc + (size_t)i * K;
        for (int k = 0; k < K; ++k) o[k] += a * wrow[k];
      }
    for (int i = 0; i < m; ++i) {
      const long r = r0 + i;
      float* o;
      if (w->kind == 3) {   /* padded output, interior only (the ring was zeroed up front) */
        const long n = r / 196;
        const int p = (int)(r % 196), y = p / 14, x = p % 14;
        o = w->out + (((n * 16 + y + 1) * 16) + x + 1) * K;
      } else {
        o = w->out + r * K;
      }
      for (int k = 0; k < K; ++k) {
        float v = w->scale[k] * acc[(size_t)i * K + k] + w->bias[k];
        o[k] = (w->relu && v < 0.f) ? 0.f : v;
      }
    }
  }
  free(cols);
  free(acc);
  return NULL;
}

int wino_host_cores(void) {
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) {
    const int n = CPU_COUNT(&set);
    if (n > 0) return n;
  }
  const long n = sysconf(_SC_NPROCESSORS_ONLN);
  return n > 0 ? (int)n : 1;
}

int wino_cpu_conv(int kind, const float* in, const float* w, const float* bias, const float* scale,
                  float* out, int N, int C, int K, int relu, int threads, int* threads_used) {
  if ((kind != 1 && kind != 3) || !in || !w || !bias || !scale || !out || N < 1 || C < 1 || K < 1) return -1;
  float* wmat = NULL;
  work_t wk;
  memset(&wk, 0, sizeof wk);
  if (kind == 3) {   /* [K][C][3][3] -> [(dy*3+dx)*C + c][K] */
    wmat = (float*)malloc((size_t)9 * C * K * sizeof(float));
    if (!wmat) return -1;
    for (int k = 0; k < K; ++k)
      for (int c = 0; c < C; ++c)
        for (int t = 0; t < 9; ++t) wmat[((size_t)t * C + c) * K + k] = w[((size_t)k * C + c) * 9 + t];
    memset(out, 0, (size_t)N * 256 * K * sizeof(float));
    wk.wmat = wmat;
  } else {
    wk.wmat = w;   /* already [Cin][Kout] */
  }
  wk.kind = kind; wk.N = N; wk.C = C; wk.K = K; wk.relu = relu;
  wk.in = in; wk.bias = bias; wk.scale = scale; wk.out = out;
  wk.rows = (long)N * 196;
  atomic_init(&wk.next_block, 0);
  if (threads < 1) threads = 1;
  if (threads > 1024) threads = 1024;
  {   /* blocks of 7..28 output pixels, about 4 per thread when the layer is large enough; never more
       * threads than blocks (N = 1: 28 threads of 7 pixels -- a thread per pixel costs more to start than to run) */
    long rpb = wk.rows / (4L * threads);
    wk.rows_per_block = rpb < 7 ? 7 : rpb > MAX_ROWS_PER_BLOCK ? MAX_ROWS_PER_BLOCK : (int)rpb;
    const long nblocks = (wk.rows + wk.rows_per_block - 1) / wk.rows_per_block;
    if (threads > nblocks) threads = (int)nblocks;
  }
  if (threads_used) *threads_used = threads;
  pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
  int started = 0, rc = 0;
  if (!th) { free(wmat); return -1; }
  for (int t = 1; t < threads; ++t)
    if (pthread_create(&th[started], NULL, worker, &wk) == 0) ++started;
  if (worker(&wk)) rc = -1;
  for (int t = 0; t < started; ++t) {
    void* r = NULL;
    pthread_join(th[t], &r);
    if (r) rc = -1;
  }
  free(th);
  free(wmat);
  return rc;
}
