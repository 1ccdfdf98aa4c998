/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py header).
 *
 * Plain-C restatement of the reference's conv(+BN+ReLU) path, used
 *   (1) by tests/ as an independent checker next to the numpy oracle, and
 *   (2) by bench.py's `cpu_baseline` leg: a naive im2col + triple-loop SGEMM
 *       (+ folded BN + ReLU) of the same layer, threaded over output rows with
 *       pthreads.
 * The product library never links this file.
 *
 * What each function follows in the reference:
 *   oracle_conv3x3_im2col   the layer cuDNN computes in the reference's comparator half:
 *                           cross-correlation, stride 1, no pad on a 16x16 image
 *                           (Kernel128_winograd.cu:352), BN-inference folded to
 *                           scale/bias (data_generator.py:41-46), ReLU
 *                           (Kernel128_winograd.cu:399); output written padded with
 *                           shift 1 like the custom path (Kernel128_winograd.cu:163).
 *   oracle_conv1x1          kernel_512_one_128 / kernel_128_one_512 /
 *                           kernel_1024_one_256 / kernel_256_one_1024
 *                           (Kernel128_one.cu:24-54,244-273; Kernel256_one.cu:26-56,246-274):
 *                           C = scale*(A.B)+bias, ReLU only on the reducing layers.
 *   oracle_winograd_f4      the three reference launches BtdB -> OuterProduct -> AtIA
 *                           (Kernel128_winograd.cu:28-213) on [36][C][K] weights.
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime, pthreads under strict -std=c11 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define HW 16
#define PQ 14

double oracle_now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

/* ------------------------------------------------------------------ */
/* generic row-parallel driver                                         */
/* ------------------------------------------------------------------ */
typedef void (*row_fn)(void* ctx, long r0, long r1);
typedef struct { row_fn fn; void* ctx; long r0, r1; } job_t;
static void* job_main(void* p) { job_t* j = (job_t*)p; j->fn(j->ctx, j->r0, j->r1); return 0; }

static void parallel_rows(row_fn fn, void* ctx, long rows, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if ((long)nthreads > rows) nthreads = (int)rows;
  pthread_t th[256];
  job_t jobs[256];
  long per = (rows + nthreads - 1) / nthreads;
  int i, started = 0;
  for (i = 0; i < nthreads; i++) {
    long r0 = i * per, r1 = r0 + per > rows ? rows : r0 + per;
    if (r0 >= r1) break;
    jobs[i].fn = fn; jobs[i].ctx = ctx; jobs[i].r0 = r0; jobs[i].r1 = r1;
    if (i == nthreads - 1 || r1 == rows) { fn(ctx, r0, r1); break; }
    pthread_create(&th[i], 0, job_main, &jobs[i]);
    started++;
  }
  for (i = 0; i < started; i++) pthread_join(th[i], 0);
}

/* ------------------------------------------------------------------ */
/* 3x3: naive im2col + triple loop                                     */
/* ------------------------------------------------------------------ */
typedef struct {
  const float *in, *wmat, *scale, *bias;
  float* out;
  int N, C, K, relu;
} c3_t;

/* rows = N*196 output pixels. For each row build the 9C-long im2col vector and
 * run the plain i-k-j triple loop against wmat[9C][K]. */
static void c3_rows(void* p, long r0, long r1) {
  c3_t* a = (c3_t*)p;
  const int C = a->C, K = a->K;
  float* col = (float*)malloc(sizeof(float) * 9 * C);
  float* acc = (float*)malloc(sizeof(float) * K);
  long r;
  for (r = r0; r < r1; r++) {
    int n = (int)(r / (PQ * PQ)), pq = (int)(r % (PQ * PQ)), y = pq / PQ, x = pq % PQ;
    int rr, ss, c, k, j;
    for (rr = 0; rr < 3; rr++)
      for (ss = 0; ss < 3; ss++)
        memcpy(col + (rr * 3 + ss) * C,
               a->in + (((long)n * HW + y + rr) * HW + x + ss) * C, sizeof(float) * C);
    for (k = 0; k < K; k++) acc[k] = 0.f;
    for (j = 0; j < 9 * C; j++) {
      const float v = col[j];
      const float* wr = a->wmat + (long)j * K;
      for (k = 0; k < K; k++) acc[k] += v * wr[k];
    }
    float* o = a->out + (((long)n * HW + y + 1) * HW + x + 1) * K;
    for (k = 0; k < K; k++) {
      float v = a->scale[k] * acc[k] + a->bias[k];
      o[k] = (a->relu && v < 0.f) ? 0.f : v;
    }
    (void)c;
  }
  free(col); free(acc);
}

/* in [N][16][16][C], w [K][C][3][3], out [N][16][16][K] (ring zeroed here). */
int oracle_conv3x3_im2col(const float* in, const float* w_kcrs, const float* scale,
                          const float* bias, float* out, int N, int C, int K, int relu,
                          int nthreads) {
  float* wmat = (float*)malloc(sizeof(float) * 9 * (size_t)C * K);
  int k, c, r, s;
  if (!wmat) return -1;
  for (k = 0; k < K; k++)
    for (c = 0; c < C; c++)
      for (r = 0; r < 3; r++)
        for (s = 0; s < 3; s++)
          wmat[((size_t)(r * 3 + s) * C + c) * K + k] = w_kcrs[(((size_t)k * C + c) * 3 + r) * 3 + s];
  memset(out, 0, sizeof(float) * (size_t)N * HW * HW * K);
  c3_t a = {in, wmat, scale, bias, out, N, C, K, relu};
  parallel_rows(c3_rows, &a, (long)N * PQ * PQ, nthreads);
  free(wmat);
  return 0;
}

/* ------------------------------------------------------------------ */
/* 1x1                                                                 */
/* ------------------------------------------------------------------ */
typedef struct {
  const float *A, *B, *scale, *bias;
  float* C;
  int Cin, Kout, relu;
} c1_t;

static void c1_rows(void* p, long r0, long r1) {
  c1_t* a = (c1_t*)p;
  const int Cin = a->Cin, K = a->Kout;
  float* acc = (float*)malloc(sizeof(float) * K);
  long r;
  int j, k;
  for (r = r0; r < r1; r++) {
    const float* ar = a->A + r * Cin;
    for (k = 0; k < K; k++) acc[k] = 0.f;
    for (j = 0; j < Cin; j++) {
      const float v = ar[j];
      const float* br = a->B + (long)j * K;
      for (k = 0; k < K; k++) acc[k] += v * br[k];
    }
    float* o = a->C + r * K;
    for (k = 0; k < K; k++) {
      float v = a->scale[k] * acc[k] + a->bias[k];
      o[k] = (a->relu && v < 0.f) ? 0.f : v;
    }
  }
  free(acc);
}

/* argument order (A, B, bnBias, bnScale, C) as the reference kernels take it */
int oracle_conv1x1(const float* A, const float* B, const float* bnBias, const float* bnScale,
                   float* Cout, long M, int Cin, int Kout, int relu, int nthreads) {
  c1_t a = {A, B, bnScale, bnBias, Cout, Cin, Kout, relu};
  parallel_rows(c1_rows, &a, M, nthreads);
  return 0;
}

/* ------------------------------------------------------------------ */
/* F(4x4,3x3): the reference's three stages, one image at a time        */
/* ------------------------------------------------------------------ */
static const float BT4[6][6] = {{4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
static const float AT4[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};

/* in [N][16][16][C]; U36 [36][C][K] (weight_winograd_C_K.bin); out [N][16][16][K] */
int oracle_winograd_f4(const float* in, const float* U36, const float* scale, const float* bias,
                       float* out, int N, int C, int K) {
  float* V = (float*)malloc(sizeof(float) * 36 * 16 * (size_t)C);   /* [36][16 tiles][C] */
  float* M = (float*)malloc(sizeof(float) * 36 * 16 * (size_t)K);   /* [36][16 tiles][K] */
  int n, tx, ty, c, k, i, j, e, t;
  if (!V || !M) return -1;
  memset(out, 0, sizeof(float) * (size_t)N * HW * HW * K);
  for (n = 0; n < N; n++) {
    const float* img = in + (size_t)n * HW * HW * C;
    /* stage 1: V = B^T d B  (Kernel128_winograd.cu:28-120); reads beyond the image = 0 */
    for (tx = 0; tx < 4; tx++) for (ty = 0; ty < 4; ty++) for (c = 0; c < C; c++) {
      float d[6][6], btd[6][6];
      for (i = 0; i < 6; i++) for (j = 0; j < 6; j++) {
        int y = 4 * tx + i, x = 4 * ty + j;
        d[i][j] = (y < HW && x < HW) ? img[((size_t)y * HW + x) * C + c] : 0.f;
      }
      for (i = 0; i < 6; i++) for (j = 0; j < 6; j++) {
        float s = 0.f; int q;
        for (q = 0; q < 6; q++) s += BT4[i][q] * d[q][j];
        btd[i][j] = s;
      }
      for (i = 0; i < 6; i++) for (j = 0; j < 6; j++) {
        float s = 0.f; int q;
        for (q = 0; q < 6; q++) s += btd[i][q] * BT4[j][q];
        V[((size_t)(i * 6 + j) * 16 + tx * 4 + ty) * C + c] = s;
      }
    }
    /* stage 2: M_e = V_e U_e  (Kernel128_winograd.cu:186-213) */
    memset(M, 0, sizeof(float) * 36 * 16 * (size_t)K);
    for (e = 0; e < 36; e++) for (t = 0; t < 16; t++) {
      float* m = M + ((size_t)e * 16 + t) * K;
      for (c = 0; c < C; c++) {
        const float v = V[((size_t)e * 16 + t) * C + c];
        const float* u = U36 + ((size_t)e * C + c) * K;
        for (k = 0; k < K; k++) m[k] += v * u[k];
      }
    }
    /* stage 3: Y = A^T M A, BN, ReLU, clip (Kernel128_winograd.cu:123-183) */
    for (tx = 0; tx < 4; tx++) for (ty = 0; ty < 4; ty++) for (k = 0; k < K; k++) {
      float m[6][6], atm[4][6];
      for (i = 0; i < 6; i++) for (j = 0; j < 6; j++)
        m[i][j] = M[((size_t)(i * 6 + j) * 16 + tx * 4 + ty) * K + k];
      for (i = 0; i < 4; i++) for (j = 0; j < 6; j++) {
        float s = 0.f; int q;
        for (q = 0; q < 6; q++) s += AT4[i][q] * m[q][j];
        atm[i][j] = s;
      }
      for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) {
        float s = 0.f, o; int q, y = 4 * tx + 1 + i, x = 4 * ty + 1 + j;
        if (y > PQ || x > PQ) continue;
        for (q = 0; q < 6; q++) s += atm[i][q] * AT4[j][q];
        o = scale[k] * s + bias[k];
        out[(((size_t)n * HW + y) * HW + x) * K + k] = o > 0.f ? o : 0.f;
      }
    }
  }
  free(V); free(M);
  return 0;
}
