/*
 * Test.c -- the ./Test <mode 0..5> command-line driver.
 *
 * Same protocol as the reference's Test.c:13-56: select device 0, run the chosen layer
 * entry point 100 times, print "---- Iter: i ----" before each call, discard the first
 * two results, average the two packed 16-bit timers over the remaining 98.
 *
 *   ./Test [mode] [N] [gpus] [iters]
 *     mode   0 kernel_128   1 kernel_256   2 kernel_128_1_in   3 kernel_128_1_out
 *            4 kernel_256_1_in   5 kernel_256_1_out            (default 0)
 *     N      batch size (default 1 = the reference)            [extension]
 *     gpus   devices to split N over, no collective (default 1) [extension]
 *     iters  number of calls (default 100)                      [extension]
 *
 * After the reference's "Average Total Time" line come a CPU-baseline line (the same layer as a
 * naive im2col + SGEMM on the box's host cores, core count stated, diffed against the GPU output;
 * BASELINE.md section 4) and a one-line JSON summary with the unclamped averages, effective
 * TFLOP/s on the algorithmic (direct-convolution) FLOPs, its fraction of the 157.3 TFLOP/s fp32
 * MFMA peak of one MI355X, and the error metrics.  WINO_STDOUT_COMPAT=1 prints the reference's
 * exact lines instead ("cuDNN" labels, nothing extra); WINO_CPU_BASELINE=0 skips the CPU pass.
 */
#include <stdio.h>
#include <stdlib.h>

#include "Kernel128_one.h"
#include "Kernel128_winograd.h"
#include "Kernel256_one.h"
#include "Kernel256_winograd.h"
#include "winograd_mi355x.h"

#define MI355X_FP32_MFMA_PEAK_TFLOPS 157.3

typedef int (*layer_fn)(void);
static const struct { const char* name; layer_fn fn; } LAYERS[6] = {
    {"kernel_128", kernel_128},           {"kernel_256", kernel_256},
    {"kernel_128_1_in", kernel_128_1_in}, {"kernel_128_1_out", kernel_128_1_out},
    {"kernel_256_1_in", kernel_256_1_in}, {"kernel_256_1_out", kernel_256_1_out}};

int main(int argc, char** argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  if (argc > 2) wino_driver_set_batch(atoi(argv[2]));
  if (argc > 3) wino_driver_set_gpus(atoi(argv[3]));
  int nTest = argc > 4 ? atoi(argv[4]) : 100;
  if (nTest < 3) nTest = 3;
  if (mode < 0 || mode > 5) mode = 0;

  if (wino_set_device(0) != WINO_OK) {
    printf("HIP failure: %s\n", wino_last_error_string());
    return EXIT_FAILURE;
  }

  long sum_mine = 0, sum_cmp = 0;
  double us_mine = 0, us_cmp = 0, us_steady = 0, worst_abs = 0, worst_rel = 0;
  long worst_cnt = 0;
  wino_driver_result r;
  for (int i = 0; i < nTest; ++i) {
    printf("---- Iter: %d ----\n", i);
    const int packed = LAYERS[mode].fn();
    wino_driver_last_result(&r);
    if (r.max_abs_err > worst_abs) worst_abs = r.max_abs_err;
    if (r.max_rel_err > worst_rel) worst_rel = r.max_rel_err;
    if (r.error_cnt > worst_cnt) worst_cnt = r.error_cnt;
    if (i > 1) { /* first two calls are warm-up (Test.c:45-48) */
      sum_mine += packed >> 16;
      sum_cmp += packed & 0xFFFF;
      us_mine += r.mine_us;
      us_cmp += r.comparator_us;
      us_steady += r.steady_us;
    }
  }
  const int counted = nTest - 2;
  const int compat = wino_driver_get_stdout_compat();
  printf(compat ? "Average Total Time: [Mine: %d us], [cuDNN: %d us]\n"
                : "Average Total Time: [Mine: %d us], [Direct: %d us]\n",
         (int)(sum_mine / counted), (int)(sum_cmp / counted));
  if (compat) return 0;   /* the reference prints nothing after this line (Test.c:50-55) */
  const char* cb_env = getenv("WINO_CPU_BASELINE");
  wino_cpu_baseline_result cb;
  const int have_cb = !(cb_env && cb_env[0] == '0') && wino_driver_cpu_baseline(&cb) == WINO_OK;
  if (have_cb)
    printf("CPU baseline (naive im2col+SGEMM+BN, %d host threads, %d reps): %.0f us, %.1f GFLOP/s; "
           "max |GPU - CPU| = %.3g (%.2g relative)\n",
           cb.threads, cb.reps, cb.us, cb.gflops, cb.max_abs_diff, cb.max_rel_diff);
  us_mine /= counted;
  us_cmp /= counted;
  us_steady /= counted;
  const double tflops = r.flops / (us_mine * 1e-6) / 1e12;
  const double tflops_steady = r.flops / (us_steady * 1e-6) / 1e12;
  printf("{\"layer\": \"%s\", \"N\": %d, \"gpus\": %d, \"iters\": %d, \"mine_us\": %.1f, "
         "\"comparator_us\": %.1f, \"effective_tflops\": %.3f, \"frac_of_fp32_mfma_peak\": %.4f, "
         "\"steady_us\": %.1f, \"steady_effective_tflops\": %.3f, \"steady_frac_of_fp32_mfma_peak\": %.4f, "
         "\"max_abs_err\": %.6g, \"max_rel_err\": %.3g, \"error_cnt_1e-5\": %ld, "
         "\"cpu_baseline_us\": %.1f, \"cpu_baseline_gflops\": %.2f, \"cpu_threads\": %d, "
         "\"gpu_vs_cpu_max_rel_diff\": %.3g}\n",
         LAYERS[mode].name, r.N, r.gpus, nTest, us_mine, us_cmp, tflops,
         tflops / (MI355X_FP32_MFMA_PEAK_TFLOPS * r.gpus), us_steady, tflops_steady,
         tflops_steady / (MI355X_FP32_MFMA_PEAK_TFLOPS * r.gpus), worst_abs, worst_rel, worst_cnt,
         have_cb ? cb.us : 0.0, have_cb ? cb.gflops : 0.0, have_cb ? cb.threads : 0,
         have_cb ? cb.max_rel_diff : 0.0);
  return 0;
}
