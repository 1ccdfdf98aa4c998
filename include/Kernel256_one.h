/*
 * Kernel256_one.h -- argument-less layer entry point(s) of the ./Test driver.
 * 1x1 conv 1024->256 + BN + ReLU / 256->1024 + BN (reference Kernel256_one.h:18-19, Kernel256_one.cu:59-242,277-449)
 *
 * Each call loads its .bin inputs (wino_data_files.h), runs the layer once on the GPU(s)
 * through the C-ABI of winograd_mi355x.h, runs the direct-conv comparator, prints the
 * reference's per-call lines and returns (mine_us << 16) | comparator_us, both clamped to
 * 0xFFFF.  Batch size / GPU count: wino_driver_set_batch / wino_driver_set_gpus.
 */
#ifndef WINO_KERNEL256_ONE_H
#define WINO_KERNEL256_ONE_H
#include "wino_data_files.h"
#ifdef __cplusplus
extern "C" {
#endif
int kernel_256_1_in(void);
int kernel_256_1_out(void);
#ifdef __cplusplus
}
#endif
#endif
