/*
 * Kernel256_winograd.h -- argument-less layer entry point(s) of the ./Test driver.
 * 3x3 conv 256->256 + BN + ReLU on data/input_14_1_256.bin (reference Kernel256_winograd.h:20, Kernel256_winograd.cu:220-429)
 *
 * Each call loads its .bin inputs (wino_data_files.h), runs the layer once on the GPU(s)
 * through the C-ABI of winograd_mi355x.h, runs the direct-conv comparator, prints the
 * reference's per-call lines and returns (mine_us << 16) | comparator_us, both clamped to
 * 0xFFFF.  Batch size / GPU count: wino_driver_set_batch / wino_driver_set_gpus.
 */
#ifndef WINO_KERNEL256_WINOGRAD_H
#define WINO_KERNEL256_WINOGRAD_H
#include "wino_data_files.h"
#ifdef __cplusplus
extern "C" {
#endif
int kernel_256(void);
#ifdef __cplusplus
}
#endif
#endif
