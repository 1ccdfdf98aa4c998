/*
 * Kernel128_winograd.h -- argument-less layer entry point(s) of the ./Test driver.
 * 3x3 conv 128->128 + BN + ReLU on data/input_14_1_128.bin (reference Kernel128_winograd.h:20, Kernel128_winograd.cu:215-434)
 *
 * Each call loads its .bin inputs (wino_data_files.h), runs the layer once on the GPU(s)
 * through the C-ABI of winograd_mi355x.h, runs the direct-conv comparator, prints the
 * reference's per-call lines and returns (mine_us << 16) | comparator_us, both clamped to
 * 0xFFFF.  Batch size / GPU count: wino_driver_set_batch / wino_driver_set_gpus.
 */
#ifndef WINO_KERNEL128_WINOGRAD_H
#define WINO_KERNEL128_WINOGRAD_H
#include "wino_data_files.h"
#ifdef __cplusplus
extern "C" {
#endif
int kernel_128(void);
#ifdef __cplusplus
}
#endif
#endif
