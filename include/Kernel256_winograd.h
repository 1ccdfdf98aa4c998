/*
 * Kernel256_winograd.h -- argument-less layer entry point(s) of the ./Test driver.
 * 3x3 conv 256->256 + BN + ReLU on data/input_14_1_256.bin (reference Kernel256_winograd.h:8-20, Kernel256_winograd.cu:220-429)
 *
 * Source-compatible with the reference header of the same name: the same entry point(s) and the
 * same file-name objects (inputName256 ... eVarName256, Kernel256_winograd.h:8-18), so host code written against the reference compiles
 * unchanged.  The objects have internal linkage here (the reference defines them with external
 * linkage in a header, which only links while a single C translation unit includes it).
 *
 * Each call loads its .bin inputs, runs the layer once on the GPU(s) through the C-ABI of
 * winograd_mi355x.h, runs the direct-conv comparator, prints the reference's per-call lines and
 * returns (mine_us << 16) | comparator_us (custom half clamped to 0x7FFF, comparator half to
 * 0xFFFF, so that `res >> 16` of Test.c:46 stays non-negative).  Batch size / GPU count:
 * wino_driver_set_batch / wino_driver_set_gpus.
 */
#ifndef WINO_KERNEL256_WINOGRAD_H
#define WINO_KERNEL256_WINOGRAD_H
#include "wino_data_files.h"
#ifdef __cplusplus
extern "C" {
#endif

static const char inputName256[] WINO_UNUSED = "data/input_14_1_256.bin";
static const char biasName256[] WINO_UNUSED = "data/bias_256.bin";
static const char weight_winograd_Name256[] WINO_UNUSED = "data/weight_winograd_256_256.bin";
static const char weight_NCHW_Name256[] WINO_UNUSED = "data/weight_NCHW_256_256.bin";
static const char bnBiasName256[] WINO_UNUSED = "data/bnBias_256.bin";
static const char bnScaleName256[] WINO_UNUSED = "data/bnScale_256.bin";
static const char bnBias_winograd_Name256[] WINO_UNUSED = "data/bnBias_winograd_256.bin";
static const char bnScale_winograd_Name256[] WINO_UNUSED = "data/bnScale_winograd_256.bin";
static const char eMeanName256[] WINO_UNUSED = "data/eMean_256.bin";
static const char eVarName256[] WINO_UNUSED = "data/eVar_256.bin";

int kernel_256(void);

#ifdef __cplusplus
}
#endif
#endif
