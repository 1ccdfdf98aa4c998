/*
 * Kernel128_winograd.h -- argument-less layer entry point(s) of the ./Test driver.
 * 3x3 conv 128->128 + BN + ReLU on data/input_14_1_128.bin (reference Kernel128_winograd.h:8-20, Kernel128_winograd.cu:215-434)
 *
 * Source-compatible with the reference header of the same name: the same entry point(s) and the
 * same file-name objects (inputName128 ... eVarName128, Kernel128_winograd.h:8-18), so host code written against the reference compiles
 * unchanged.  The objects have internal linkage here (the reference defines them with external
 * linkage in a header, which only links while a single C translation unit includes it).
 *
 * Each call loads its .bin inputs, runs the layer once on the GPU(s) through the C-ABI of
 * winograd_mi355x.h, runs the direct-conv comparator, prints the reference's per-call lines and
 * returns (mine_us << 16) | comparator_us (custom half clamped to 0x7FFF, comparator half to
 * 0xFFFF, so that `res >> 16` of Test.c:46 stays non-negative).  Batch size / GPU count:
 * wino_driver_set_batch / wino_driver_set_gpus.
 */
#ifndef WINO_KERNEL128_WINOGRAD_H
#define WINO_KERNEL128_WINOGRAD_H
#include "wino_data_files.h"
#ifdef __cplusplus
extern "C" {
#endif

static const char inputName128[] WINO_UNUSED = "data/input_14_1_128.bin";
static const char biasName128[] WINO_UNUSED = "data/bias_128.bin";
static const char weight_winograd_Name128[] WINO_UNUSED = "data/weight_winograd_128_128.bin";
static const char weight_NCHW_Name128[] WINO_UNUSED = "data/weight_NCHW_128_128.bin";
static const char bnBiasName128[] WINO_UNUSED = "data/bnBias_128.bin";
static const char bnScaleName128[] WINO_UNUSED = "data/bnScale_128.bin";
static const char bnBias_winograd_Name128[] WINO_UNUSED = "data/bnBias_winograd_128.bin";
static const char bnScale_winograd_Name128[] WINO_UNUSED = "data/bnScale_winograd_128.bin";
static const char eMeanName128[] WINO_UNUSED = "data/eMean_128.bin";
static const char eVarName128[] WINO_UNUSED = "data/eVar_128.bin";

int kernel_128(void);

#ifdef __cplusplus
}
#endif
#endif
