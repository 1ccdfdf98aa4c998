/*
 * Kernel128_one.h -- argument-less layer entry point(s) of the ./Test driver.
 * 1x1 conv 512->128 + BN + ReLU / 128->512 + BN (reference Kernel128_one.h:8-19, Kernel128_one.cu:57-240,276-447)
 *
 * Source-compatible with the reference header of the same name: the same entry point(s) and the
 * same file-name objects (inputName128one ... eVarName128one, Kernel128_one.h:8-16), so host code written against the reference compiles
 * unchanged.  The objects have internal linkage here (the reference defines them with external
 * linkage in a header, which only links while a single C translation unit includes it).
 *
 * Each call loads its .bin inputs, runs the layer once on the GPU(s) through the C-ABI of
 * winograd_mi355x.h, runs the direct-conv comparator, prints the reference's per-call lines and
 * returns (mine_us << 16) | comparator_us (custom half clamped to 0x7FFF, comparator half to
 * 0xFFFF, so that `res >> 16` of Test.c:46 stays non-negative).  Batch size / GPU count:
 * wino_driver_set_batch / wino_driver_set_gpus.
 */
#ifndef WINO_KERNEL128_ONE_H
#define WINO_KERNEL128_ONE_H
#include "wino_data_files.h"
#ifdef __cplusplus
extern "C" {
#endif

static const char inputName128one[] WINO_UNUSED = "data/input_one_14_1024.bin";
static const char weightName128one[] WINO_UNUSED = "data/weight_one_1024.bin";
static const char bnBiasName128one[] WINO_UNUSED = "data/bnBias_one_1024.bin";
static const char bnScaleName128one[] WINO_UNUSED = "data/bnScale_one_1024.bin";
static const char bnBias_myKernel_Name128one[] WINO_UNUSED = "data/bnBias_myKernel_one_1024.bin";
static const char bnScale_myKernel_Name128one[] WINO_UNUSED = "data/bnScale_myKernel_one_1024.bin";
static const char eMeanName128one[] WINO_UNUSED = "data/eMean_one_1024.bin";
static const char eVarName128one[] WINO_UNUSED = "data/eVar_one_1024.bin";

int kernel_128_1_in(void);
int kernel_128_1_out(void);

#ifdef __cplusplus
}
#endif
#endif
