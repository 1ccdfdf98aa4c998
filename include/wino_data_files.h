/*
 * wino_data_files.h -- the on-disk data contract of the ./Test drivers.
 *
 * cwd-relative raw little-endian fp32 files written by data_generator.py; the names are
 * the ones the reference hard-codes in Kernel128_winograd.h:8-18, Kernel256_winograd.h:8-18,
 * Kernel128_one.h:8-16 and Kernel256_one.h:8-16 (layouts: SURVEY.md section 2.3).
 * `%d` is the channel count (128 or 256).
 */
#ifndef WINO_DATA_FILES_H
#define WINO_DATA_FILES_H

#define WINO_F_INPUT_3X3        "data/input_14_1_%d.bin"            /* [16][16][C]            */
#define WINO_F_INPUT_3X3_BATCH  "data/input_14_1_%d_N%d.bin"        /* [N][16][16][C] (new)   */
#define WINO_F_WEIGHT_WINOGRAD  "data/weight_winograd_%d_%d.bin"    /* [36][C][K] F(4x4,3x3)  */
#define WINO_F_WEIGHT_NCHW      "data/weight_NCHW_%d_%d.bin"        /* [K][C][3][3]           */
#define WINO_F_BN_BIAS_FOLDED   "data/bnBias_winograd_%d.bin"       /* beta - gamma*mu/sd     */
#define WINO_F_BN_SCALE_FOLDED  "data/bnScale_winograd_%d.bin"      /* gamma/sd               */

#define WINO_F_ONE_INPUT        "data/input_one_14_1024.bin"        /* [196][<=1024] prefix   */
#define WINO_F_ONE_INPUT_BATCH  "data/input_one_14_1024_N%d.bin"    /* [N*196*1024] (new)     */
#define WINO_F_ONE_WEIGHT       "data/weight_one_1024.bin"          /* [Cin][Kout] prefix     */
#define WINO_F_ONE_BN_BIAS      "data/bnBias_myKernel_one_1024.bin"
#define WINO_F_ONE_BN_SCALE     "data/bnScale_myKernel_one_1024.bin"

#endif
