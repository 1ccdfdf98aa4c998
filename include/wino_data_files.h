/*
 * wino_data_files.h -- the on-disk data contract of the ./Test drivers.
 *
 * cwd-relative raw little-endian fp32 files written by data_generator.py.  The reference hard-codes
 * their names as `const char[]` objects in Kernel128_winograd.h:8-18, Kernel256_winograd.h:8-18,
 * Kernel128_one.h:8-16 and Kernel256_one.h:8-16; the same-named headers here define the same objects
 * (layouts: SURVEY.md section 2.3).  The batched inputs are an extension.
 */
#ifndef WINO_DATA_FILES_H
#define WINO_DATA_FILES_H

/* the file-name objects have internal linkage and may go unused in a given translation unit */
#if defined(__GNUC__)
#define WINO_UNUSED __attribute__((unused))
#else
#define WINO_UNUSED
#endif

#define WINO_F_INPUT_3X3_BATCH  "data/input_14_1_%d_N%d.bin"        /* [N][16][16][C] (new)   */
#define WINO_F_ONE_INPUT_BATCH  "data/input_one_14_1024_N%d.bin"    /* [N*196*1024] (new)     */

#endif
