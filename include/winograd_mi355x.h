/*
 * winograd_mi355x.h -- C-ABI of libwinograd_mi355x.so
 *
 * MI355X (gfx950) native fused Winograd F(2x2,3x3) conv + BN + ReLU and
 * 1x1-conv GEMM + BN (+ReLU).  Plain C: pointers, sizes, ints.  No HIP or
 * torch types; a stream is passed as an opaque `void*` (a hipStream_t, NULL =
 * the default stream).  All `float*` tensor arguments are DEVICE pointers
 * unless a name ends in `_host`.
 *
 * Every function returns WINO_OK (0) or a negative WINO_E_* code;
 * wino_last_error_string() describes the last failure of the calling thread.
 * Nothing here falls back to the CPU: without a usable GPU every compute entry
 * point fails with WINO_E_HIP.
 *
 * Reference interface each group replaces (paths into bssrdf/CUDA-Winograd):
 *   runtime plumbing   cudaSetDevice (Test.c:15), cudaMalloc/cudaMemset/cudaMemcpy/
 *                      cudaFree/cudaDeviceSynchronize/cudaGetErrorName
 *                      (Kernel128_winograd.cu:236-286)
 *   filter transforms  data_generator.py:63-78 (offline G g G^T)
 *   wino_conv3x3_*     the three launches kernel_{128,256}_winograd_BtdB ->
 *                      kernel_*_OuterProduct_* -> kernel_*_winograd_AtIA
 *                      (Kernel128_winograd.cu:263-265, Kernel256_winograd.cu:266-268)
 *   wino_conv1x1_bn    kernel_512_one_128 / kernel_128_one_512 (Kernel128_one.cu:98,316),
 *                      kernel_1024_one_256 / kernel_256_one_1024 (Kernel256_one.cu:100,318)
 *   wino_conv3x3_direct  the comparator role cuDNN plays in the reference
 *                      (Kernel128_winograd.cu:382-404), as an independent
 *                      non-Winograd GPU kernel
 * The six argument-less reference entry points themselves are declared in
 * Kernel128_winograd.h, Kernel256_winograd.h, Kernel128_one.h, Kernel256_one.h
 * (same directory) and exported by the same library.
 */
#ifndef WINOGRAD_MI355X_H
#define WINOGRAD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped when an existing entry point changes meaning or signature; additions leave it alone.
 * Added since the first cut of version 1 (all additive): wino_conv3x3_prepare(_hw),
 * wino_conv3x3_bn_relu_hw, wino_conv3x3_direct_hw, wino_conv3x3_plan, wino_conv3x3_f4_*,
 * wino_conv1x1_prepare, wino_conv1x1_plan, wino_conv1x1_bn_ex_hw, wino_residual_block_hw,
 * wino_residual_block_workspace_bytes_hw; wino_stream_destroy now also releases the stream's
 * scratch; wino_conv3x3_bn_relu(_hw) take any batch (they used to reject tensors of 4 GiB);
 * wino_last_status_name, wino_debug_reload_knobs, wino_residual_block_prepare(_hw),
 * wino_driver_set_gpu_alias, wino_driver_set_stdout_compat, wino_driver_cpu_baseline,
 * wino_diag_conv3x3_clock, wino_debug_tickets_in_use, wino_stream_check, wino_stream_reset_scratch,
 * wino_debug_poison_ticket, wino_diag_last_clock, wino_conv3x3_small_plan, wino_conv1x1_small_plan,
 * wino_conv3x3_plan_groups, wino_conv1x1_small_plan2, wino_conv3x3_small_plan2, wino_debug_conv1x1_models,
 * WINO_E_STATE.  The library-owned stream-K scratch is never freed or moved while its
 * stream lives (it used to be reallocated when a larger shape arrived). */
#define WINO_ABI_VERSION 1

enum {
  WINO_OK = 0,
  WINO_E_HIP = -1,       /* a HIP runtime call failed (no device, OOM, launch error) */
  WINO_E_SHAPE = -2,     /* unsupported / inconsistent shape argument */
  WINO_E_ARG = -3,       /* NULL pointer, bad enum, workspace too small, a tensor pointer that is not 16-byte aligned
                            (the kernels move 16 bytes per lane; wino_malloc / hipMalloc give 256; BN vectors need 4) */
  WINO_E_STATE = -4,     /* the stream's library-owned scratch cannot be trusted (an earlier launch on it
                            failed or was aborted): wino_stream_reset_scratch() recovers */
};

/* geometry fixed by the reference's 14x14 stage */
#define WINO_HW 16        /* padded input / output extent */
#define WINO_PQ 14        /* valid output extent          */
#define WINO_TILES 49     /* F(2x2) tiles per image (7x7) */

typedef void* wino_stream_t;

/* ---- runtime plumbing (thin wrappers so that C hosts need no HIP headers) ---- */
int wino_abi_version(void);
const char* wino_last_error_string(void);
/* hipGetErrorName() of the status the calling thread's most recent memcpy / synchronise wrapper got
 * ("hipSuccess" when it worked): the line the reference prints after each copy-back,
 * cudaGetErrorName(cudaMemcpy(...)) (Kernel128_winograd.cu:274-275,408-409). */
const char* wino_last_status_name(void);
int wino_device_count(int* count);
int wino_set_device(int device);
int wino_device_name(int device, char* buf, size_t buflen);
int wino_malloc(void** dptr, size_t bytes);
int wino_free(void* dptr);
int wino_memset(void* dptr, int value, size_t bytes);
int wino_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int wino_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int wino_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes);
int wino_device_synchronize(void);
int wino_stream_create(wino_stream_t* stream);
int wino_stream_destroy(wino_stream_t stream);   /* waits for the stream, frees the library's scratch of it */
int wino_stream_synchronize(wino_stream_t stream);
/* Fail-fast contract for the one piece of state the library keeps (the reference keeps none: every call
 * re-allocates, Kernel128_winograd.cu:236-256, and a CUDA error exits, :16-22).  The stream-K / split-C
 * kernels hand partial sums between workgroups through ticket counters that must be zero when a launch
 * begins; every launch returns them to zero.  A launch that fails on the host side, or a kernel that draws
 * a ticket on a counter that cannot have been zero at launch (a launch that died mid-way before it), puts
 * the (device, stream) into an error state: from then on every compute entry point on that stream returns
 * WINO_E_STATE instead of computing with counters it cannot trust.
 *   wino_stream_check          waits for the stream; WINO_OK or WINO_E_STATE
 *   wino_stream_reset_scratch  waits for the stream, zeroes its counters (slabs are kept), clears the state */
int wino_stream_check(wino_stream_t stream);
int wino_stream_reset_scratch(wino_stream_t stream);
/* events: timing on the stream the kernels run on (hipEvent based) */
int wino_event_create(void** event);
int wino_event_destroy(void* event);
int wino_event_record(void* event, wino_stream_t stream);
int wino_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */

/* ---- 3x3: Winograd-domain filters ------------------------------------------ */
/* Number of floats of the packed F(2x2,3x3) filter buffer `U` for C in-channels and
 * K out-channels (= 16*C*K).  Its internal layout ([C/8][K/64][16][64][8], LDS-bank
 * swizzled) is private to the library. */
size_t wino_filter_f2_elems(int C, int K);
/* Position (in floats) of Winograd point e (0..15 = 4*row+col of G g G^T), in-channel c, out-channel k
 * inside the packed buffer; -1 for out-of-range arguments.  Host-side, no GPU needed. */
long wino_filter_f2_index(int C, int K, int e, int c, int k);
/* w_kcrs: [K][C][3][3] (weight_NCHW_C_K.bin) -> U.  G g G^T evaluated in fp64, stored fp32. */
int wino_filter_transform_f2(const float* w_kcrs, float* U, int C, int K, wino_stream_t s);
/* u36: the reference's pre-transformed F(4x4,3x3) weights [36][C][K]
 * (weight_winograd_C_K.bin, data_generator.py:63-78).  The 3x3 taps are recovered
 * exactly (g = L u L^T, L = left inverse of the reference's G) and re-transformed to
 * F(2x2,3x3), so the reference's weight file is consumed as is. */
int wino_filter_import_f4(const float* u36, float* U, int C, int K, wino_stream_t s);

/* ---- 3x3 conv + folded BN + ReLU ---------------------------------------------
 * in  [N][16][16][C]   NHWC, ring included (a valid conv of the 16x16 image)
 * U   from wino_filter_transform_f2 / wino_filter_import_f4
 * out [N][16][16][K]   14x14 result at [1..14][1..14], ring written as 0
 *                       (= the next 3x3 layer's padded input, Kernel128_winograd.cu:163)
 * out = relu(bnScale[k] * conv + bnBias[k]);  argument order (in, bias, scale, out)
 * follows kernel_*_winograd_AtIA (Kernel128_winograd.cu:123).
 * Constraints: C % 8 == 0, K % 64 == 0, N >= 1.  One launch.  The throughput kernel splits the
 * work evenly over the CUs (stream-K) and hands partial sums between workgroups through a small
 * scratch buffer the LIBRARY owns, one per (device, stream), allocated at the first call that needs
 * it (a synchronous hipMalloc).  Launches on one stream serialise, so they share it safely; two
 * host threads must not launch on the SAME stream concurrently. */
int wino_conv3x3_bn_relu(const float* in, const float* U, const float* bnBias,
                         const float* bnScale, float* out, int N, int C, int K, int relu,
                         wino_stream_t s);
/* Any N >= 1: one launch addresses its tensors with 32-bit byte offsets, so a batch whose input or
 * output would reach 4 GiB (N >= 16384 at 256 channels) goes out as several launches of whole images
 * on `s`.  wino_conv3x3_plan describes one launch and rejects such a batch. */
/* Allocates that scratch for (current device, `s`) and this shape ahead of time -- e.g. before
 * capturing wino_conv3x3_bn_relu / wino_residual_block into a HIP graph, where an allocation inside
 * the capture is not allowed.  Optional otherwise. */
int wino_conv3x3_prepare(int N, int C, int K, wino_stream_t s);

/* Other feature-map sizes (ResNet's 56x56 and 28x28 stages; the reference hard-codes
 * 14x14): H x W outputs (odd sizes such as the 7x7 stage included: the last tile row / column is
 * clipped), in [N][H+2][W+2][C], out [N][H+2][W+2][K] with the result at [1..H][1..W] and the ring
 * written as 0.  Same kernel, same packed filters; H = W = 14 is exactly
 * wino_conv3x3_bn_relu.  The latency kernel for tiny batches exists for 14x14 only. */
int wino_conv3x3_bn_relu_hw(const float* in, const float* U, const float* bnBias,
                            const float* bnScale, float* out, int N, int H, int W, int C, int K,
                            int relu, wino_stream_t s);
int wino_conv3x3_prepare_hw(int N, int H, int W, int C, int K, wino_stream_t s);

/* Host-side only (no GPU needed): the launch plan of the throughput kernel on a device with `cus`
 * compute units -- `grid` logical workgroups run `rounds` whole items each (item = 64 tiles x 64
 * out-channels, `iters_per_item` = C/8 chunk iterations) and share `tail_iters` further iterations
 * as a stream-K tail: workgroup l takes tail iterations [l*tail_iters/grid, (l+1)*tail_iters/grid). */
int wino_conv3x3_plan(int N, int H, int W, int C, int K, int cus, int* grid, int* rounds, long* tail_iters,
                      int* iters_per_item);
/* Host-side only: how that launch cuts and places its tail.  *groups = 1: one item-major list of tail_iters iterations
 * as described above; K/64 (taken whenever grid is a multiple of it): one list per out-channel block -- group k owns
 * the tail items of k-block k (item rounds*grid + groups*j + k, j = 0, 1, ...: one per tile block) and tail_iters /
 * groups iterations, cut into Gp = grid / groups equal ranges.  The workgroup l = groups*j + k at position j of group
 * k runs range  (phase_inv * (j / phase_copies)) % phase_period + phase_period * (j % phase_copies)  of its group: a
 * permutation of 0 .. Gp-1 (the identity for period 1) that gives the workgroups of one XCD -- consecutive positions
 * -- ranges starting at consecutive channel phases.  The groups' workgroups at one position walk the same tile blocks
 * and channel chunks in step and share the patches in the XCD's L2; neighbours in phase share the filter chunks. */
int wino_conv3x3_plan_groups(int N, int H, int W, int C, int K, int cus, int* groups, int* phase_period, int* phase_inv,
                             int* phase_copies);
/* Host-side only: whether this shape takes the latency kernel instead (small batches of the 14x14 stage: the
 * reference's own N = 1), and in which form: *point_rows (4, 2 or 1 rows of the 4x4 point grid per wave task)
 * and *split (workgroups that share one 16-tile x 16-out-channel block's contraction, meeting through
 * library-owned slabs + tickets when > 1); *workgroups = blocks x split. */
int wino_conv3x3_small_plan(int N, int H, int W, int C, int K, int cus, int* use, int* point_rows, int* split,
                            int* workgroups);
/* The same with the block width: a wave holds *col_tiles MFMA tiles side by side (a block = 16 tiles x
 * 16 col_tiles out-channels; 1 at the reference's N = 1, 2 or 4 for the batches between that and the
 * throughput kernel's range, two point rows per task then). */
int wino_conv3x3_small_plan2(int N, int H, int W, int C, int K, int cus, int* use, int* point_rows, int* split,
                             int* col_tiles, int* workgroups);

/* ---- F(4x4,3x3) compatibility path ------------------------------
 * The reference's own three-stage arithmetic on its own pre-transformed weight file, consumed as is:
 * u36 = weight_winograd_C_K.bin, [36][C][K] (data_generator.py:63-78).  V = B^T d B (6x6 patches, 16
 * tiles per image), M_e = V_e . U_e for the 36 points (one batched MFMA GEMM launch), out = relu(scale *
 * A^T M A + bias) clipped to 14x14 (Kernel128_winograd.cu:28-213).  V and M live in `workspace`
 * (wino_conv3x3_f4_workspace_bytes), like the reference's t_input / ip buffers.  Same in / out layout as
 * wino_conv3x3_bn_relu.  Unfused and HBM-bound by design; the product path is the fused F(2x2) kernel.
 * Constraints: C % 32 == 0, K % 64 == 0. */
size_t wino_conv3x3_f4_workspace_bytes(int N, int C, int K);
int wino_conv3x3_f4_bn_relu(const float* in, const float* u36, const float* bnBias, const float* bnScale,
                            float* out, int N, int C, int K, int relu, void* workspace,
                            size_t workspace_bytes, wino_stream_t s);

/* Independent comparator: direct (non-Winograd) 3x3 conv + BN + ReLU on the GPU,
 * w_kcrs [K][C][3][3]; same in/out layout as above.  Slow by design. */
int wino_conv3x3_direct(const float* in, const float* w_kcrs, const float* bnBias,
                        const float* bnScale, float* out, int N, int C, int K, int relu,
                        wino_stream_t s);

int wino_conv3x3_direct_hw(const float* in, const float* w_kcrs, const float* bnBias,
                           const float* bnScale, float* out, int N, int H, int W, int C, int K,
                           int relu, wino_stream_t s);

/* ---- 1x1 conv as GEMM + folded BN (+ReLU) --------------------------------------
 * A [M][Cin] (M = N*196 pixels, HWC flat), B [Cin][Kout] row-major, C [M][Kout].
 * C = bnScale[k]*(A.B) + bnBias[k], ReLU if `relu`.  Argument order as the reference
 * kernels (A, B, bnBias, bnScale, C), Kernel128_one.cu:24.
 * Constraints: Cin % 32 == 0, Kout % 64 == 0 (workgroups are 128 columns wide when Kout % 128 == 0
 * and both dimensions exceed 128, 64 columns otherwise), M >= 1 (any M: the last row tile is ragged). */
int wino_conv1x1_bn(const float* A, const float* B, const float* bnBias, const float* bnScale,
                    float* C, long M, int Cin, int Kout, int relu, wino_stream_t s);
/* Extended form used when layers are chained (the residual block):
 *   WINO_RELU          ReLU after BN (+ residual)
 *   WINO_A_PADDED      A's rows are the 14x14 interior pixels of a padded [N][16][16][Cin] tensor
 *                      (what wino_conv3x3_bn_relu writes), M = N*196
 *   WINO_C_PADDED      C's rows go to the interior of a padded [N][16][16][Kout] tensor and the ring
 *                      is written as 0 (what wino_conv3x3_bn_relu reads), M = N*196
 *   WINO_ADD_RESIDUAL  C = act(bnScale*(A.B) + bnBias + residual), residual [M][Kout] unpadded
 * The reference has no such glue: its 1x1 layers are unpadded [196][C] and its 3x3 layers padded
 * [16][16][C], and no kernel adds the skip connection. */
#define WINO_RELU 1
#define WINO_A_PADDED 2
#define WINO_C_PADDED 4
#define WINO_ADD_RESIDUAL 8
int wino_conv1x1_bn_ex(const float* A, const float* B, const float* bnBias, const float* bnScale,
                       const float* residual, float* C, long M, int Cin, int Kout, int flags,
                       wino_stream_t s);
/* The same for any feature-map size: A [N*H*W][Cin] or, with WINO_A_PADDED,
 * [N][H+2][W+2][Cin]; C [N*H*W][Kout] or, with WINO_C_PADDED, [N][H+2][W+2][Kout] with its ring
 * written as 0 -- the layouts wino_conv3x3_bn_relu_hw reads and writes.  H = W = 14 is exactly
 * wino_conv1x1_bn_ex with M = N*196. */
int wino_conv1x1_bn_ex_hw(const float* A, const float* B, const float* bnBias, const float* bnScale,
                          const float* residual, float* C, int N, int H, int W, int Cin, int Kout,
                          int flags, wino_stream_t s);
/* Shapes whose tile count leaves the last round of workgroups mostly empty (the reference's
 * 512->128 and 1024->256 layers at N = 128: 448 tiles on 256 CUs) are launched in stream-K form
 * and use library-owned scratch of stream `s`, allocated on the first such launch.  Call this
 * once per (shape, stream) before capturing the layer into a HIP graph; it launches nothing.
 * Results do not depend on the launch form chosen beyond fp32 summation order, and are bitwise
 * reproducible from launch to launch.  WINO_1X1_SK=0 in the environment disables the form. */
int wino_conv1x1_prepare(long M, int Cin, int Kout, wino_stream_t s);
/* Host-side only (no GPU needed): the launch form of this shape on a device with `cus` compute
 * units.  The output is `row_tiles` x `col_blocks` tiles (112 rows x 64 or 128 columns) of
 * `k_steps` = Cin/32 pipeline steps.  stream_k = 0: `grid` workgroups, one whole tile each (some of
 * the grid may be padding).  stream_k = 1: the (row tile, k-step) space is cut into grid/col_blocks
 * equal ranges -- range r covers [r*T/R, (r+1)*T/R) of T = row_tiles*k_steps, R = grid/col_blocks --
 * and logical workgroup r*col_blocks + nb runs range r for column block nb. */
int wino_conv1x1_plan(long M, int Cin, int Kout, int cus, int* grid, int* row_tiles, int* col_blocks,
                      int* k_steps, int* stream_k);
/* Host-side only: plain layers (wino_conv1x1_bn; no padded operand, no residual) with few pixel rows -- the
 * reference's own M = 196 -- take a latency form instead of the tiled kernel wino_conv1x1_plan describes:
 * 16 x 16 output blocks, 4 waves per workgroup, a block's K loop split over *k_split of them (4, 2 or 1).
 * *use = 0: the tiled kernel runs. */
int wino_conv1x1_small_plan(long M, int Cin, int Kout, int cus, int* use, int* k_split, int* workgroups);
/* The same with the block shape: a wave holds *row_tiles x *col_tiles MFMA tiles (16 x 16 each; 1 x 1 at M = 196,
 * 2 x 2 -- half the operand bytes per FLOP -- from a few images on). */
int wino_conv1x1_small_plan2(long M, int Cin, int Kout, int cus, int* use, int* k_split, int* row_tiles, int* col_tiles,
                             int* workgroups);
/* Host-side only (developer aid): the two launch models' times for this shape, the latency form's best candidate and the
 * tiled kernel's, in microseconds; the latency form is taken while the first beats the second by the policy's margin. */
int wino_debug_conv1x1_models(long M, int Cin, int Kout, int cus, double* t_latency_us, double* t_tiled_us);

/* ---- ResNet bottleneck block of the 14x14 stage (BASELINE.json configs[4]) ---------
 * out = relu( bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(x, w1))), U2))), w3)) + x )
 * x, out [N][14][14][C4] (unpadded, = [N*196][C4]); w1 [C4][Cm], w3 [Cm][C4] (the reference's
 * [Cin][Kout] 1x1 layout); U2 = packed F(2x2,3x3) filters of the Cm->Cm 3x3 layer; all BN folded.
 * Three launches on `s`, intermediates in `workspace` (wino_residual_block_workspace_bytes). */
size_t wino_residual_block_workspace_bytes(int N, int Cm);
int wino_residual_block(const float* x, const float* w1, const float* bn1Bias, const float* bn1Scale,
                        const float* U2, const float* bn2Bias, const float* bn2Scale,
                        const float* w3, const float* bn3Bias, const float* bn3Scale, float* out,
                        int N, int C4, int Cm, void* workspace, size_t workspace_bytes,
                        wino_stream_t s);
/* The block at any feature-map size (ResNet's 56x56 / 28x28 / 7x7 stages):
 * x, out [N][H][W][C4].  H = W = 14 is exactly wino_residual_block. */
size_t wino_residual_block_workspace_bytes_hw(int N, int H, int W, int Cm);
int wino_residual_block_hw(const float* x, const float* w1, const float* bn1Bias, const float* bn1Scale,
                           const float* U2, const float* bn2Bias, const float* bn2Scale,
                           const float* w3, const float* bn3Bias, const float* bn3Scale, float* out,
                           int N, int H, int W, int C4, int Cm, void* workspace, size_t workspace_bytes,
                           wino_stream_t s);

/* Allocates the library-owned scratch the block's three launches on stream `s` will use, ahead of a
 * graph capture (the block's analogue of wino_conv3x3_prepare / wino_conv1x1_prepare). */
int wino_residual_block_prepare(int N, int C4, int Cm, wino_stream_t s);
int wino_residual_block_prepare_hw(int N, int H, int W, int C4, int Cm, wino_stream_t s);

/* Independent comparator for the 1x1 layers: one thread per output, fp32 FMA loop. */
int wino_conv1x1_direct(const float* A, const float* B, const float* bnBias,
                        const float* bnScale, float* C, long M, int Cin, int Kout, int relu,
                        wino_stream_t s);

/* ---- driver configuration for the argument-less reference entry points ---------
 * kernel_128() & co take no arguments (Kernel128_winograd.h:20); batch size, GPU count
 * and verbosity come from here (defaults N=1, 1 GPU = the reference's behaviour) or
 * from the environment (WINO_BATCH, WINO_GPUS, WINO_QUIET) at first use. */
int wino_driver_set_batch(int N);
int wino_driver_set_gpus(int ngpu);
int wino_driver_set_quiet(int quiet);
int wino_driver_get_batch(void);
int wino_driver_get_gpus(void);
/* Last call's unpacked results (the packed int overflows at 65.5 ms, Test.c:46-47). */
typedef struct {
  double mine_us;        /* custom path, wall clock launch..sync (max over GPUs) */
  double comparator_us;  /* direct-conv comparator, same protocol */
  double max_abs_err;    /* output_checker semantics (util.c:46-63) over all images */
  double max_rel_err;    /* max|diff| / max|comparator| */
  long error_cnt;        /* |diff| > 1e-5, reference threshold */
  double flops;          /* algorithmic FLOPs of the layer call (2*N*P*Q*K*C*R*S) */
  int N, gpus;
  double steady_us;      /* mean of 100 further back-to-back launches (warm), max over GPUs */
} wino_driver_result;
int wino_driver_last_result(wino_driver_result* r);
/* The custom path's output of the last kernel_*() call, on the host: [N][16][16][K] (3x3, ring 0) or
 * [N*196][Kout] (1x1); *elems receives the element count.  Valid until the next kernel_*() call.
 * NULL before the first call.  (The reference keeps this in a local array, Kernel128_winograd.cu:257.) */
const float* wino_driver_last_output(size_t* elems);
/* The packed return value of the kernel_*() entry points, (mine << 16) | comparator
 * (Kernel128_winograd.cu:433), with the custom half clamped to 0x7FFF and the comparator half to
 * 0xFFFF so that the reference's signed decode `res >> 16`, `res & 0xFFFF` (Test.c:46-47) never
 * sees a negative number. */
int wino_driver_pack_times(uint64_t mine_us, uint64_t comparator_us);
/* Developer knob (WINO_GPUS_ALIAS=1): job g of a multi-GPU call runs on device g % visible, so the
 * threaded batch-split path (one host thread + one stream per job, common start barrier) can be
 * exercised on a box with fewer GPUs than requested. */
int wino_driver_set_gpu_alias(int on);
/* WINO_STDOUT_COMPAT=1: the per-call lines carry the reference's exact labels -- "cuDNN TotalTime =
 * %d us" and cuda-prefixed status names (Kernel128_winograd.cu:270-275,404-409) -- for scrapers written
 * against the reference; ./Test then also prints "[cuDNN: %d us]" (Test.c:50-53) and nothing extra.
 * The comparator behind that label is this library's direct-convolution kernel, not cuDNN. */
int wino_driver_set_stdout_compat(int on);
int wino_driver_get_stdout_compat(void);
/* CPU baseline of the layer the LAST kernel_*() call ran (same inputs, same N): naive im2col +
 * three-loop SGEMM + folded BN (+ReLU) on all host cores, timed with the wall clock (one warm-up,
 * then repetitions for about half a second), and diffed against that call's GPU output.
 * A reported baseline, never a fallback: nothing the library returns comes from it. */
typedef struct {
  double us;             /* mean wall time of one CPU pass over the layer */
  double gflops;         /* algorithmic FLOPs / us */
  int threads;           /* host threads used = cores this process may run on */
  int reps;
  double max_abs_diff;   /* max |GPU - CPU| over the valid outputs */
  double max_rel_diff;   /* ... / max |CPU| */
} wino_cpu_baseline_result;
int wino_driver_cpu_baseline(wino_cpu_baseline_result* r);

/* ---- diagnostics (measurement infrastructure, not part of the reference interface) -------------
 * Re-reads the WINO_* developer knobs (the library reads them once per process). */
int wino_debug_reload_knobs(void);
/* Synchronises `s` and counts the non-zero stream-K ticket counters of its scratch on the current device
 * (0 when the stream has none).  Every launch leaves them at zero: a non-zero count between launches
 * means an item was never finalized (tests assert 0). */
int wino_debug_tickets_in_use(wino_stream_t s, long* nonzero);
/* Test hook: overwrites ticket counter `index` of the stream's scratch on the current device with `value`,
 * as a launch that died mid-way would leave it. */
int wino_debug_poison_ticket(wino_stream_t s, long index, unsigned value);
/* The clock the chip held inside the MOST RECENT launch of a product kernel on the current device:
 * workgroup 0 of every launch stores {s_memtime, s_memrealtime} at its entry and at its exit into a
 * 32-byte slot of the code object (four stores per launch; no output depends on them).  kernel: 0 = the
 * fused 3x3 throughput kernel, 1 = the 1x1 GEMM kernel.  Synchronises `s`, then stamps[0..3] = {cycles,
 * 100 MHz ticks} at entry, the same pair at exit: clock = (stamps[2] - stamps[0]) / (stamps[3] - stamps[1])
 * * 0.1 GHz.  bench.py reads it after the last launch of a timed burst -- the clock OF the timed region. */
int wino_diag_last_clock(int kernel, wino_stream_t s, unsigned long long stamps[4]);
/* The 3x3 throughput kernel's stamped build (same source, s_memtime / s_memrealtime around its main
 * loop): runs one launch of it on `s` with the arguments of wino_conv3x3_bn_relu and writes four
 * uint64 per workgroup to stamps_dev (at least 4 * 2048 uint64): {shader cycles, 100 MHz ticks} at the
 * start of its main loop and the same pair at its end.  *workgroups receives the number of
 * quadruples.  in-kernel clock = d(cycles) / d(ticks) * 0.1 GHz.  14x14 only. */
int wino_diag_conv3x3_clock(const float* in, const float* U, const float* bnBias, const float* bnScale,
                            float* out, int N, int C, int K, unsigned long long* stamps_dev,
                            int* workgroups, wino_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* WINOGRAD_MI355X_H */
