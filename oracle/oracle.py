"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A numpy restatement of the reference's conv(+BN+ReLU) hot path.  Nothing in the
product (``cuda-winograd_amd/``) may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only
as the checker.

Parity status: the reference ships no golden outputs (``data/*.bin`` is
git-ignored, reference .gitignore:28; its only check is a run-time diff against
cuDNN, util.c:46-63).  The oracle is therefore pinned as follows:
  * inputs / offline transforms: byte-identical to the reference generator
    imported in the build container (tests/golden/reference_files.json holds the
    SHA-256 of every file for seeds 0 and 1; tests/golden/make_golden.py is the
    script that made them);
  * outputs: ``tests/golden/*.npz`` were produced by ``make_golden.py`` from the
    *reference generator's own files* with the fp64 direct convolution below,
    and the stage-by-stage restatement of the reference kernels
    (``winograd_f4_reference``) is checked against it.
  * the cuDNN comparator itself: parity unpinned (closed source, no recorded
    outputs); its role is taken by the fp64 direct convolution.

Layouts:
  input   [N][16][16][C]  NHWC, the 1-pixel ring is part of the data
  weights [K][C][3][3]    (weight_NCHW_C_K.bin) / [36][C][K] (weight_winograd_C_K.bin)
  output  [N][16][16][K]  14x14 result in the interior, ring exactly 0
  1x1:    A [M][Cin], B [Cin][Kout], C [M][Kout]
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# Transform matrices
# --------------------------------------------------------------------------
# F(4x4,3x3): reference Kernel128_winograd.cu:44-70 (B^T), :138-147,162-179 (A^T),
# data_generator.py:65 (G)
BT_F4 = np.array([
    [4, 0, -5, 0, 1, 0],
    [0, -4, -4, 1, 1, 0],
    [0, 4, -4, -1, 1, 0],
    [0, -2, -1, 2, 1, 0],
    [0, 2, -1, -2, 1, 0],
    [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT_F4 = np.array([
    [1, 1, 1, 1, 1, 0],
    [0, 1, -1, 2, -2, 0],
    [0, 1, 1, 4, 4, 0],
    [0, 1, -1, 8, -8, 1]], dtype=np.float64)
G_F4 = np.array([
    [0.25, 0, 0],
    [-1.0 / 6, -1.0 / 6, -1.0 / 6],
    [-1.0 / 6, 1.0 / 6, -1.0 / 6],
    [1.0 / 24, 1.0 / 12, 1.0 / 6],
    [1.0 / 24, -1.0 / 12, 1.0 / 6],
    [0, 0, 1]], dtype=np.float64)
# F(2x2,3x3) (Lavin & Gray 2015) -- the algorithm the HIP path runs
BT_F2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT_F2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
G_F2 = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)

H = W = 16   # padded input extent
P = Q = 14   # output extent


# --------------------------------------------------------------------------
# Ground truth: fp64 direct cross-correlation + folded BN + ReLU
# (what the reference's cuDNN half computes: Kernel128_winograd.cu:352,384-399)
# --------------------------------------------------------------------------
def im2col_3x3(inp: np.ndarray) -> np.ndarray:
    """[N,Hp,Wp,C] -> [N*(Hp-2)*(Wp-2), 9*C] with column order (r, s, c); the reference's case is
    Hp = Wp = 16."""
    N, Hp, Wp, C = inp.shape
    p, q = Hp - 2, Wp - 2
    cols = np.empty((N, p, q, 3, 3, C), dtype=inp.dtype)
    for r in range(3):
        for s in range(3):
            cols[:, :, :, r, s, :] = inp[:, r:r + p, s:s + q, :]
    return cols.reshape(N * p * q, 9 * C)


def conv3x3_bn_relu_direct(inp, w_kcrs, bn_scale, bn_bias, relu=True, dtype=np.float64):
    """Valid 3x3 cross-correlation of the padded image, y = relu(scale*conv + bias),
    written into the interior of a zero [N,Hp,Wp,K] buffer (reference output
    layout, Kernel128_winograd.cu:163; the reference's size is Hp = Wp = 16)."""
    inp = np.asarray(inp, dtype=dtype)
    N, Hp, Wp, C = inp.shape
    K = w_kcrs.shape[0]
    wmat = np.asarray(w_kcrs, dtype=dtype).transpose(2, 3, 1, 0).reshape(9 * C, K)  # (r,s,c) x k
    y = im2col_3x3(inp) @ wmat
    y = y * np.asarray(bn_scale, dtype)[None, :] + np.asarray(bn_bias, dtype)[None, :]
    if relu:
        y = np.maximum(y, 0)
    out = np.zeros((N, Hp, Wp, K), dtype=dtype)
    out[:, 1:Hp - 1, 1:Wp - 1, :] = y.reshape(N, Hp - 2, Wp - 2, K)
    return out


# --------------------------------------------------------------------------
# Stage-by-stage restatement of the reference's F(4x4,3x3) kernels (fp32)
# --------------------------------------------------------------------------
def f4_input_transform(inp):
    """kernel_*_winograd_BtdB (Kernel128_winograd.cu:28-120).
    [N,16,16,C] -> V [36][N*16 tiles][C].  Tile (tx,ty) covers rows 4tx..4tx+5,
    cols 4ty..4ty+5; reads past row/col 15 see zeros (the reference zero-fills a
    double-sized buffer, :236,242; column over-reads wrap into the next row there,
    but only feed outputs that AtIA drops, :155,171,177)."""
    inp = np.asarray(inp, np.float32)
    N, _, _, C = inp.shape
    padded = np.zeros((N, 18, 18, C), np.float32)
    padded[:, :16, :16, :] = inp
    V = np.empty((36, N, 4, 4, C), np.float32)
    BT = BT_F4.astype(np.float32)
    for tx in range(4):
        for ty in range(4):
            d = padded[:, 4 * tx:4 * tx + 6, 4 * ty:4 * ty + 6, :]          # [N,6,6,C]
            btd = np.einsum('ij,njkc->nikc', BT, d).astype(np.float32)        # rows  (:42-73)
            v = np.einsum('nikc,lk->nilc', btd, BT).astype(np.float32)        # cols  (:83-114)
            V[:, :, tx, ty, :] = v.reshape(N, 36, C).transpose(1, 0, 2)
    return V.reshape(36, N * 16, C)


def f4_outer_product(V, U36):
    """kernel_*_OuterProduct_* (Kernel128_winograd.cu:186-213): M_e = V_e @ U_e."""
    return np.einsum('etc,eck->etk', V, np.asarray(U36, np.float32)).astype(np.float32)


def f4_output_transform(M, bn_scale, bn_bias, N):
    """kernel_*_winograd_AtIA (Kernel128_winograd.cu:123-183): Y = A^T M A, BN, ReLU,
    clipped to 14x14 and written at [4tx+1+a][4ty+1+b]."""
    K = M.shape[2]
    AT = AT_F4.astype(np.float32)
    m = M.reshape(6, 6, N, 4, 4, K)
    y = np.einsum('ai,ijntuk->ajntuk', AT, m).astype(np.float32)
    y = np.einsum('ajntuk,bj->abntuk', y, AT).astype(np.float32)             # [4,4,N,4,4,K]
    y = np.asarray(bn_scale, np.float32) * y + np.asarray(bn_bias, np.float32)
    y = np.maximum(y, 0)
    full = np.zeros((N, 18, 18, K), np.float32)
    for tx in range(4):
        for ty in range(4):
            full[:, 4 * tx + 1:4 * tx + 5, 4 * ty + 1:4 * ty + 5, :] = \
                y[:, :, :, tx, ty, :].transpose(2, 0, 1, 3)
    out = np.zeros((N, H, W, K), np.float32)
    out[:, 1:15, 1:15, :] = full[:, 1:15, 1:15, :]
    return out


def winograd_f4_reference(inp, U36, bn_scale, bn_bias):
    """The reference's three launches chained (Kernel128_winograd.cu:263-265)."""
    N = inp.shape[0]
    V = f4_input_transform(inp)
    M = f4_outer_product(V, U36)
    return f4_output_transform(M, bn_scale, bn_bias, N)


# --------------------------------------------------------------------------
# F(2x2,3x3) restatement (the algorithm of the HIP path), fp32
# --------------------------------------------------------------------------
def f2_filter_transform(w_kcrs, dtype=np.float64):
    """U[e=4x+y][c][k] = (G g_{k,c} G^T)[x][y]."""
    g = np.asarray(w_kcrs, dtype)
    G = G_F2.astype(dtype)
    u = np.einsum('xr,kcrs,ys->xyck', G, g, G)
    return u.reshape(16, g.shape[1], g.shape[0])


def f2_input_transform(inp):
    """[N,16,16,C] -> V[16][N*49][C]; tile (ty,tx) covers rows 2ty..2ty+3, cols 2tx..2tx+3."""
    inp = np.asarray(inp, np.float32)
    N, _, _, C = inp.shape
    BT = BT_F2.astype(np.float32)
    V = np.empty((16, N, 7, 7, C), np.float32)
    for ty in range(7):
        for tx in range(7):
            d = inp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4, :]
            v = np.einsum('ij,njkc,lk->nilc', BT, d, BT).astype(np.float32)
            V[:, :, ty, tx, :] = v.reshape(N, 16, C).transpose(1, 0, 2)
    return V.reshape(16, N * 49, C)


def winograd_f2(inp, w_kcrs, bn_scale, bn_bias, relu=True):
    N = inp.shape[0]
    K = w_kcrs.shape[0]
    U = f2_filter_transform(w_kcrs).astype(np.float32)
    V = f2_input_transform(inp)
    M = np.einsum('etc,eck->etk', V, U).astype(np.float32).reshape(4, 4, N, 7, 7, K)
    AT = AT_F2.astype(np.float32)
    y = np.einsum('ai,ijntuk,bj->ntaubk', AT, M, AT).astype(np.float32)       # [N,7,2,7,2,K]
    y = y.reshape(N, 14, 14, K)
    y = np.asarray(bn_scale, np.float32) * y + np.asarray(bn_bias, np.float32)
    if relu:
        y = np.maximum(y, 0)
    out = np.zeros((N, H, W, K), np.float32)
    out[:, 1:15, 1:15, :] = y
    return out


# --------------------------------------------------------------------------
# 1x1 layers (Kernel128_one.cu:24-54,244-273; Kernel256_one.cu:26-56,246-274)
# --------------------------------------------------------------------------
def conv1x1_bn(A, B, bn_bias, bn_scale, relu, dtype=np.float64):
    """C = scale * (A @ B) + bias, optional ReLU.  A [M][Cin], B [Cin][Kout].
    Argument order (A, B, bnBias, bnScale) follows the reference kernels."""
    y = np.asarray(A, dtype) @ np.asarray(B, dtype)
    y = np.asarray(bn_scale, dtype)[None, :] * y + np.asarray(bn_bias, dtype)[None, :]
    if relu:
        y = np.maximum(y, 0)
    return y


def residual_block(x, w1, bn1, w2_kcrs, bn2, w3, bn3, dtype=np.float64):
    """ResNet bottleneck as the composition of the three layer oracles (not in the reference,
    SURVEY.md D5 / section 8f): x [N][h][w][C4] (the reference's stage is 14 x 14);
    1x1 C4->Cm +BN+ReLU, 3x3 Cm->Cm (zero padding 1) +BN+ReLU, 1x1 Cm->C4 +BN, + x, ReLU.
    bnX = (bias, scale) folded."""
    x = np.asarray(x, dtype)
    N, h, w, C4 = x.shape
    Cm = np.asarray(w1).shape[1]
    t1 = conv1x1_bn(x.reshape(-1, C4), w1, bn1[0], bn1[1], True, dtype).reshape(N, h, w, Cm)
    t1p = np.zeros((N, h + 2, w + 2, Cm), dtype)
    t1p[:, 1:-1, 1:-1, :] = t1
    t2 = conv3x3_bn_relu_direct(t1p, w2_kcrs, bn2[1], bn2[0], True, dtype)[:, 1:-1, 1:-1, :]
    t3 = conv1x1_bn(t2.reshape(-1, Cm), w3, bn3[0], bn3[1], False, dtype).reshape(N, h, w, C4)
    return np.maximum(t3 + x, 0)


# (Cin, Kout, relu) of the four 1x1 entry points, in ./Test mode order 2..5
ONE_BY_ONE_LAYERS = {
    "kernel_128_1_in": (512, 128, True),    # Kernel128_one.cu:53
    "kernel_128_1_out": (128, 512, False),  # Kernel128_one.cu:271-272
    "kernel_256_1_in": (1024, 256, True),   # Kernel256_one.cu:55
    "kernel_256_1_out": (256, 1024, False), # Kernel256_one.cu:273
}


# --------------------------------------------------------------------------
# The reference's checker (util.c:46-63)
# --------------------------------------------------------------------------
def output_checker(A, B, length, channel, shift):
    """A is the (optionally padded by `shift`) custom output, B the unpadded
    comparator output.  Returns (max_error, error_cnt) with the reference's
    absolute 1e-5 threshold."""
    A = np.asarray(A, np.float32).reshape(length + 2 * shift, length + 2 * shift, channel)
    B = np.asarray(B, np.float32).reshape(length, length, channel)
    diff = np.abs(A[shift:shift + length, shift:shift + length, :] - B)
    return float(diff.max()), int((diff > 1e-5).sum())


def rel_error(got, want):
    """max |got-want| / max|want| -- the relative metric used for the 1e-3 bar
    (BASELINE.json north_star)."""
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
