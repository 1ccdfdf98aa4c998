"""Pin the CPU oracle (numpy + C) against the golden vectors.

tests/golden/outputs_seed0.npz = fp64 direct convolution / GEMM (+BN, +ReLU) of the six
./Test layers, computed from the files the REFERENCE generator wrote (make_golden.py).
Here the oracle is re-run on this repo's own generator output (byte-identical inputs, see
test_generator.py) and every restatement -- fp64 direct, the stage-by-stage F(4x4,3x3)
restatement of the reference kernels (numpy and C), the F(2x2,3x3) restatement, the naive
C im2col baseline and the 1x1 GEMMs -- must agree with the golden outputs."""
import numpy as np
import pytest

from conftest import load_bin, ptr

TOL_REL = 1e-3  # BASELINE.json north_star: outputs within 1e-3 relative


def _layer3(data_dir, C):
    x = load_bin(data_dir, f"input_14_1_{C}.bin").reshape(1, 16, 16, C)
    w = load_bin(data_dir, f"weight_NCHW_{C}_{C}.bin").reshape(C, C, 3, 3)
    u36 = load_bin(data_dir, f"weight_winograd_{C}_{C}.bin").reshape(36, C, C)
    s = load_bin(data_dir, f"bnScale_winograd_{C}.bin")
    b = load_bin(data_dir, f"bnBias_winograd_{C}.bin")
    return x, w, u36, s, b


@pytest.mark.parametrize("C", [128, 256])
def test_direct_fp64_reproduces_golden(C, data_dir, O, golden_outputs):
    x, w, _, s, b = _layer3(data_dir, C)
    y = O.conv3x3_bn_relu_direct(x, w, s, b)
    np.testing.assert_allclose(y[0, 1:15, 1:15, :], golden_outputs[f"kernel_{C}"], rtol=0, atol=1e-6)
    ring = np.ones((16, 16), bool); ring[1:15, 1:15] = False
    assert (y[0][ring] == 0).all()


@pytest.mark.parametrize("C", [128, 256])
def test_f4_reference_restatement(C, data_dir, O, golden_outputs):
    """The three reference launches restated (Kernel128_winograd.cu:28-213), fed with the
    reference's own pre-transformed weights: max abs err ~1e-5 (report.pdf section 5)."""
    x, _, u36, s, b = _layer3(data_dir, C)
    y = O.winograd_f4_reference(x, u36, s, b)
    g = golden_outputs[f"kernel_{C}"]
    assert np.abs(y[0, 1:15, 1:15, :] - g).max() < 1e-4
    assert O.rel_error(y[0, 1:15, 1:15, :], g) < TOL_REL
    ring = np.ones((16, 16), bool); ring[1:15, 1:15] = False
    assert (y[0][ring] == 0).all()
    # the reference checker on this pair: few elements above 1e-5 (report: < 0.1 %)
    max_err, cnt = O.output_checker(y[0], g, 14, C, 1)
    assert max_err < 1e-4 and cnt < 0.05 * g.size


@pytest.mark.parametrize("C", [128, 256])
def test_f2_restatement(C, data_dir, O, golden_outputs):
    x, w, _, s, b = _layer3(data_dir, C)
    y = O.winograd_f2(x, w, s, b)
    assert O.rel_error(y[0, 1:15, 1:15, :], golden_outputs[f"kernel_{C}"]) < 1e-5


def test_c_oracle_3x3(data_dir, O, c_oracle, golden_outputs):
    x, w, u36, s, b = _layer3(data_dir, 128)
    out = np.full((1, 16, 16, 128), np.nan, np.float32)
    assert c_oracle.oracle_conv3x3_im2col(ptr(x), ptr(w), ptr(s), ptr(b), ptr(out), 1, 128, 128, 1, 4) == 0
    assert O.rel_error(out[0, 1:15, 1:15, :], golden_outputs["kernel_128"]) < 1e-5
    out4 = np.full((1, 16, 16, 128), np.nan, np.float32)
    assert c_oracle.oracle_winograd_f4(ptr(x), ptr(u36), ptr(s), ptr(b), ptr(out4), 1, 128, 128) == 0
    assert np.abs(out4[0, 1:15, 1:15, :] - golden_outputs["kernel_128"]).max() < 1e-4
    ring = np.ones((16, 16), bool); ring[1:15, 1:15] = False
    assert (out[0][ring] == 0).all() and (out4[0][ring] == 0).all()


def test_c_oracle_batched_and_threaded(O, c_oracle):
    rng = np.random.RandomState(7)
    N, C, K = 3, 16, 64
    x = (rng.rand(N, 16, 16, C) - 0.5).astype(np.float32)
    w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    s = (rng.rand(K) - 0.5).astype(np.float32)
    b = (rng.rand(K) - 0.5).astype(np.float32)
    want = O.conv3x3_bn_relu_direct(x, w, s, b)
    for nt in (1, 3, 8):
        out = np.empty((N, 16, 16, K), np.float32)
        c_oracle.oracle_conv3x3_im2col(ptr(x), ptr(w), ptr(s), ptr(b), ptr(out), N, C, K, 1, nt)
        assert O.rel_error(out, want) < 1e-5
    norelu = np.empty((N, 16, 16, K), np.float32)
    c_oracle.oracle_conv3x3_im2col(ptr(x), ptr(w), ptr(s), ptr(b), ptr(norelu), N, C, K, 0, 2)
    assert (norelu < 0).any()
    assert O.rel_error(norelu, O.conv3x3_bn_relu_direct(x, w, s, b, relu=False)) < 1e-5


@pytest.mark.parametrize("name", ["kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in", "kernel_256_1_out"])
def test_one_by_one_layers(name, data_dir, O, c_oracle, golden_outputs):
    Cin, Kout, relu = O.ONE_BY_ONE_LAYERS[name]
    A = load_bin(data_dir, "input_one_14_1024.bin", 196 * Cin).reshape(196, Cin).copy()
    B = load_bin(data_dir, "weight_one_1024.bin", Cin * Kout).reshape(Cin, Kout).copy()
    s = load_bin(data_dir, "bnScale_myKernel_one_1024.bin", Kout).copy()
    b = load_bin(data_dir, "bnBias_myKernel_one_1024.bin", Kout).copy()
    g = golden_outputs[name]
    y = O.conv1x1_bn(A, B, b, s, relu)
    assert O.rel_error(y, g) < 1e-6
    y32 = O.conv1x1_bn(A, B, b, s, relu, dtype=np.float32)
    assert O.rel_error(y32, g) < 1e-5
    out = np.empty((196, Kout), np.float32)
    c_oracle.oracle_conv1x1(ptr(A), ptr(B), ptr(b), ptr(s), ptr(out), 196, Cin, Kout, int(relu), 4)
    assert O.rel_error(out, g) < 1e-5
    # ReLU only on the reducing layers (Kernel128_one.cu:53 vs :271-272)
    assert (g.min() >= 0) == relu


def test_output_checker_semantics(O):
    """util.c:46-63: A padded by `shift`, B not; counts |diff| > 1e-5."""
    rng = np.random.RandomState(1)
    B = rng.rand(14, 14, 8).astype(np.float32)
    A = np.zeros((16, 16, 8), np.float32)
    A[1:15, 1:15] = B
    assert O.output_checker(A, B, 14, 8, 1) == (0.0, 0)
    A[3, 4, 5] += 0.5
    m, c = O.output_checker(A, B, 14, 8, 1)
    assert abs(m - 0.5) < 1e-6 and c == 1
    assert O.output_checker(B, B, 14, 8, 0) == (0.0, 0)


def test_f4_weight_file_recovers_taps(data_dir, O):
    """g = L u L^T with L G4 = I: the identity wino_filter_import_f4 relies on."""
    _, w, u36, _, _ = _layer3(data_dir, 128)
    L = np.array([[4, 0, 0, 0, 0, 0], [0, -3, 3, 0, 0, 0], [0, 0, 0, 0, 0, 1]], np.float64)
    np.testing.assert_allclose(L @ O.G_F4, np.eye(3), atol=1e-15)
    g = np.einsum("ix,xyck,jy->kcij", L, u36.reshape(6, 6, 128, 128).astype(np.float64), L)
    assert np.abs(g - w).max() < 5e-7
