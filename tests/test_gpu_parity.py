"""GPU parity tests (run with `-m gpu` on an MI355X).  Everything goes through the C-ABI of
libwinograd_mi355x.so; the CPU oracle is only the checker.

Tolerance: BASELINE.json's north_star asks for outputs within 1e-3 RELATIVE of the
reference maths; `REL` below is that bar (max|got-want| / max|want|).  fp32 F(2x2,3x3) lands
around 1e-6, so a much tighter `TIGHT` is asserted too to catch indexing slips that a loose
bound would hide."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_bin

pytestmark = pytest.mark.gpu

REL = 1e-3
TIGHT = 2e-5


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch, torch.device("cuda:0")


def _t(torch_dev, a):
    torch, dev = torch_dev
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ring():
    r = np.ones((16, 16), bool)
    r[1:15, 1:15] = False
    return r


def _rand_layer(rng, N, C, K):
    x = (rng.rand(N, 16, 16, C) - 0.5).astype(np.float32)
    w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    s = (rng.rand(K) - 0.5).astype(np.float32)
    b = (rng.rand(K) - 0.5).astype(np.float32)
    return x, w, s, b


# ------------------------------------------------------------------ golden: Test-0 / Test-1
@pytest.mark.parametrize("C", [128, 256])
@pytest.mark.parametrize("weights", ["transform_f2", "import_f4"])
def test_reference_layers_match_golden(C, weights, data_dir, pkg, O, golden_outputs, torch_dev):
    """./Test 0 and ./Test 1 shapes on the reference's own seed-0 files, N = 1."""
    x = load_bin(data_dir, f"input_14_1_{C}.bin").reshape(1, 16, 16, C)
    s = load_bin(data_dir, f"bnScale_winograd_{C}.bin")
    b = load_bin(data_dir, f"bnBias_winograd_{C}.bin")
    if weights == "transform_f2":
        w = load_bin(data_dir, f"weight_NCHW_{C}_{C}.bin").reshape(C, C, 3, 3)
        U = pkg.filter_transform_f2(_t(torch_dev, w))
    else:  # the file the reference's custom path actually reads (Kernel128_winograd.cu:232)
        u36 = load_bin(data_dir, f"weight_winograd_{C}_{C}.bin").reshape(36, C, C)
        U = pkg.filter_import_f4(_t(torch_dev, u36))
    torch, _ = torch_dev
    out = torch.full((1, 16, 16, C), float("nan"), device="cuda:0")
    pkg.conv3x3_bn_relu(_t(torch_dev, x), U, _t(torch_dev, b), _t(torch_dev, s), out=out)
    got = out.cpu().numpy()
    g = golden_outputs[f"kernel_{C}"]
    assert O.rel_error(got[0, 1:15, 1:15, :], g) < TIGHT
    assert (got[0][_ring()] == 0).all(), "ring must be written as zero"
    max_err, cnt = O.output_checker(got[0], g, 14, C, 1)   # the reference's own check
    assert max_err < 1e-4 and cnt < 0.01 * g.size


@pytest.mark.parametrize("C", [128, 256])
def test_f4_compat_path_on_reference_files(C, data_dir, pkg, O, golden_outputs, torch_dev):
    """the reference's own F(4x4,3x3) arithmetic on its own
    weight_winograd_C_K.bin, consumed as is.  Against (a) the oracle's stage-by-stage restatement of
    the reference's three launches (same fp32 recipe: agreement to rounding), (b) the fp64
    direct-convolution golden output of ./Test 0 / ./Test 1 within the reference's own acceptance,
    (c) the fused F(2x2) product path; zero ring."""
    x = load_bin(data_dir, f"input_14_1_{C}.bin").reshape(1, 16, 16, C)
    s = load_bin(data_dir, f"bnScale_winograd_{C}.bin")
    b = load_bin(data_dir, f"bnBias_winograd_{C}.bin")
    u36 = load_bin(data_dir, f"weight_winograd_{C}_{C}.bin").reshape(36, C, C)
    got = pkg.conv3x3_f4_bn_relu(_t(torch_dev, x), _t(torch_dev, u36), _t(torch_dev, b), _t(torch_dev, s)).cpu().numpy()
    staged = O.winograd_f4_reference(x, u36, s, b)
    assert O.rel_error(got, staged) < 2e-5   # two fp32 pipelines, different summation order over C
    g = golden_outputs[f"kernel_{C}"]
    assert O.rel_error(got[0, 1:15, 1:15, :], g) < 1e-4          # F(4x4) in fp32: ~1e-5 (report section 5)
    max_err, cnt = O.output_checker(got[0], g, 14, C, 1)         # the reference's own check
    assert max_err < 1e-3
    assert (got[0][_ring()] == 0).all()
    f2 = pkg.conv3x3_bn_relu(_t(torch_dev, x), pkg.filter_import_f4(_t(torch_dev, u36)), _t(torch_dev, b),
                             _t(torch_dev, s)).cpu().numpy()
    assert O.rel_error(got, f2) < 1e-4


@pytest.mark.parametrize("N,C,K", [(13, 64, 128), (3, 160, 192)])
def test_f4_compat_path_batched(N, C, K, pkg, O, torch_dev):
    """Batch, several images per row tile of the batched GEMM (M = 16 N not a multiple of 112),
    C != K, no ReLU: against the fp64 direct convolution with filters transformed the reference's
    way (G g G^T in fp64, stored fp32).  K = 192 with C > 128: a column count that is a multiple of
    64 but not of 128 (the batched GEMM must take its 64-column form)."""
    rng = np.random.RandomState(91)
    x, w, s, b = _rand_layer(rng, N, C, K)
    G = O.G_F4
    u36 = np.einsum('xr,kcrs,ys->xyck', G, w.astype(np.float64), G).reshape(36, C, K).astype(np.float32)
    for relu in (True, False):
        got = pkg.conv3x3_f4_bn_relu(_t(torch_dev, x), _t(torch_dev, u36), _t(torch_dev, b), _t(torch_dev, s),
                                     relu=relu).cpu().numpy()
        want = O.conv3x3_bn_relu_direct(x, w, s, b, relu=relu)
        assert O.rel_error(got, want) < 1e-4
        assert (got[:, _ring(), :] == 0).all()


def test_filter_transforms_match_oracle(data_dir, pkg, O, torch_dev):
    """a12: the offline filter transform.  U = G g G^T (fp64 -> fp32) lands where
    wino_filter_f2_index says, for both entry points (raw taps and the reference's F(4x4) file)."""
    C = K = 128
    w = load_bin(data_dir, f"weight_NCHW_{C}_{K}.bin").reshape(K, C, 3, 3)
    u36 = load_bin(data_dir, f"weight_winograd_{C}_{K}.bin").reshape(36, C, K)
    want = O.f2_filter_transform(w).astype(np.float32)                   # [16][C][K]
    L = pkg.lib()
    e, c, k = np.meshgrid(np.arange(16), np.arange(C), np.arange(K), indexing="ij")
    # index of (e, c, k) -- vectorised from three probes per axis would hide bugs: call it for all
    idx = np.fromiter((L.wino_filter_f2_index(C, K, int(a), int(b), int(d))
                       for a, b, d in zip(e.ravel(), c.ravel(), k.ravel())), dtype=np.int64, count=e.size)
    U1 = pkg.filter_transform_f2(_t(torch_dev, w)).cpu().numpy()
    U2 = pkg.filter_import_f4(_t(torch_dev, u36)).cpu().numpy()
    assert np.array_equal(U1[idx], want.ravel())
    assert np.abs(U2[idx] - want.ravel()).max() < 2e-6       # taps recovered from fp32 F(4x4) weights


# ------------------------------------------------------------------ batch sizes, ragged blocks
@pytest.mark.parametrize("N,C,K", [(1, 8, 64), (2, 16, 64), (3, 128, 128), (5, 64, 192),
                                   (9, 256, 256), (17, 128, 256)])
def test_conv3x3_vs_oracle(N, C, K, pkg, O, torch_dev):
    """N*49 tiles never divide the 64-tile workgroup block: partial blocks, blocks that
    straddle image boundaries, several k-blocks, smallest legal C."""
    rng = np.random.RandomState(100 + N)
    x, w, s, b = _rand_layer(rng, N, C, K)
    U = pkg.filter_transform_f2(_t(torch_dev, w))
    for relu in (True, False):
        got = pkg.conv3x3_bn_relu(_t(torch_dev, x), U, _t(torch_dev, b), _t(torch_dev, s), relu=relu)
        got = got.cpu().numpy()
        want = O.conv3x3_bn_relu_direct(x, w, s, b, relu=relu)
        assert O.rel_error(got, want) < TIGHT
        assert (got[:, _ring(), :] == 0).all()


def test_conv3x3_full_size_properties(pkg, O, torch_dev):
    """BASELINE configs[2]: 256->256, N=128.  Checked by (a) the fp64 oracle on a sample of
    images, (b) the independent GPU comparator on every element, (c) size-independent
    properties: per-image independence (bitwise), linearity without ReLU, zero ring."""
    torch, dev = torch_dev
    rng = np.random.RandomState(42)
    N, C, K = 128, 256, 256
    x, w, s, b = _rand_layer(rng, N, C, K)
    xt, wt, st, bt = (_t(torch_dev, a) for a in (x, w, s, b))
    U = pkg.filter_transform_f2(wt)
    out = pkg.conv3x3_bn_relu(xt, U, bt, st)
    got = out.cpu().numpy()
    # (a) oracle on 6 images spread over tile-block boundaries
    idx = [0, 1, 63, 64, 100, 127]
    want = O.conv3x3_bn_relu_direct(x[idx], w, s, b)
    assert O.rel_error(got[idx], want) < TIGHT
    # (b) every element against the direct-convolution comparator kernel
    cmp_ = pkg.conv3x3_direct(xt, wt, bt, st).cpu().numpy()
    assert O.rel_error(got, cmp_) < TIGHT
    # (c1) ring
    assert (got[:, _ring(), :] == 0).all()
    # (c2) images are independent: a sub-batch gives the same values as the same images inside
    # the full batch (both run the throughput kernel; tile blocks straddle image boundaries
    # differently and the stream-K ranges cut the channel sums at different chunks in the two
    # runs, so equality is to rounding, not bitwise) ...
    sub = pkg.conv3x3_bn_relu(xt[37:101].contiguous(), U, bt, st).cpu().numpy()
    assert O.rel_error(sub, got[37:101]) < 2e-6
    # ... and an image run alone (N = 1 takes the latency kernel, which contracts the channels in
    # a different order) agrees to rounding
    for n in (0, 77, 127):
        alone = pkg.conv3x3_bn_relu(xt[n:n + 1].contiguous(), U, bt, st).cpu().numpy()
        assert O.rel_error(alone[0], got[n]) < TIGHT
    # (c3) linearity of conv+scale (no ReLU, zero bias): f(2x) == 2 f(x) exactly in fp32
    zb = torch.zeros_like(bt)
    f1 = pkg.conv3x3_bn_relu(xt[:4].contiguous(), U, zb, st, relu=False)
    f2 = pkg.conv3x3_bn_relu((2 * xt[:4]).contiguous(), U, zb, st, relu=False)
    assert torch.equal(f2, 2 * f1)
    # (c4) checksum of checksums: sum over k of the un-normalised conv equals conv with summed filters
    ones, zeros = torch.ones_like(st), torch.zeros_like(bt)
    y = pkg.conv3x3_bn_relu(xt[:2].contiguous(), U, zeros, ones, relu=False).cpu().numpy().astype(np.float64)
    wsum = w.astype(np.float64).sum(axis=0, keepdims=True)                     # [1][C][3][3]
    ysum = O.conv3x3_bn_relu_direct(x[:2], wsum, np.ones(1), np.zeros(1), relu=False)
    assert np.abs(y.sum(axis=3) - ysum[..., 0]).max() < 1e-3 * np.abs(ysum).max()


def test_conv3x3_is_deterministic_under_load(pkg, torch_dev):
    """The fused kernel hands LDS stages between LDS-DMA writers and ds_read readers with one
    barrier + vmcnt per chunk; a misplaced wait would show up as run-to-run differences (the
    arithmetic itself has a fixed order).  30 back-to-back launches, alternating with a second
    shape so that cache / LDS state differs between repeats, must agree bit for bit."""
    torch, dev = torch_dev
    g = torch.Generator(device="cpu").manual_seed(7)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    x, w, s, b = mk(128, 16, 16, 256), mk(256, 256, 3, 3), mk(256), mk(256)
    x2, w2, s2, b2 = mk(40, 16, 16, 128), mk(128, 128, 3, 3), mk(128), mk(128)
    U, U2 = pkg.filter_transform_f2(w), pkg.filter_transform_f2(w2)
    ref = pkg.conv3x3_bn_relu(x, U, b, s).clone()
    ref2 = pkg.conv3x3_bn_relu(x2, U2, b2, s2).clone()
    for _ in range(30):
        assert torch.equal(pkg.conv3x3_bn_relu(x, U, b, s), ref)
        assert torch.equal(pkg.conv3x3_bn_relu(x2, U2, b2, s2), ref2)


@pytest.mark.parametrize("N,C,K", [(128, 256, 256), (40, 128, 128), (17, 64, 192), (9, 8, 64)])
def test_conv3x3_streamk_decompositions_agree(N, C, K, pkg, O, torch_dev, knobs):
    """The throughput kernel splits T = items * C/8 chunk iterations evenly over G logical
    workgroups (stream-K); an item cut by a range boundary is finished by whichever workgroup
    draws its last ticket.  Every G must give the same values (to rounding: the cut points move
    the order of the channel sum), each G must be bitwise reproducible, and the per-item ticket
    counters must be back at zero afterwards (the next launch relies on it).  G = 8 makes long
    ranges (many whole items per workgroup), G = 2048 short ones (items cut into many segments,
    more workgroups than CUs: late workgroups find slabs published long before)."""
    torch, dev = torch_dev
    knobs.set("WINO_3X3_ALGO", "big")
    rng = np.random.RandomState(5 + N)
    x, w, s, b = _rand_layer(rng, N, C, K)
    xt, wt, st, bt = (_t(torch_dev, a) for a in (x, w, s, b))
    U = pkg.filter_transform_f2(wt)
    knobs.unset("WINO_SK_GRID")
    ref = pkg.conv3x3_bn_relu(xt, U, bt, st).clone()
    want = O.conv3x3_bn_relu_direct(x[:2], w, s, b)
    assert O.rel_error(ref[:2].cpu().numpy(), want) < TIGHT
    scale = float(ref.abs().max())
    for grid in (8, 24, 64, 136, 200, 256, 392, 2048):
        knobs.set("WINO_SK_GRID", str(grid))
        a = pkg.conv3x3_bn_relu(xt, U, bt, st).clone()
        c = pkg.conv3x3_bn_relu(xt, U, bt, st)
        assert torch.equal(a, c), f"grid {grid}: not reproducible"
        assert float((a - ref).abs().max()) < 2e-6 * scale, f"grid {grid}"
        assert (a.cpu().numpy()[:, _ring(), :] == 0).all()


@pytest.mark.parametrize("N,C,K,grids", [(128, 256, 256, (257, 300, 333, 391, 400)), (128, 64, 256, (300, 391)),
                                         (100, 128, 128, (257, 290))])
def test_conv3x3_late_workgroups_finish_their_items(N, C, K, grids, pkg, torch_dev, knobs):
    """Whole-item rounds + a stream-K tail on a grid LARGER than the CU count: the workgroups beyond the
    first wave of residents start when others have finished, so the order in which an item's tickets are
    drawn is the opposite of the usual one.  A range of such a launch holds up to three segments -- the end
    of one tail item, the start of the next (both partial, both with deferred tickets), then whole items
    -- and round 1's kernel dropped the first deferred ticket's result when the second was armed: with the
    usual order somebody else draws last and nothing shows; with late workgroups the item was never
    finalized and its counters stayed non-zero, which corrupted every later launch on the stream (found by
    tools/soak.py with a second stream competing for the CUs).  Here: NaN-filled outputs, every element
    against the automatic grid's result and (sampled images) the direct comparator, counters back at zero
    after every launch, and the automatic grid still right afterwards."""
    torch, dev = torch_dev
    knobs.set("WINO_3X3_ALGO", "big")
    g = torch.Generator(device="cpu").manual_seed(N + C)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    x, w, s, b = mk(N, 16, 16, C), mk(K, C, 3, 3), mk(K), mk(K)
    U = pkg.filter_transform_f2(w)
    knobs.unset("WINO_SK_GRID")
    ref = pkg.conv3x3_bn_relu(x, U, b, s).clone()
    assert pkg.tickets_in_use() == 0
    direct = pkg.conv3x3_direct(x[:3], w, b, s)
    scale = float(ref.abs().max())
    assert float((ref[:3] - direct).abs().max()) < TIGHT * scale
    for grid in grids:
        knobs.set("WINO_SK_GRID", str(grid))
        for rep in range(3):
            out = torch.full((N, 16, 16, K), float("nan"), device=dev)
            pkg.conv3x3_bn_relu(x, U, b, s, out=out)
            assert pkg.tickets_in_use() == 0, f"grid {grid}: an item was left unfinished"
            assert not bool(torch.isnan(out).any()), f"grid {grid}: outputs never written"
            assert float((out - ref).abs().max()) < 2e-6 * scale, f"grid {grid}"
    knobs.unset("WINO_SK_GRID")
    assert torch.equal(pkg.conv3x3_bn_relu(x, U, b, s), ref)


def test_conv3x3_streamk_with_a_competing_stream(pkg, torch_dev, knobs):
    """The same hazard without a forced grid: a second stream's launches occupy CUs, so part of the
    headline launch's 256 workgroups start late.  Results must not move (bitwise), counters must be zero."""
    torch, dev = torch_dev
    knobs.set("WINO_3X3_ALGO", "big")
    g = torch.Generator(device="cpu").manual_seed(11)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    x, w, s, b = mk(128, 16, 16, 256), mk(256, 256, 3, 3), mk(256), mk(256)
    xs, ws, vs = mk(64, 16, 16, 128), mk(128, 128, 3, 3), mk(128)
    U, Us = pkg.filter_transform_f2(w), pkg.filter_transform_f2(ws)
    ref = pkg.conv3x3_bn_relu(x, U, b, s).clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(40):
        with torch.cuda.stream(side):
            for _ in range(3):
                pkg.conv3x3_bn_relu(xs, Us, vs, vs)
        outs = [pkg.conv3x3_bn_relu(x, U, b, s) for _ in range(5)]
        for o in outs:
            assert torch.equal(o, ref), rep
    torch.cuda.synchronize()
    assert pkg.tickets_in_use() == 0


def test_conv1x1_streamk_with_a_competing_stream(pkg, torch_dev, knobs):
    """The 1x1 kernel's stream-K and split-K forms while a second stream's launches hold CUs (late and
    non-resident workgroups): results must not move bitwise, counters must return to zero.  Automatic forms at
    the reference's 1024->256 (stream-K at N = 128, split-K at N = 2) and a forced grid beyond the resident
    capacity (two 8-wave workgroups per CU)."""
    torch, dev = torch_dev
    knobs.set("WINO_1X1_ALGO", "big")   # the tiled kernel's hand-over forms are what is tested (N = 2 would take the latency form)
    g = torch.Generator(device="cpu").manual_seed(23)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    xs, ws, vs = mk(64, 16, 16, 128), mk(128, 128, 3, 3), mk(128)
    Us = pkg.filter_transform_f2(ws)
    side = torch.cuda.Stream()
    for (N, Cin, Kout, grid) in [(128, 1024, 256, None), (2, 1024, 256, None), (100, 512, 128, "1024"), (60, 2048, 512, "768")]:
        A, Bm, b, s = mk(N * 196, Cin), mk(Cin, Kout), mk(Kout), mk(Kout)
        if grid:
            knobs.set("WINO_1X1_SK", "1"); knobs.set("WINO_1X1_SK_GRID", grid)
        try:
            ref = pkg.conv1x1_bn(A, Bm, b, s, True).clone()
            want = torch.relu((A.double() @ Bm.double()) * s.double() + b.double())
            assert float((ref.double() - want).abs().max() / want.abs().max()) < TIGHT, (N, Cin, Kout, grid)
            torch.cuda.synchronize()
            for rep in range(15):
                with torch.cuda.stream(side):
                    for _ in range(3):
                        pkg.conv3x3_bn_relu(xs, Us, vs, vs)
                for o in [pkg.conv1x1_bn(A, Bm, b, s, True) for _ in range(4)]:
                    assert torch.equal(o, ref), (N, Cin, Kout, grid, rep)
            torch.cuda.synchronize()
            assert pkg.tickets_in_use() == 0, (N, Cin, Kout, grid)
        finally:
            knobs.unset("WINO_1X1_SK"); knobs.unset("WINO_1X1_SK_GRID")


@pytest.mark.parametrize("N,H,W,C,K", [(3, 28, 28, 128, 128), (2, 56, 56, 64, 64), (5, 8, 12, 16, 64),
                                       (7, 2, 2, 8, 64), (2, 30, 6, 24, 192), (64, 28, 28, 128, 128),
                                       (9, 7, 7, 512, 512), (4, 1, 1, 8, 64), (3, 5, 8, 16, 64), (2, 13, 3, 40, 128)])
def test_conv3x3_other_feature_maps(N, H, W, C, K, pkg, O, torch_dev):
    """the reference hard-codes ResNet's 14x14 stage; the same kernel
    with the geometry in its arguments covers any H x W (the 56x56, 28x28 and 7x7 stages, odd
    sizes whose last tile row / column is clipped, odd aspect ratios, a single tile or pixel).  Oracle on every element for the small cases, oracle on a sample +
    the direct GPU comparator on every element for the large one; zero ring."""
    torch, dev = torch_dev
    rng = np.random.RandomState(H * 131 + W)
    x = (rng.rand(N, H + 2, W + 2, C) - 0.5).astype(np.float32)
    w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    s = (rng.rand(K) - 0.5).astype(np.float32)
    b = (rng.rand(K) - 0.5).astype(np.float32)
    xt, wt, st, bt = (_t(torch_dev, a) for a in (x, w, s, b))
    U = pkg.filter_transform_f2(wt)
    got_t = pkg.conv3x3_bn_relu(xt, U, bt, st)
    again = pkg.conv3x3_bn_relu(xt, U, bt, st)
    assert torch.equal(got_t, again)
    got = got_t.cpu().numpy()
    idx = list(range(N)) if N <= 8 else [0, N // 2, N - 1]
    assert O.rel_error(got[idx], O.conv3x3_bn_relu_direct(x[idx], w, s, b)) < TIGHT
    assert O.rel_error(got, pkg.conv3x3_direct(xt, wt, bt, st).cpu().numpy()) < TIGHT
    ring = np.ones((H + 2, W + 2), bool)
    ring[1:H + 1, 1:W + 1] = False
    assert (got[:, ring, :] == 0).all()


def test_conv3x3_streams_and_graph(pkg, O, torch_dev):
    """The stream-K scratch is library-owned, one per (device, stream): launches on two streams
    must not disturb each other, and after wino_conv3x3_prepare() the launch can be captured into a
    HIP graph (no allocation inside the capture) and replayed."""
    torch, dev = torch_dev
    g = torch.Generator(device="cpu").manual_seed(11)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    xa, xb = mk(96, 16, 16, 256), mk(72, 16, 16, 256)
    w, s, b = mk(256, 256, 3, 3), mk(256), mk(256)
    U = pkg.filter_transform_f2(w)
    ref_a = pkg.conv3x3_bn_relu(xa, U, b, s).clone()
    ref_b = pkg.conv3x3_bn_relu(xb, U, b, s).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(8):   # interleaved on two streams, both shapes cut items (stream-K tails)
        with torch.cuda.stream(sa):
            outs_a.append(pkg.conv3x3_bn_relu(xa, U, b, s))
        with torch.cuda.stream(sb):
            outs_b.append(pkg.conv3x3_bn_relu(xb, U, b, s))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref_a) for o in outs_a)
    assert all(torch.equal(o, ref_b) for o in outs_b)
    # graph capture on a side stream
    sg = torch.cuda.Stream()
    out = torch.empty_like(ref_a)
    with torch.cuda.stream(sg):
        pkg.conv3x3_prepare(96, 256, 256)
        pkg.conv3x3_bn_relu(xa, U, b, s, out=out)   # warm (kernel attributes) outside the capture
    sg.synchronize()
    graph = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(graph, stream=sg):
        pkg.conv3x3_bn_relu(xa, U, b, s, out=out)
    for _ in range(3):
        out.fill_(7.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref_a)


def test_conv3x3_config2_128(pkg, O, torch_dev):
    """BASELINE configs[1]: 128->128, N=128, against the comparator + oracle sample."""
    rng = np.random.RandomState(43)
    N, C, K = 128, 128, 128
    x, w, s, b = _rand_layer(rng, N, C, K)
    xt, wt, st, bt = (_t(torch_dev, a) for a in (x, w, s, b))
    got = pkg.conv3x3_bn_relu(xt, pkg.filter_transform_f2(wt), bt, st).cpu().numpy()
    assert O.rel_error(got, pkg.conv3x3_direct(xt, wt, bt, st).cpu().numpy()) < TIGHT
    idx = [0, 31, 127]
    assert O.rel_error(got[idx], O.conv3x3_bn_relu_direct(x[idx], w, s, b)) < TIGHT


def test_comparator_is_independent_and_correct(pkg, O, torch_dev):
    rng = np.random.RandomState(5)
    x, w, s, b = _rand_layer(rng, 2, 32, 64)
    got = pkg.conv3x3_direct(*(_t(torch_dev, a) for a in (x, w, b, s))).cpu().numpy()
    assert O.rel_error(got, O.conv3x3_bn_relu_direct(x, w, s, b)) < TIGHT


@pytest.mark.parametrize("C", [128, 256])
def test_comparator_3x3_matches_golden(C, data_dir, pkg, O, golden_outputs, torch_dev):
    """The comparator kernels stand in for cuDNN in ./Test's second timer and diff; they are checked
    against the oracle at the reference's own Test-0 / Test-1 layers (golden vectors), so that
    "agrees with the comparator" elsewhere in this suite means something."""
    x = load_bin(data_dir, f"input_14_1_{C}.bin").reshape(1, 16, 16, C)
    w = load_bin(data_dir, f"weight_NCHW_{C}_{C}.bin").reshape(C, C, 3, 3)
    s = load_bin(data_dir, f"bnScale_winograd_{C}.bin")
    b = load_bin(data_dir, f"bnBias_winograd_{C}.bin")
    got = pkg.conv3x3_direct(*(_t(torch_dev, a) for a in (x, w, b, s))).cpu().numpy()
    want = golden_outputs["kernel_128" if C == 128 else "kernel_256"]
    assert O.rel_error(got[:, 1:15, 1:15, :].reshape(want.shape), want) < TIGHT
    assert (got[:, _ring(), :] == 0).all()


@pytest.mark.parametrize("name", ["kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in", "kernel_256_1_out"])
def test_comparator_1x1_matches_golden(name, data_dir, pkg, O, golden_outputs, torch_dev):
    Cin, Kout, relu = O.ONE_BY_ONE_LAYERS[name]
    A = load_bin(data_dir, "input_one_14_1024.bin", 196 * Cin).reshape(196, Cin)
    B = load_bin(data_dir, "weight_one_1024.bin", Cin * Kout).reshape(Cin, Kout)
    s = load_bin(data_dir, "bnScale_myKernel_one_1024.bin", Kout)
    b = load_bin(data_dir, "bnBias_myKernel_one_1024.bin", Kout)
    got = pkg.conv1x1_direct(_t(torch_dev, A), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s), relu)
    assert O.rel_error(got.cpu().numpy(), golden_outputs[name]) < TIGHT


# ------------------------------------------------------------------ 1x1 layers
@pytest.mark.parametrize("name", ["kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in", "kernel_256_1_out"])
def test_one_by_one_golden(name, data_dir, pkg, O, golden_outputs, torch_dev):
    Cin, Kout, relu = O.ONE_BY_ONE_LAYERS[name]
    A = load_bin(data_dir, "input_one_14_1024.bin", 196 * Cin).reshape(196, Cin)
    B = load_bin(data_dir, "weight_one_1024.bin", Cin * Kout).reshape(Cin, Kout)
    s = load_bin(data_dir, "bnScale_myKernel_one_1024.bin", Kout)
    b = load_bin(data_dir, "bnBias_myKernel_one_1024.bin", Kout)
    got = pkg.conv1x1_bn(_t(torch_dev, A), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s), relu)
    assert O.rel_error(got.cpu().numpy(), golden_outputs[name]) < TIGHT


@pytest.mark.parametrize("M", [1, 15, 111, 112, 113, 197, 1000])
def test_one_by_one_ragged_rows(M, pkg, O, torch_dev):
    rng = np.random.RandomState(M)
    Cin, Kout = 64, 128
    A = ((rng.rand(M, Cin) - 0.5) * 40).astype(np.float32)
    B = ((rng.rand(Cin, Kout) - 0.5) * 40).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = (rng.rand(Kout) - 0.5).astype(np.float32)
    for relu in (True, False):
        got = pkg.conv1x1_bn(_t(torch_dev, A), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s), relu)
        assert O.rel_error(got.cpu().numpy(), O.conv1x1_bn(A, B, b, s, relu)) < TIGHT


@pytest.mark.parametrize("M,Cin,Kout", [(700, 256, 192), (333, 512, 320), (1500, 160, 448), (112, 1024, 64)])
def test_one_by_one_column_counts(M, Cin, Kout, pkg, O, torch_dev):
    """Kout is any multiple of 64: with a large Cin the 128-column workgroup form would leave the last
    64 columns of Kout = 192, 320, 448 uncomputed.  Every column against the fp64 oracle, on an
    output buffer pre-filled with NaN."""
    torch, dev = torch_dev
    rng = np.random.RandomState(Kout)
    A = ((rng.rand(M, Cin) - 0.5) * 4).astype(np.float32)
    B = ((rng.rand(Cin, Kout) - 0.5) * 4).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = ((rng.rand(Kout) - 0.5) * 4).astype(np.float32)
    out = torch.full((M, Kout), float("nan"), device=dev)
    pkg.conv1x1_bn(_t(torch_dev, A), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s), False, out=out)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert O.rel_error(got, O.conv1x1_bn(A, B, b, s, False)) < TIGHT


@pytest.mark.parametrize("name", ["kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in", "kernel_256_1_out"])
def test_one_by_one_full_size(name, pkg, O, torch_dev):
    """BASELINE configs[3]: N = 128 -> M = 25088 rows, against the fp64 GEMM oracle."""
    Cin, Kout, relu = O.ONE_BY_ONE_LAYERS[name]
    rng = np.random.RandomState(9)
    M = 128 * 196
    A = ((rng.rand(M, Cin) - 0.5) * 40).astype(np.float32)
    B = ((rng.rand(Cin, Kout) - 0.5) * 40).astype(np.float32)
    s = ((rng.rand(Kout) - 0.5)).astype(np.float32)
    b = ((rng.rand(Kout) - 0.5) * 40).astype(np.float32)
    At, Bt, bt, st = (_t(torch_dev, a) for a in (A, B, b, s))
    got = pkg.conv1x1_bn(At, Bt, bt, st, relu).cpu().numpy()
    assert O.rel_error(got, O.conv1x1_bn(A, B, b, s, relu)) < TIGHT
    assert O.rel_error(got, pkg.conv1x1_direct(At, Bt, bt, st, relu).cpu().numpy()) < TIGHT


@pytest.mark.parametrize("M,Cin,Kout", [(196, 1024, 256), (1000, 512, 128), (3 * 196, 256, 1024), (2500, 128, 512), (113, 64, 64), (700, 256, 192)])
def test_one_by_one_streamk_decompositions_agree(M, Cin, Kout, pkg, O, torch_dev, knobs):
    """The 1x1 kernel's stream-K form cuts tiles x k-steps into G equal ranges; a tile cut by a range
    boundary is finished by whichever workgroup draws its last ticket, which adds the segments'
    slabs in k order.  Every G must agree with the fp64 oracle and with the plain launch (to
    rounding: the cuts move the order of the channel sum), every G must be bitwise reproducible,
    and the tile counters must be back at zero (the second launch of a pair relies on it).  Small
    G gives long ranges (whole tiles in the middle), large G tiles cut into many segments."""
    torch, dev = torch_dev
    rng = np.random.RandomState(M + Cin)
    A = ((rng.rand(M, Cin) - 0.5) * 4).astype(np.float32)
    B = ((rng.rand(Cin, Kout) - 0.5) * 4).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = ((rng.rand(Kout) - 0.5) * 4).astype(np.float32)
    At, Bt, bt, st = (_t(torch_dev, a) for a in (A, B, b, s))
    want = O.conv1x1_bn(A, B, b, s, True)
    knobs.set("WINO_1X1_SK", "0")
    knobs.unset("WINO_1X1_SK_GRID")
    plain = pkg.conv1x1_bn(At, Bt, bt, st, True).clone()
    assert O.rel_error(plain.cpu().numpy(), want) < TIGHT
    scale = float(plain.abs().max())
    knobs.set("WINO_1X1_SK", "1")
    for grid in (None, 8, 16, 40, 104, 256, 512, 1000, 4096):
        if grid is None:
            knobs.unset("WINO_1X1_SK_GRID")
        else:
            knobs.set("WINO_1X1_SK_GRID", str(grid))
        a = pkg.conv1x1_bn(At, Bt, bt, st, True).clone()
        c = pkg.conv1x1_bn(At, Bt, bt, st, True)
        assert torch.equal(a, c), f"grid {grid}: not reproducible"
        assert O.rel_error(a.cpu().numpy(), want) < TIGHT, f"grid {grid}"
        assert float((a - plain).abs().max()) < 4e-6 * scale, f"grid {grid}"


def test_one_by_one_streamk_flags_and_graph(pkg, O, torch_dev, knobs):
    """Stream-K form under the chaining flags (padded A, padded C with its zero ring, residual before
    the ReLU), then captured into a HIP graph after wino_conv1x1_prepare (no allocation inside the
    capture) and replayed."""
    torch, dev = torch_dev
    rng = np.random.RandomState(77)
    N, Cin, Kout = 6, 512, 128
    A = (rng.rand(N, 14, 14, Cin) - 0.5).astype(np.float32)
    B = (rng.rand(Cin, Kout) - 0.5).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = (rng.rand(Kout) - 0.5).astype(np.float32)
    R = (rng.rand(N * 196, Kout) - 0.5).astype(np.float32)
    Ap = rng.rand(N, 16, 16, Cin).astype(np.float32) * 100
    Ap[:, 1:15, 1:15, :] = A
    want = O.conv1x1_bn(A.reshape(-1, Cin), B, b, s, True)
    want_res = np.maximum(O.conv1x1_bn(A.reshape(-1, Cin), B, b, s, False) + R, 0)
    knobs.set("WINO_1X1_SK", "1")
    for grid in ("24", "200"):
        knobs.set("WINO_1X1_SK_GRID", grid)
        out = torch.full((N, 16, 16, Kout), float("nan"), device=dev)
        pkg.conv1x1_bn_ex(_t(torch_dev, Ap), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s),
                          pkg.RELU | pkg.A_PADDED | pkg.C_PADDED, out=out)
        got = out.cpu().numpy()
        assert O.rel_error(got[:, 1:15, 1:15, :].reshape(-1, Kout), want) < TIGHT
        assert (got[:, _ring(), :] == 0).all()
        got3 = pkg.conv1x1_bn_ex(_t(torch_dev, A), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s),
                                 pkg.RELU | pkg.ADD_RESIDUAL, residual=_t(torch_dev, R)).cpu().numpy()
        assert O.rel_error(got3, want_res) < TIGHT
    # graph capture on a side stream: scratch allocated by prepare, outside the capture
    knobs.set("WINO_1X1_SK_GRID", "200")
    At, Bt, bt, st = (_t(torch_dev, a) for a in (A.reshape(-1, Cin), B, b, s))
    outg = torch.zeros(N * 196, Kout, device=dev)
    sg = torch.cuda.Stream()
    with torch.cuda.stream(sg):
        pkg.conv1x1_prepare(N * 196, Cin, Kout)
    sg.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=sg):
        pkg.conv1x1_bn(At, Bt, bt, st, True, out=outg)
    for _ in range(3):
        outg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert O.rel_error(outg.cpu().numpy(), want) < TIGHT


def test_one_by_one_activations_beyond_4gib(pkg, O, torch_dev):
    """The 1x1 kernel addresses A through a buffer descriptor (32-bit byte offsets) that is re-based
    at every tile's first row, so A itself may be larger than 4 GiB: 1.1 M rows x 1024 channels =
    4.5 GB here.  Rows on both sides of the 4 GiB boundary (row 2^32 / 4096 = 1 048 576), the first
    and the last tile against the fp64 oracle; every output finite."""
    torch, dev = torch_dev
    M, Cin, Kout = 1_100_000, 1024, 64
    free, _ = torch.cuda.mem_get_info()
    if free < 8 * (1 << 30):
        pytest.skip("needs 8 GiB of free device memory")
    g = torch.Generator(device=dev).manual_seed(4)
    A = torch.rand(M, Cin, device=dev, generator=g) - 0.5
    rng = np.random.RandomState(4)
    B = (rng.rand(Cin, Kout) - 0.5).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = (rng.rand(Kout) - 0.5).astype(np.float32)
    out = torch.full((M, Kout), float("nan"), device=dev)
    pkg.conv1x1_bn(A, _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s), False, out=out)
    assert bool(torch.isfinite(out).all())
    boundary = (1 << 32) // (Cin * 4)
    rows = np.unique(np.concatenate([np.arange(0, 224), np.arange(boundary - 300, boundary + 300),
                                     np.arange(M - 224, M), rng.randint(0, M, 500)]))
    idx = torch.as_tensor(rows, device=dev)
    want = O.conv1x1_bn(A[idx].cpu().numpy(), B, b, s, False)
    assert O.rel_error(out[idx].cpu().numpy(), want) < TIGHT
    del A, out
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ chaining
def test_conv1x1_padded_in_out_and_residual(pkg, O, torch_dev):
    torch, dev = torch_dev
    rng = np.random.RandomState(21)
    N, Cin, Kout = 3, 64, 128
    A = (rng.rand(N, 14, 14, Cin) - 0.5).astype(np.float32)
    B = (rng.rand(Cin, Kout) - 0.5).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = (rng.rand(Kout) - 0.5).astype(np.float32)
    R = (rng.rand(N * 196, Kout) - 0.5).astype(np.float32)
    want = O.conv1x1_bn(A.reshape(-1, Cin), B, b, s, True)
    # C_PADDED: result in the interior of [N][16][16][Kout], ring written as exact zeros
    out = torch.full((N, 16, 16, Kout), float("nan"), device=dev)
    pkg.conv1x1_bn_ex(_t(torch_dev, A), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s),
                      pkg.RELU | pkg.C_PADDED, out=out)
    got = out.cpu().numpy()
    assert O.rel_error(got[:, 1:15, 1:15, :].reshape(-1, Kout), want) < TIGHT
    assert (got[:, _ring(), :] == 0).all()
    # A_PADDED: read the interior of a padded tensor whose ring holds garbage that must be ignored
    Ap = rng.rand(N, 16, 16, Cin).astype(np.float32) * 100
    Ap[:, 1:15, 1:15, :] = A
    got2 = pkg.conv1x1_bn_ex(_t(torch_dev, Ap), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s),
                             pkg.RELU | pkg.A_PADDED).cpu().numpy()
    assert O.rel_error(got2, want) < TIGHT
    # ADD_RESIDUAL before the ReLU
    want3 = np.maximum(O.conv1x1_bn(A.reshape(-1, Cin), B, b, s, False) + R, 0)
    got3 = pkg.conv1x1_bn_ex(_t(torch_dev, A), _t(torch_dev, B), _t(torch_dev, b), _t(torch_dev, s),
                             pkg.RELU | pkg.ADD_RESIDUAL, residual=_t(torch_dev, R)).cpu().numpy()
    assert O.rel_error(got3, want3) < TIGHT


@pytest.mark.parametrize("N,H,W,Cin,Kout", [(2, 28, 28, 64, 128), (1, 56, 56, 64, 64), (3, 7, 7, 512, 128), (2, 5, 9, 96, 192), (4, 1, 1, 32, 64)])
def test_conv1x1_padded_other_feature_maps(N, H, W, Cin, Kout, pkg, O, torch_dev):
    """the chaining layouts at any feature-map size (the reference's stage is
    14 x 14).  C_PADDED into a NaN-filled [N][H+2][W+2][Kout] (interior against the oracle, ring exact
    zeros), A_PADDED from a tensor whose ring holds garbage, residual added before the ReLU."""
    torch, dev = torch_dev
    rng = np.random.RandomState(H * 100 + W)
    A = (rng.rand(N, H, W, Cin) - 0.5).astype(np.float32)
    B = (rng.rand(Cin, Kout) - 0.5).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = (rng.rand(Kout) - 0.5).astype(np.float32)
    R = (rng.rand(N * H * W, Kout) - 0.5).astype(np.float32)
    t = lambda a: _t(torch_dev, a)
    want = O.conv1x1_bn(A.reshape(-1, Cin), B, b, s, True)
    out = torch.full((N, H + 2, W + 2, Kout), float("nan"), device=dev)
    pkg.conv1x1_bn_ex(t(A), t(B), t(b), t(s), pkg.RELU | pkg.C_PADDED, out=out)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert O.rel_error(got[:, 1:-1, 1:-1, :].reshape(-1, Kout), want) < TIGHT
    ring = np.ones((H + 2, W + 2), bool)
    ring[1:-1, 1:-1] = False
    assert (got[:, ring, :] == 0).all()
    Ap = rng.rand(N, H + 2, W + 2, Cin).astype(np.float32) * 100
    Ap[:, 1:-1, 1:-1, :] = A
    got2 = pkg.conv1x1_bn_ex(t(Ap), t(B), t(b), t(s), pkg.RELU | pkg.A_PADDED).cpu().numpy()
    assert got2.shape == (N * H * W, Kout)
    assert O.rel_error(got2, want) < TIGHT
    want3 = np.maximum(O.conv1x1_bn(A.reshape(-1, Cin), B, b, s, False) + R, 0)
    got3 = pkg.conv1x1_bn_ex(t(Ap), t(B), t(b), t(s), pkg.RELU | pkg.A_PADDED | pkg.ADD_RESIDUAL,
                             residual=t(R)).cpu().numpy()
    assert O.rel_error(got3, want3) < TIGHT


@pytest.mark.parametrize("N,H,W,C4,Cm", [(2, 28, 28, 512, 128), (1, 56, 56, 256, 64), (3, 7, 7, 2048, 512), (2, 9, 5, 256, 64)])
def test_residual_block_other_feature_maps(N, H, W, C4, Cm, pkg, O, torch_dev):
    """The bottleneck block at ResNet's other stages (56x56x256/64, 28x28x512/128, 7x7x2048/512) and
    an odd size: three launches chained through padded [N][H+2][W+2][Cm] intermediates, against the
    fp64 composition of the layer oracles."""
    rng = np.random.RandomState(H + 7 * W)
    x = (rng.rand(N, H, W, C4) - 0.5).astype(np.float32)
    w1 = ((rng.rand(C4, Cm) - 0.5) / np.sqrt(C4) * 4).astype(np.float32)
    w2 = ((rng.rand(Cm, Cm, 3, 3) - 0.5) / np.sqrt(9 * Cm) * 4).astype(np.float32)
    w3 = ((rng.rand(Cm, C4) - 0.5) / np.sqrt(Cm) * 4).astype(np.float32)
    bn = [((rng.rand(c) - 0.5).astype(np.float32), (rng.rand(c) + 0.5).astype(np.float32)) for c in (Cm, Cm, C4)]
    want = O.residual_block(x, w1, bn[0], w2, bn[1], w3, bn[2])
    t = lambda a: _t(torch_dev, a)
    U2 = pkg.filter_transform_f2(t(w2))
    got = pkg.residual_block(t(x), t(w1), (t(bn[0][0]), t(bn[0][1])), U2, (t(bn[1][0]), t(bn[1][1])),
                             t(w3), (t(bn[2][0]), t(bn[2][1]))).cpu().numpy()
    assert got.shape == want.shape
    assert O.rel_error(got, want) < TIGHT
    assert (want > 0).mean() > 0.2


@pytest.mark.parametrize("N,C4,Cm", [(2, 256, 128), (5, 1024, 256), (3, 384, 192)])
def test_residual_block(N, C4, Cm, pkg, O, torch_dev):
    """BASELINE configs[4]: 1x1 -> 3x3 -> 1x1 + skip, against the fp64 composition of the layer
    oracles.  Weights are scaled so that activations stay O(1) through the block."""
    rng = np.random.RandomState(33)
    x = (rng.rand(N, 14, 14, C4) - 0.5).astype(np.float32)
    w1 = ((rng.rand(C4, Cm) - 0.5) / np.sqrt(C4) * 4).astype(np.float32)
    w2 = ((rng.rand(Cm, Cm, 3, 3) - 0.5) / np.sqrt(9 * Cm) * 4).astype(np.float32)
    w3 = ((rng.rand(Cm, C4) - 0.5) / np.sqrt(Cm) * 4).astype(np.float32)
    bn = [((rng.rand(c) - 0.5).astype(np.float32), (rng.rand(c) + 0.5).astype(np.float32)) for c in (Cm, Cm, C4)]
    want = O.residual_block(x, w1, bn[0], w2, bn[1], w3, bn[2])
    t = lambda a: _t(torch_dev, a)
    U2 = pkg.filter_transform_f2(t(w2))
    got = pkg.residual_block(t(x), t(w1), (t(bn[0][0]), t(bn[0][1])), U2, (t(bn[1][0]), t(bn[1][1])),
                             t(w3), (t(bn[2][0]), t(bn[2][1]))).cpu().numpy()
    assert got.shape == want.shape
    assert O.rel_error(got, want) < TIGHT
    assert (want > 0).mean() > 0.2  # the test exercises both sides of the final ReLU


def test_residual_block_config5_per_gpu_share(pkg, O, torch_dev):
    """BASELINE configs[4] as one GPU of eight sees it: the 1024 -> 256 -> 256 -> 1024 bottleneck at
    N = 128, where the three launches take the forms they take in bench.py (1x1 1024->256 stream-K,
    3x3 whole-item rounds + stream-K tail, 1x1 256->1024 plain) -- each form is tested alone at N = 128
    elsewhere, this is the chain.  Sampled images against the fp64 composition of the layer oracles,
    every element against a chain of the comparator kernels."""
    torch, dev = torch_dev
    rng = np.random.RandomState(128)
    N, C4, Cm = 128, 1024, 256
    x = (rng.rand(N, 14, 14, C4) - 0.5).astype(np.float32)
    w1 = ((rng.rand(C4, Cm) - 0.5) / np.sqrt(C4) * 4).astype(np.float32)
    w2 = ((rng.rand(Cm, Cm, 3, 3) - 0.5) / np.sqrt(9 * Cm) * 4).astype(np.float32)
    w3 = ((rng.rand(Cm, C4) - 0.5) / np.sqrt(Cm) * 4).astype(np.float32)
    bn = [((rng.rand(c) - 0.5).astype(np.float32), (rng.rand(c) + 0.5).astype(np.float32)) for c in (Cm, Cm, C4)]
    t = lambda a: _t(torch_dev, a)
    bnt = [(t(a), t(b)) for a, b in bn]
    xt, w1t, w2t, w3t = t(x), t(w1), t(w2), t(w3)
    out = torch.full((N, 14, 14, C4), float("nan"), device=dev)
    pkg.residual_block(xt, w1t, bnt[0], pkg.filter_transform_f2(w2t), bnt[1], w3t, bnt[2], out=out)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    idx = [0, 63, 127]
    want = O.residual_block(x[idx], w1, bn[0], w2, bn[1], w3, bn[2])
    assert O.rel_error(got[idx], want) < TIGHT
    assert (want > 0).mean() > 0.2
    # every element: the same block from the comparator kernels (direct 1x1, direct 3x3, direct 1x1) + skip
    t1 = pkg.conv1x1_direct(xt.reshape(-1, C4), w1t, bnt[0][0], bnt[0][1], True)
    t1p = torch.zeros((N, 16, 16, Cm), device=dev)
    t1p[:, 1:15, 1:15, :] = t1.reshape(N, 14, 14, Cm)
    t2p = pkg.conv3x3_direct(t1p, w2t, bnt[1][0], bnt[1][1], True)
    t2 = t2p[:, 1:15, 1:15, :].reshape(-1, Cm)
    t3 = pkg.conv1x1_direct(t2, w3t, bnt[2][0], bnt[2][1], False)
    chain = torch.relu(t3.reshape(N, 14, 14, C4) + xt).cpu().numpy()
    assert O.rel_error(got, chain) < TIGHT


def test_conv3x3_batch_beyond_one_launch(pkg, O, torch_dev):
    """One launch addresses its tensors with 32-bit byte offsets (< 4 GiB each); the entry point takes
    any batch and cuts larger ones into launches of whole images.  5100 images of 56x56x64 are 4.4 GB
    in and 4.4 GB out: 4928 + 172.  The first and last image and the two on either side of the cut
    against the fp64 oracle, zero ring, every output finite."""
    torch, dev = torch_dev
    N, H, W, C, K = 5100, 56, 56, 64, 64
    free, _ = torch.cuda.mem_get_info()
    if free < 12 * (1 << 30):
        pytest.skip("needs 12 GiB of free device memory")
    per_image = (H + 2) * (W + 2) * max(C, K) * 4
    limit = ((1 << 32) - 1) // per_image
    cut = limit - limit % 64
    assert cut < N <= 2 * cut
    g = torch.Generator(device=dev).manual_seed(6)
    x = torch.rand(N, H + 2, W + 2, C, device=dev, generator=g) - 0.5
    rng = np.random.RandomState(6)
    w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    s = (rng.rand(K) - 0.5).astype(np.float32)
    b = (rng.rand(K) - 0.5).astype(np.float32)
    U = pkg.filter_transform_f2(_t(torch_dev, w))
    out = torch.full((N, H + 2, W + 2, K), float("nan"), device=dev)
    pkg.conv3x3_bn_relu(x, U, _t(torch_dev, b), _t(torch_dev, s), relu=True, out=out)
    assert bool(torch.isfinite(out).all())
    ring = np.ones((H + 2, W + 2), bool)
    ring[1:-1, 1:-1] = False
    for n in (0, cut - 1, cut, N - 1):
        got = out[n:n + 1].cpu().numpy()
        want = O.conv3x3_bn_relu_direct(x[n:n + 1].cpu().numpy(), w, s, b, relu=True)
        assert O.rel_error(got, want) < TIGHT, n
        assert (got[:, ring, :] == 0).all(), n
    del x, out
    torch.cuda.empty_cache()


def test_streamk_scratch_is_never_stale(pkg, O, torch_dev, knobs):
    """The hand-over slabs are the same memory in every launch, written by one workgroup and read
    by another, possibly on another XCD.  Launches that repeat the same inputs cannot tell a fresh
    slab from a line cached by an earlier launch; two different inputs alternated can: each result
    must equal the one its own input gives in the plain (1x1) or whole-item (3x3) form, to summation
    order, and must be bitwise what the same input gave the time before."""
    torch, dev = torch_dev
    rng = np.random.RandomState(314)
    t = lambda a: _t(torch_dev, a)
    # 1x1: many partial tiles per launch
    M, Cin, Kout = 9 * 196, 512, 256
    Bm = t((rng.rand(Cin, Kout) - 0.5).astype(np.float32))
    sc, bi = t((rng.rand(Kout) - 0.5).astype(np.float32)), t((rng.rand(Kout) - 0.5).astype(np.float32))
    As = [t(((rng.rand(M, Cin) - 0.5) * s).astype(np.float32)) for s in (1.0, 37.0)]
    knobs.set("WINO_1X1_SK", "0")
    knobs.unset("WINO_1X1_SK_GRID")
    plain = [pkg.conv1x1_bn(a, Bm, bi, sc, False).clone() for a in As]
    knobs.set("WINO_1X1_SK", "1")
    knobs.set("WINO_1X1_SK_GRID", "120")
    first = [None, None]
    for rep in range(12):
        i = rep & 1
        got = pkg.conv1x1_bn(As[i], Bm, bi, sc, False)
        assert float((got - plain[i]).abs().max()) < 4e-6 * float(plain[i].abs().max()), (rep, i)
        if first[i] is None:
            first[i] = got.clone()
        assert torch.equal(got, first[i]), (rep, i)
    # 3x3: every item cut into segments
    N, C, K = 9, 64, 128
    knobs.set("WINO_3X3_ALGO", "big")
    w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    U = pkg.filter_transform_f2(t(w))
    s3, b3 = t((rng.rand(K) - 0.5).astype(np.float32)), t((rng.rand(K) - 0.5).astype(np.float32))
    xs = [t(((rng.rand(N, 16, 16, C) - 0.5) * s).astype(np.float32)) for s in (1.0, 53.0)]
    knobs.set("WINO_SK_GRID", "8")      # items / 8 whole-item rounds, short tail
    whole = [pkg.conv3x3_bn_relu(x, U, b3, s3, relu=False).clone() for x in xs]
    knobs.set("WINO_SK_GRID", "200")    # more workgroups than items: everything is tail
    first = [None, None]
    for rep in range(12):
        i = rep & 1
        got = pkg.conv3x3_bn_relu(xs[i], U, b3, s3, relu=False)
        assert float((got - whole[i]).abs().max()) < 4e-6 * float(whole[i].abs().max()), (rep, i)
        if first[i] is None:
            first[i] = got.clone()
        assert torch.equal(got, first[i]), (rep, i)


def test_streams_created_and_destroyed_do_not_leak_scratch(pkg, O, torch_dev, knobs):
    """The stream-K scratch is owned by the library per (device, stream); wino_stream_destroy gives
    it back.  Fifty short-lived C-ABI streams, a stream-K launch on each: results right, device
    memory where it was (one stream's scratch is 32 MiB; a leak would cost 1.6 GB)."""
    torch, dev = torch_dev
    L = pkg.lib()
    rng = np.random.RandomState(50)
    M, Cin, Kout = 5 * 196, 512, 128
    A = ((rng.rand(M, Cin) - 0.5) * 4).astype(np.float32)
    B = ((rng.rand(Cin, Kout) - 0.5) * 4).astype(np.float32)
    s = (rng.rand(Kout) - 0.5).astype(np.float32)
    b = (rng.rand(Kout) - 0.5).astype(np.float32)
    At, Bt, bt, st = (_t(torch_dev, a) for a in (A, B, b, s))
    out = torch.empty(M, Kout, device=dev)
    want = O.conv1x1_bn(A, B, b, s, True)
    knobs.set("WINO_1X1_SK", "1")
    knobs.set("WINO_1X1_SK_GRID", "64")
    torch.cuda.synchronize()
    free0 = None
    for i in range(50):
        h = ctypes.c_void_p()
        assert L.wino_stream_create(ctypes.byref(h)) == 0
        out.zero_()
        torch.cuda.synchronize()
        rc = L.wino_conv1x1_bn(At.data_ptr(), Bt.data_ptr(), bt.data_ptr(), st.data_ptr(), out.data_ptr(),
                               M, Cin, Kout, 1, h)
        assert rc == 0, L.wino_last_error_string()
        assert L.wino_stream_destroy(h) == 0      # waits for the launch
        if i in (0, 49):
            assert O.rel_error(out.cpu().numpy(), want) < TIGHT
        if i == 4:
            free0 = torch.cuda.mem_get_info()[0]
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 * (1 << 20), (free0, free1)


# ------------------------------------------------------------------ random legal shapes
def test_random_legal_shapes(pkg, O, torch_dev, knobs):
    """Seeded sweep over shapes the C-ABI declares legal (1x1: any M, Cin % 32, Kout % 64; 3x3: any
    N, H, W, C % 8, K % 64), each against the fp64 oracle on an output pre-filled with NaN: the
    corners between the hand-picked cases (a column count that is a multiple of 64 but not of
    128 went uncomputed until a sweep like this one).  Stream-K forms are forced on every other
    1x1 case so that both launch forms see odd shapes."""
    torch, dev = torch_dev
    rng = np.random.RandomState(2024)
    t = lambda a: _t(torch_dev, a)
    for i in range(24):
        M = int(rng.randint(1, 2600)); Cin = 32 * int(rng.randint(1, 20)); Kout = 64 * int(rng.randint(1, 11))
        A = ((rng.rand(M, Cin) - 0.5) * 4).astype(np.float32)
        B = ((rng.rand(Cin, Kout) - 0.5) * 4).astype(np.float32)
        s = (rng.rand(Kout) - 0.5).astype(np.float32)
        b = ((rng.rand(Kout) - 0.5) * 4).astype(np.float32)
        relu = bool(i & 1)
        if i % 2:
            knobs.set("WINO_1X1_SK", "1")
            knobs.set("WINO_1X1_SK_GRID", str(8 * int(rng.randint(1, 64))))
        try:
            out = torch.full((M, Kout), float("nan"), device=dev)
            pkg.conv1x1_bn(t(A), t(B), t(b), t(s), relu, out=out)
            got = out.cpu().numpy()
        finally:
            knobs.unset("WINO_1X1_SK")
            knobs.unset("WINO_1X1_SK_GRID")
        assert np.isfinite(got).all(), (M, Cin, Kout)
        assert O.rel_error(got, O.conv1x1_bn(A, B, b, s, relu)) < TIGHT, (M, Cin, Kout)
    for i in range(16):
        N = int(rng.randint(1, 6)); H = int(rng.randint(1, 21)); W = int(rng.randint(1, 21))
        C = 8 * int(rng.randint(1, 20)); K = 64 * int(rng.randint(1, 4))
        x = (rng.rand(N, H + 2, W + 2, C) - 0.5).astype(np.float32)
        w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
        s = (rng.rand(K) - 0.5).astype(np.float32)
        b = (rng.rand(K) - 0.5).astype(np.float32)
        relu = bool(i & 1)
        U = pkg.filter_transform_f2(t(w))
        out = torch.full((N, H + 2, W + 2, K), float("nan"), device=dev)
        pkg.conv3x3_bn_relu(t(x), U, t(b), t(s), relu=relu, out=out)
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), (N, H, W, C, K)
        want = O.conv3x3_bn_relu_direct(x, w, s, b, relu=relu)
        assert O.rel_error(got, want) < TIGHT, (N, H, W, C, K)
        ring = np.ones((H + 2, W + 2), bool); ring[1:-1, 1:-1] = False
        assert (got[:, ring, :] == 0).all(), (N, H, W, C, K)


def test_random_forced_grids_3x3(pkg, torch_dev, knobs):
    """Seeded sweep of the 3x3 throughput kernel's hand-off over random shapes AND random forced grids (1 ... 700
    logical workgroups: fewer than items, more than CUs, whole-item rounds with short and long tails, items cut
    into many segments), at the reference's 14x14 and at other feature maps: NaN-filled outputs against the direct
    comparator kernel, every launch twice (bitwise equal), ticket counters at zero after every launch."""
    torch, dev = torch_dev
    knobs.set("WINO_3X3_ALGO", "big")
    rng = np.random.RandomState(777)
    g = torch.Generator(device="cpu").manual_seed(777)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    for i in range(20):
        if i % 2 == 0:
            N, H, W = int(rng.randint(2, 90)), 14, 14
        else:
            N, H, W = int(rng.randint(1, 9)), int(rng.randint(3, 30)), int(rng.randint(3, 30))
        C = 8 * int(rng.randint(1, 24)); K = 64 * int(rng.randint(1, 4))
        grid = int(rng.randint(1, 700))
        x, w, sc, bi = mk(N, H + 2, W + 2, C), mk(K, C, 3, 3), mk(K), mk(K)
        U = pkg.filter_transform_f2(w)
        want = pkg.conv3x3_direct(x, w, bi, sc)
        scale = float(want.abs().max())
        knobs.set("WINO_SK_GRID", str(grid))
        outs = []
        for rep in range(2):
            out = torch.full((N, H + 2, W + 2, K), float("nan"), device=dev)
            pkg.conv3x3_bn_relu(x, U, bi, sc, out=out)
            assert pkg.tickets_in_use() == 0, (i, N, H, W, C, K, grid)
            outs.append(out)
        knobs.unset("WINO_SK_GRID")
        assert not bool(torch.isnan(outs[0]).any()), (i, N, H, W, C, K, grid)
        assert torch.equal(outs[0], outs[1]), (i, N, H, W, C, K, grid)
        assert float((outs[0] - want).abs().max()) < TIGHT * scale, (i, N, H, W, C, K, grid)


def test_residual_block_in_a_graph(pkg, O, torch_dev):
    """The three launches of the bottleneck block captured into one HIP graph: the prepare calls
    allocate every launch's stream-K scratch on the capture stream beforehand (an allocation inside
    a capture is an error), the replayed graph gives the eager result bit for bit and the oracle's
    to tolerance."""
    torch, dev = torch_dev
    rng = np.random.RandomState(808)
    N, C4, Cm = 20, 1024, 256      # 1x1 1024->256 at N = 20 takes its split-K form, the 3x3 its stream-K tail
    x = (rng.rand(N, 14, 14, C4) - 0.5).astype(np.float32)
    w1 = ((rng.rand(C4, Cm) - 0.5) / np.sqrt(C4) * 4).astype(np.float32)
    w2 = ((rng.rand(Cm, Cm, 3, 3) - 0.5) / np.sqrt(9 * Cm) * 4).astype(np.float32)
    w3 = ((rng.rand(Cm, C4) - 0.5) / np.sqrt(Cm) * 4).astype(np.float32)
    bn = [((rng.rand(c) - 0.5).astype(np.float32), (rng.rand(c) + 0.5).astype(np.float32)) for c in (Cm, Cm, C4)]
    want = O.residual_block(x, w1, bn[0], w2, bn[1], w3, bn[2])
    t = lambda a: _t(torch_dev, a)
    xt, w1t, w3t = t(x), t(w1), t(w3)
    U2 = pkg.filter_transform_f2(t(w2))
    bnt = [(t(b), t(s)) for b, s in bn]
    eager = pkg.residual_block(xt, w1t, bnt[0], U2, bnt[1], w3t, bnt[2]).clone()
    assert O.rel_error(eager.cpu().numpy(), want) < TIGHT
    out = torch.zeros_like(xt)
    ws = torch.empty(pkg.lib().wino_residual_block_workspace_bytes(N, Cm) // 4, device=dev)
    sg = torch.cuda.Stream()
    with torch.cuda.stream(sg):
        pkg.conv1x1_prepare(N * 196, C4, Cm)
        pkg.conv3x3_prepare(N, Cm, Cm, 14, 14)
        pkg.conv1x1_prepare(N * 196, Cm, C4)
    sg.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=sg):
        pkg.residual_block(xt, w1t, bnt[0], U2, bnt[1], w3t, bnt[2], out=out, workspace=ws)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)


# ------------------------------------------------------------------ errors
def test_bad_shapes_raise(pkg, torch_dev):
    torch, dev = torch_dev
    z = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(pkg.WinoError):
        pkg.conv3x3_bn_relu(z(1, 16, 16, 12), z(16 * 12 * 64), z(64), z(64))       # C % 8
    with pytest.raises(pkg.WinoError):
        pkg.conv3x3_bn_relu(z(1, 2, 14, 16), z(16 * 16 * 64), z(64), z(64))        # no room for one output row
    with pytest.raises(pkg.WinoError):
        pkg.conv1x1_bn(z(8, 48), z(48, 128), z(128), z(128), True)                 # Cin % 32


def test_misaligned_tensor_pointers_are_refused(pkg, torch_dev):
    """The kernels move 16 bytes per lane: a tensor pointer that is not 16-byte aligned (a view one float into a
    buffer) is refused with WINO_E_ARG at the C-ABI instead of being handed to the loads."""
    torch, dev = torch_dev
    z = lambda n: torch.zeros(n, device=dev)
    x, U, b, s, out = z(16 * 16 * 64 + 4), z(16 * 64 * 64 + 4), z(64), z(64), z(16 * 16 * 64 + 4)
    L = pkg.lib()
    args = lambda xo, uo, oo: (x.data_ptr() + xo, U.data_ptr() + uo, b.data_ptr(), s.data_ptr(), out.data_ptr() + oo, 1, 64, 64, 1, None)
    assert L.wino_conv3x3_bn_relu(*args(0, 0, 0)) == 0
    for off in ((4, 0, 0), (0, 8, 0), (0, 0, 12)):
        assert L.wino_conv3x3_bn_relu(*args(*off)) == -3, off
        assert b"16-byte aligned" in L.wino_last_error_string()
    A, Bm, C = z(8 * 64 + 4), z(64 * 64 + 4), z(8 * 64 + 4)
    one = lambda ao, bo, co: L.wino_conv1x1_bn(A.data_ptr() + ao, Bm.data_ptr() + bo, b.data_ptr(), s.data_ptr(), C.data_ptr() + co, 8, 64, 64, 1, None)
    assert one(0, 0, 0) == 0
    for off in ((4, 0, 0), (0, 4, 0), (0, 0, 4)):
        assert one(*off) == -3, off
    torch.cuda.synchronize()


# ------------------------------------------------------------------ the reference entry points
LAYERS = ["kernel_128", "kernel_256", "kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in", "kernel_256_1_out"]


@pytest.mark.parametrize("mode", range(6))
def test_test_binary_modes(mode, data_dir):
    """`./Test <mode>` on the reference data set: reference stdout protocol, small errors."""
    exe = os.path.join(ROOT, "Test")
    assert os.path.exists(exe), "run make first"
    r = subprocess.run([exe, str(mode), "1", "1", "4"], cwd=data_dir, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "---- Iter: 0 ----" and lines[1].startswith("TotalTime = ")
    assert any(l.startswith("Average Total Time: [Mine: ") for l in lines)
    import json
    js = json.loads(lines[-1])
    assert js["layer"] == LAYERS[mode] and js["N"] == 1
    assert js["max_rel_err"] < TIGHT, js
    if mode < 2:  # 3x3 outputs are O(1): the reference's absolute checker is meaningful
        assert js["max_abs_err"] < 1e-4


def test_entry_points_batched_via_abi(data_dir, pkg):
    """kernel_256() with N=16 set through the driver configuration calls."""
    L = pkg.lib()
    cwd = os.getcwd()
    os.chdir(data_dir)
    try:
        L.wino_driver_set_quiet(1)
        L.wino_driver_set_batch(16)
        packed = L.kernel_256()
        res = pkg.DriverResult()
        assert L.wino_driver_last_result(ctypes.byref(res)) == 0
        assert res.N == 16 and res.gpus == 1
        assert res.max_rel_err < TIGHT and res.max_abs_err < 1e-4
        assert (packed >> 16) == min(int(res.mine_us), 0x7FFF)
    finally:
        L.wino_driver_set_batch(1)
        L.wino_driver_set_quiet(0)
        os.chdir(cwd)
