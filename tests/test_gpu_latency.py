"""GPU parity tests of the latency forms -- the reference's own operating point is ONE image
(`./Test 0..5`: Kernel128_winograd.cu:263-265, Kernel128_one.cu:98,316, Kernel256_one.cu:100,318) -- and of
the recovery contract for the library's ticket counters.  Run with `-m gpu`; everything through the C-ABI,
the CPU oracle is only the checker."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 2e-5


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch, torch.device("cuda:0")


def _ring():
    r = np.ones((16, 16), bool)
    r[1:15, 1:15] = False
    return r


def _layer(torch_dev, seed, N, C, K):
    torch, dev = torch_dev
    rng = np.random.RandomState(seed)
    x = (rng.rand(N, 16, 16, C) - 0.5).astype(np.float32)
    w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    s = (rng.rand(K) - 0.5).astype(np.float32)
    b = (rng.rand(K) - 0.5).astype(np.float32)
    return (x, w, s, b), tuple(torch.from_numpy(a).to(dev) for a in (x, w, s, b))


# ------------------------------------------------------------------ 3x3 latency kernel: every (block width, split) form
@pytest.mark.parametrize("N,C,K", [(1, 256, 256), (1, 128, 128), (2, 256, 256), (3, 128, 128), (1, 64, 192), (5, 16, 64),
                                   (2, 48, 128), (4, 512, 64), (4, 256, 256), (7, 128, 128), (3, 64, 192), (5, 32, 64),
                                   (9, 512, 64), (2, 16, 64), (3, 80, 128)])
def test_conv3x3_latency_forms_agree(N, C, K, pkg, O, torch_dev, knobs):
    """A block (16 tiles x 16 CT out-channels, CT MFMA tiles per wave) has its contraction dealt over 4 S waves as
    (16-channel super-chunk, row group) tasks -- the pixels staged through LDS by the whole workgroup, the filter
    points prefetched progressively -- and reduced in two levels (LDS inside a workgroup; write-through slabs + one
    ticket per workgroup across the S workgroups).  Every legal (CT, S) must give the fp64 oracle's values on
    NaN-filled outputs (ragged last tile block, odd super-chunk counts and ring included), be bitwise reproducible,
    leave the counters at zero, and differ from the automatic form only by fp32 summation order."""
    torch, dev = torch_dev
    (x, w, s, b), (xt, wt, st, bt) = _layer(torch_dev, 100 + N + C, N, C, K)
    U = pkg.filter_transform_f2(wt)
    want = O.conv3x3_bn_relu_direct(x, w, s, b)
    scale = float(np.abs(want).max())
    knobs.set("WINO_3X3_ALGO", "small")
    nsuper = C // 16
    ref = None
    for ct in (1, 2, 4):
        knobs.set("WINO_SMALL_CT", ct)
        for sp in (1, 2, 3, 4, 5, 8):
            if sp > 1 and 4 * sp > nsuper * 2:
                continue
            knobs.set("WINO_SMALL_SPLIT", sp)
            use, gpr, gsp, gct, wgs = pkg.small_plan_3x3_full(N, C, K)
            assert (use, gpr, gsp, gct) == (1, 2, sp, ct), (ct, sp)
            assert wgs == -(-N * 49 // 16) * (K // (16 * ct)) * sp
            out = torch.full((N, 16, 16, K), float("nan"), device=dev)
            pkg.conv3x3_bn_relu(xt, U, bt, st, out=out)
            assert pkg.tickets_in_use() == 0, (ct, sp)
            got = out.cpu().numpy()
            assert not np.isnan(got).any(), (ct, sp)
            assert O.rel_error(got, want) < TIGHT, (ct, sp, O.rel_error(got, want))
            assert (got[:, _ring(), :] == 0).all(), (ct, sp)
            for _ in range(3):
                assert torch.equal(pkg.conv3x3_bn_relu(xt, U, bt, st), out), (ct, sp)
            if ref is None:
                ref = out
            assert float((out - ref).abs().max()) < 4e-6 * scale, (ct, sp)
    for k in ("WINO_SMALL_CT", "WINO_SMALL_SPLIT", "WINO_3X3_ALGO"):
        knobs.unset(k)
    auto = pkg.conv3x3_bn_relu(xt, U, bt, st)
    assert float((auto - ref).abs().max()) < 4e-6 * scale
    assert pkg.tickets_in_use() == 0


@pytest.mark.parametrize("N,H,W,C,K", [(1, 7, 7, 512, 512), (1, 28, 28, 128, 128), (1, 56, 56, 64, 64), (2, 5, 9, 32, 64),
                                       (3, 1, 1, 16, 64), (1, 13, 15, 48, 128), (2, 8, 12, 16, 64), (1, 2, 2, 16, 192)])
def test_conv3x3_latency_other_feature_maps(N, H, W, C, K, pkg, O, torch_dev, knobs):
    """the latency kernel with the geometry in its arguments (ResNet's 56x56, 28x28 and 7x7
    stages at one image, odd sizes whose last tile row / column is clipped, a single tile or pixel): every block width
    and split on NaN-filled [N][H+2][W+2][K] outputs -- interior against the fp64 oracle, ring exact zeros, bitwise
    repeatable, counters at zero -- and the automatic choice."""
    torch, dev = torch_dev
    rng = np.random.RandomState(H * 131 + W + C)
    x = (rng.rand(N, H + 2, W + 2, C) - 0.5).astype(np.float32)
    w = (rng.rand(K, C, 3, 3) - 0.5).astype(np.float32)
    s = (rng.rand(K) - 0.5).astype(np.float32)
    b = (rng.rand(K) - 0.5).astype(np.float32)
    xt, wt, st, bt = (torch.from_numpy(a).to(dev) for a in (x, w, s, b))
    U = pkg.filter_transform_f2(wt)
    want = O.conv3x3_bn_relu_direct(x, w, s, b)
    scale = float(np.abs(want).max())
    ring = np.ones((H + 2, W + 2), bool)
    ring[1:H + 1, 1:W + 1] = False
    tiles = ((H + 1) // 2) * ((W + 1) // 2)
    knobs.set("WINO_3X3_ALGO", "small")
    ref = None
    for ct in (1, 2, 4):
        if K % (16 * ct):
            continue
        knobs.set("WINO_SMALL_CT", ct)
        for sp in (1, 2, 4, 8):
            if sp > 1 and 4 * sp > (C // 16) * 2:
                continue
            knobs.set("WINO_SMALL_SPLIT", sp)
            use, gpr, gsp, gct, wgs = pkg.small_plan_3x3_full(N, C, K, H=H, W=W)
            assert (use, gsp, gct) == (1, sp, ct) and wgs == -(-N * tiles // 16) * (K // (16 * ct)) * sp, (ct, sp)
            out = torch.full((N, H + 2, W + 2, K), float("nan"), device=dev)
            pkg.conv3x3_bn_relu(xt, U, bt, st, out=out)
            assert pkg.tickets_in_use() == 0, (ct, sp)
            got = out.cpu().numpy()
            assert not np.isnan(got).any(), (ct, sp)
            assert O.rel_error(got, want) < TIGHT, (ct, sp, O.rel_error(got, want))
            assert (got[:, ring, :] == 0).all(), (ct, sp)
            assert torch.equal(pkg.conv3x3_bn_relu(xt, U, bt, st), out), (ct, sp)
            if ref is None:
                ref = out
            assert float((out - ref).abs().max()) < 4e-6 * scale, (ct, sp)
    for k in ("WINO_SMALL_CT", "WINO_SMALL_SPLIT", "WINO_3X3_ALGO"):
        knobs.unset(k)
    assert ref is not None
    auto = pkg.conv3x3_bn_relu(xt, U, bt, st)
    assert float((auto - ref).abs().max()) < 4e-6 * scale
    assert pkg.tickets_in_use() == 0


def test_conv3x3_latency_with_a_competing_stream(pkg, torch_dev, knobs):
    """Split blocks while a second stream's launches hold CUs: the S workgroups of a block then start at
    different times and any of them may be the finisher.  Bitwise equal results, counters at zero."""
    torch, dev = torch_dev
    (_, _, _, _), (xt, wt, st, bt) = _layer(torch_dev, 7, 1, 256, 256)
    (_, _, _, _), (xs, ws, ss, bs) = _layer(torch_dev, 8, 64, 128, 128)
    U, Us = pkg.filter_transform_f2(wt), pkg.filter_transform_f2(ws)
    assert pkg.small_plan_3x3(1, 256, 256)[2] > 1
    ref = pkg.conv3x3_bn_relu(xt, U, bt, st).clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(30):
        with torch.cuda.stream(side):
            for _ in range(2):
                pkg.conv3x3_bn_relu(xs, Us, bs, ss)
        for o in [pkg.conv3x3_bn_relu(xt, U, bt, st) for _ in range(8)]:
            assert torch.equal(o, ref), rep
    torch.cuda.synchronize()
    assert pkg.tickets_in_use() == 0


def test_conv3x3_latency_in_a_graph(pkg, torch_dev):
    """conv3x3_prepare allocates the split form's slabs and tickets ahead of a capture; the replayed graph gives
    the eager result."""
    torch, dev = torch_dev
    (_, _, _, _), (xt, wt, st, bt) = _layer(torch_dev, 9, 1, 128, 128)
    U = pkg.filter_transform_f2(wt)
    ref = pkg.conv3x3_bn_relu(xt, U, bt, st).clone()
    stream = torch.cuda.Stream()
    out = torch.zeros_like(ref)
    with torch.cuda.stream(stream):
        pkg.conv3x3_prepare(1, 128, 128)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            pkg.conv3x3_bn_relu(xt, U, bt, st, out=out)
        for _ in range(3):
            out.fill_(float("nan"))
            graph.replay()
            stream.synchronize()
            assert torch.equal(out, ref)
        assert pkg.tickets_in_use() == 0


# ------------------------------------------------------------------ 1x1 latency form
@pytest.mark.parametrize("M,Cin,Kout,relu", [(196, 1024, 256, True), (196, 512, 128, True), (196, 128, 512, False),
                                            (196, 256, 1024, False), (1, 32, 64, True), (17, 96, 64, False),
                                            (392, 1024, 256, True), (50, 160, 192, True), (200, 2048, 64, True)])
def test_one_by_one_latency_forms_agree(M, Cin, Kout, relu, pkg, torch_dev, knobs):
    """Blocks of (16 RT) x (16 CT) outputs per wave, the K loop split over KS waves of a workgroup: every legal
    (RT, CT, KS) form, the forced tiled kernel and the automatic choice must all match an fp64 GEMM, the latency
    forms bit for bit from launch to launch and on NaN-filled outputs (ragged last row block, ragged K groups)."""
    torch, dev = torch_dev
    g = torch.Generator(device="cpu").manual_seed(M + Cin)
    mk = lambda *s: ((torch.rand(*s, generator=g) - 0.5) * 4).to(dev)
    A, Bm, b, s = mk(M, Cin), mk(Cin, Kout), mk(Kout), mk(Kout)
    want = (A.double() @ Bm.double()) * s.double() + b.double()
    if relu:
        want = torch.relu(want)
    scale = float(want.abs().max())

    def check(tag):
        out = torch.full((M, Kout), float("nan"), device=dev)
        pkg.conv1x1_bn(A, Bm, b, s, relu, out=out)
        assert not bool(torch.isnan(out).any()), tag
        assert float((out.double() - want).abs().max()) < TIGHT * scale, tag
        for _ in range(2):
            assert torch.equal(pkg.conv1x1_bn(A, Bm, b, s, relu), out), tag
        return out

    knobs.set("WINO_1X1_ALGO", "big")
    big = check("big")
    knobs.set("WINO_1X1_ALGO", "small")
    forms = 0
    for rt in (1, 2):
        for ct in (1, 2, 4):   # 4: strided column tiles, 16-byte filter loads
            for ks in (1, 2, 4):
                if Cin % (16 * ks) or Kout % ((4 // ks) * ct * 16) or (Cin // ks < 64 and ks > 1):
                    continue
                knobs.set("WINO_1X1_SMALL_KS", ks)
                knobs.set("WINO_1X1_SMALL_RT", rt)
                knobs.set("WINO_1X1_SMALL_CT", ct)
                assert pkg.small_plan_1x1_full(M, Cin, Kout)[:4] == (1, ks, rt, ct)
                out = check((rt, ct, ks))
                assert float((out - big).abs().max()) < 4e-6 * scale, (rt, ct, ks)
                forms += 1
    assert forms >= 2
    for k in ("WINO_1X1_ALGO", "WINO_1X1_SMALL_KS", "WINO_1X1_SMALL_RT", "WINO_1X1_SMALL_CT"):
        knobs.unset(k)
    check("auto")
    assert pkg.tickets_in_use() == 0


@pytest.mark.parametrize("N,H,W,Cin,Kout", [(1, 14, 14, 1024, 256), (2, 14, 14, 256, 1024), (3, 7, 7, 512, 128), (2, 5, 9, 96, 192),
                                            (8, 14, 14, 256, 1024)])
def test_one_by_one_latency_chained_forms(N, H, W, Cin, Kout, pkg, torch_dev, knobs):
    """The bottleneck block's chaining layouts on the latency form: output into the interior of a
    NaN-filled padded tensor with the ring written as exact zeros, input from a padded tensor whose ring holds garbage,
    residual added before the ReLU -- every block shape against an fp64 GEMM and against the tiled kernel."""
    torch, dev = torch_dev
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + Cin)
    mk = lambda *s: ((torch.rand(*s, generator=g) - 0.5) * 2).to(dev)
    A, Bm, b, s, R = mk(N, H, W, Cin), mk(Cin, Kout), mk(Kout), mk(Kout), mk(N * H * W, Kout)
    Ap = (mk(N, H + 2, W + 2, Cin) * 100).contiguous()
    Ap[:, 1:-1, 1:-1, :] = A
    lin = (A.reshape(-1, Cin).double() @ Bm.double()) * s.double() + b.double()
    want, want_res = torch.relu(lin), torch.relu(lin + R.double())
    scale = float(want_res.abs().max())
    ring = torch.ones(H + 2, W + 2, dtype=torch.bool, device=dev)
    ring[1:-1, 1:-1] = False

    def run(tag):
        out = torch.full((N, H + 2, W + 2, Kout), float("nan"), device=dev)
        pkg.conv1x1_bn_ex(A, Bm, b, s, pkg.RELU | pkg.C_PADDED, out=out, hw=(H, W))
        assert bool((out[:, ring, :] == 0).all()), tag
        assert float((out[:, 1:-1, 1:-1, :].reshape(-1, Kout).double() - want).abs().max()) < TIGHT * scale, tag
        got2 = pkg.conv1x1_bn_ex(Ap, Bm, b, s, pkg.RELU | pkg.A_PADDED, hw=(H, W))
        assert float((got2.reshape(-1, Kout).double() - want).abs().max()) < TIGHT * scale, tag
        got3 = pkg.conv1x1_bn_ex(Ap, Bm, b, s, pkg.RELU | pkg.A_PADDED | pkg.ADD_RESIDUAL, residual=R, hw=(H, W))
        assert float((got3.reshape(-1, Kout).double() - want_res).abs().max()) < TIGHT * scale, tag
        return got3

    knobs.set("WINO_1X1_ALGO", "big")
    big = run("big")
    knobs.set("WINO_1X1_ALGO", "small")
    forms = 0
    for rt, ct, ks in ((1, 1, 4), (1, 1, 1), (2, 2, 2), (1, 4, 4), (2, 4, 1), (2, 1, 2)):
        if Cin % (16 * ks) or Kout % ((4 // ks) * ct * 16) or (Cin // ks < 64 and ks > 1):
            continue
        knobs.set("WINO_1X1_SMALL_KS", ks)
        knobs.set("WINO_1X1_SMALL_RT", rt)
        knobs.set("WINO_1X1_SMALL_CT", ct)
        got = run((rt, ct, ks))
        assert float((got - big).abs().max()) < 4e-6 * scale, (rt, ct, ks)
        forms += 1
    assert forms >= 1
    for k in ("WINO_1X1_ALGO", "WINO_1X1_SMALL_KS", "WINO_1X1_SMALL_RT", "WINO_1X1_SMALL_CT"):
        knobs.unset(k)
    run("auto")


def test_latency_kernels_random_shapes(pkg, torch_dev, knobs):
    """Seeded sweep of the two latency kernels with their forms FORCED (the automatic choice would send most of these
    shapes to the throughput kernels): random legal shapes -- 3x3: any N, H, W, C % 16, K % 64 (wider blocks where K
    allows), a random split; 1x1: any M, Cin % 32, Kout % 64, random (K-split, row tiles, column tiles), random
    chaining flags -- against the direct GPU comparators (no Winograd, no MFMA) on NaN-filled outputs."""
    torch, dev = torch_dev
    rng = np.random.RandomState(4242)
    mk = lambda *s: torch.from_numpy(((rng.rand(*s) - 0.5) * 2).astype(np.float32)).to(dev)
    knobs.set("WINO_3X3_ALGO", "small")
    for i in range(24):
        N = int(rng.randint(1, 5)); H = int(rng.randint(1, 19)); W = int(rng.randint(1, 19))
        C = 16 * int(rng.randint(1, 13)); K = 64 * int(rng.randint(1, 4))
        ct = int(rng.choice([1, 2, 4]))
        smax = max(1, min(8, (C // 16) // 2))
        sp = int(rng.randint(1, smax + 1))
        knobs.set("WINO_SMALL_CT", ct)
        knobs.set("WINO_SMALL_SPLIT", sp)
        x, w, s, b = mk(N, H + 2, W + 2, C), mk(K, C, 3, 3), mk(K), mk(K)
        U = pkg.filter_transform_f2(w)
        assert pkg.small_plan_3x3_full(N, C, K, H=H, W=W)[:4] == (1, 2, sp, ct), (N, H, W, C, K, ct, sp)
        out = torch.full((N, H + 2, W + 2, K), float("nan"), device=dev)
        pkg.conv3x3_bn_relu(x, U, b, s, relu=bool(i & 1), out=out)
        want = pkg.conv3x3_direct(x, w, b, s, relu=bool(i & 1))
        assert not bool(torch.isnan(out).any()), (N, H, W, C, K, ct, sp)
        inner = (slice(None), slice(1, H + 1), slice(1, W + 1), slice(None))
        assert float((out[inner] - want[inner]).abs().max()) < TIGHT * float(want[inner].abs().max() + 1e-6), (N, H, W, C, K, ct, sp)
        ring = torch.ones(H + 2, W + 2, dtype=torch.bool, device=dev)
        ring[1:-1, 1:-1] = False
        assert bool((out[:, ring, :] == 0).all()), (N, H, W, C, K, ct, sp)
        assert pkg.tickets_in_use() == 0
    for k in ("WINO_3X3_ALGO", "WINO_SMALL_CT", "WINO_SMALL_SPLIT"):
        knobs.unset(k)
    knobs.set("WINO_1X1_ALGO", "small")
    done = 0
    for i in range(60):
        H = int(rng.randint(1, 15)); W = int(rng.randint(1, 15)); N = int(rng.randint(1, 4))
        M = N * H * W
        Cin = 32 * int(rng.randint(1, 17)); Kout = 64 * int(rng.randint(1, 9))
        ks, rt, ct = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2])), int(rng.choice([1, 2, 4]))
        if Cin % (16 * ks) or Kout % ((4 // ks) * ct * 16) or (Cin // ks < 64 and ks > 1):
            continue
        knobs.set("WINO_1X1_SMALL_KS", ks)
        knobs.set("WINO_1X1_SMALL_RT", rt)
        knobs.set("WINO_1X1_SMALL_CT", ct)
        flags = int(rng.choice([0, pkg.RELU])) | int(rng.choice([0, pkg.A_PADDED])) | int(rng.choice([0, pkg.C_PADDED])) | int(rng.choice([0, pkg.ADD_RESIDUAL]))
        A, Bm, b, s, R = mk(N, H, W, Cin), mk(Cin, Kout), mk(Kout), mk(Kout), mk(M, Kout)
        Ap = mk(N, H + 2, W + 2, Cin) * 50
        Ap[:, 1:-1, 1:-1, :] = A
        lin = pkg.conv1x1_direct(A.reshape(M, Cin), Bm, b, s, False)
        if flags & pkg.ADD_RESIDUAL:
            lin = lin + R
        want = torch.relu(lin) if flags & pkg.RELU else lin
        shape = (N, H + 2, W + 2, Kout) if flags & pkg.C_PADDED else (M, Kout)
        out = torch.full(shape, float("nan"), device=dev)
        pkg.conv1x1_bn_ex(Ap if flags & pkg.A_PADDED else A, Bm, b, s, flags, residual=R if flags & pkg.ADD_RESIDUAL else None,
                          out=out, hw=(H, W))
        tag = (M, Cin, Kout, ks, rt, ct, flags)
        if flags & pkg.C_PADDED:
            ring = torch.ones(H + 2, W + 2, dtype=torch.bool, device=dev)
            ring[1:-1, 1:-1] = False
            assert bool((out[:, ring, :] == 0).all()), tag
            got = out[:, 1:-1, 1:-1, :].reshape(M, Kout)
        else:
            got = out
        assert not bool(torch.isnan(got).any()), tag
        assert float((got - want).abs().max()) < TIGHT * float(want.abs().max() + 1e-6), tag
        done += 1
    assert done >= 20
    for k in ("WINO_1X1_ALGO", "WINO_1X1_SMALL_KS", "WINO_1X1_SMALL_RT", "WINO_1X1_SMALL_CT"):
        knobs.unset(k)


# ------------------------------------------------------------------ recovery after an aborted launch
@pytest.mark.parametrize("kind", ["3x3 throughput", "3x3 latency", "3x3 latency wide", "1x1"])
def test_a_dirty_ticket_counter_is_reported_and_reset_recovers(kind, pkg, torch_dev, knobs):
    """The reference holds no state between calls and exits on the first CUDA error (Kernel128_winograd.cu:16-22,
    236-256).  The one piece of state this library keeps is the ticket counters of a stream's scratch, zero
    between launches.  A launch that died mid-way would leave some non-zero; wino_debug_poison_ticket fakes
    that.  Contract: the kernel that meets a counter that cannot have been zero at launch says so, from then
    on every launch on the stream fails with WINO_E_STATE (never computes with counters it cannot trust),
    wino_stream_reset_scratch() recovers, and results are then bitwise those of the clean run."""
    torch, dev = torch_dev
    g = torch.Generator(device="cpu").manual_seed(31)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        if kind == "1x1":
            knobs.set("WINO_1X1_ALGO", "big")
            A, Bm, b, s = mk(2 * 196, 1024), mk(1024, 256), mk(256), mk(256)
            run = lambda: pkg.conv1x1_bn(A, Bm, b, s, True)
            n_tickets = 8
        else:
            N = {"3x3 throughput": 40, "3x3 latency": 1, "3x3 latency wide": 4}[kind]
            if kind == "3x3 throughput":
                knobs.set("WINO_3X3_ALGO", "big")
                knobs.set("WINO_SK_GRID", "256")
            x, w, s, b = mk(N, 16, 16, 256), mk(256, 256, 3, 3), mk(256), mk(256)
            U = pkg.filter_transform_f2(w)
            run = lambda: pkg.conv3x3_bn_relu(x, U, b, s)
            if kind == "3x3 latency wide":
                assert pkg.small_plan_3x3_full(N, 256, 256)[1:] == (2, 2, 2, 208)
            n_tickets = {"3x3 latency": 64, "3x3 latency wide": 104}.get(kind, 8 * 4 * ((N * 49 + 63) // 64))
        ref = run().clone()
        assert pkg.tickets_in_use() == 0
        pkg.stream_check()
        # every counter the launch draws on is left one too high, as by a launch that never finished
        for i in range(n_tickets):
            pkg.poison_ticket(i, 1)
        run()   # computes with dirty counters: its results are not to be trusted, and it must say so
        with pytest.raises(pkg.WinoError, match="rc=-4"):
            pkg.stream_check()
        with pytest.raises(pkg.WinoError, match="rc=-4"):
            run()
        pkg.stream_reset_scratch()
        pkg.stream_check()
        assert pkg.tickets_in_use() == 0
        for _ in range(3):
            assert torch.equal(run(), ref)
        assert pkg.tickets_in_use() == 0
    torch.cuda.synchronize()


def test_in_kernel_clock_of_the_last_launch(pkg, torch_dev, knobs):
    """wino_diag_last_clock: workgroup 0 of every product launch stamps {cycles, 100 MHz ticks} at entry and
    exit.  The clock must be a plausible shader clock and the stamped span must fit inside the launch."""
    torch, dev = torch_dev
    g = torch.Generator(device="cpu").manual_seed(5)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    x, w, s, b = mk(128, 16, 16, 256), mk(256, 256, 3, 3), mk(256), mk(256)
    U = pkg.filter_transform_f2(w)
    for _ in range(50):
        pkg.conv3x3_bn_relu(x, U, b, s)
    ghz, cycles, us = pkg.last_clock_ghz(0)
    assert 1.0 < ghz < 2.6 and 50 < us < 400, (ghz, cycles, us)
    A, Bm = mk(128 * 196, 256), mk(256, 1024)
    b2, s2 = mk(1024), mk(1024)
    for _ in range(20):
        pkg.conv1x1_bn(A, Bm, b2, s2, False)
    ghz, cycles, us = pkg.last_clock_ghz(1)
    assert 1.0 < ghz < 2.6 and 1 < us < 400, (ghz, cycles, us)
