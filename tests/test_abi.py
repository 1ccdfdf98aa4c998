"""The C-ABI library loads, exports every symbol include/*.h declares, and its host-side
logic (util.c helpers, driver configuration, argument checking) behaves like the
reference's.  No GPU compute is attempted here."""
import ctypes
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, ptr

INCLUDE = os.path.join(ROOT, "include")


def _declared_functions():
    names = set()
    for h in sorted(os.listdir(INCLUDE)):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(INCLUDE, h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    declared = _declared_functions()
    assert {"wino_conv3x3_bn_relu", "wino_conv1x1_bn", "kernel_128", "kernel_256_1_out",
            "get_parameter", "output_checker"} <= declared
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(pkg.ABI_SYMBOLS), declared ^ set(pkg.ABI_SYMBOLS)
    assert L.wino_abi_version() == 1


def test_exported_table_has_no_torch_types():
    out = subprocess.check_output(["nm", "-D", "--defined-only",
                                   os.path.join(ROOT, "cuda-winograd_amd", "libwinograd_mi355x.so")]).decode()
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert "wino_conv3x3_bn_relu" in syms and "kernel_128" in syms
    assert not [s for s in syms if "torch" in s.lower() or "at::" in s]


def test_get_parameter_and_transpose(pkg, tmp_path):
    L = pkg.lib()
    L.get_parameter.restype = ctypes.POINTER(ctypes.c_float)
    L.get_parameter.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.transpose.restype = ctypes.POINTER(ctypes.c_float)
    L.transpose.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int]
    a = np.arange(12, dtype="<f4")
    p = tmp_path / "x.bin"
    a.tofile(p)
    buf = L.get_parameter(str(p).encode(), 12)
    got = np.ctypeslib.as_array(buf, shape=(12,)).copy()
    np.testing.assert_array_equal(got, a)
    # transpose(weight, h, w): input [w][h] -> output [h][w] (util.c:15-26), frees its input
    t = L.transpose(buf, 3, 4)
    got_t = np.ctypeslib.as_array(t, shape=(3, 4)).copy()
    np.testing.assert_array_equal(got_t, a.reshape(4, 3).T)


def test_get_parameter_missing_file_exits_zero(tmp_path):
    """Reference behaviour: message + exit(0) (util.c:30-39)."""
    code = ("import ctypes,sys;L=ctypes.CDLL(sys.argv[1]);"
            "L.get_parameter.argtypes=[ctypes.c_char_p,ctypes.c_int];L.get_parameter(b'/nonexistent/x.bin',4);"
            "print('survived')")
    r = subprocess.run(["python", "-c", code, os.path.join(ROOT, "cuda-winograd_amd", "libwinograd_mi355x.so")],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "Bad file path" in r.stdout and "survived" not in r.stdout


def test_output_checker_matches_oracle(pkg, O, capfd):
    L = pkg.lib()
    L.output_checker.restype = ctypes.c_float
    L.output_checker.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rng = np.random.RandomState(2)
    B = rng.rand(14, 14, 16).astype(np.float32)
    A = np.zeros((16, 16, 16), np.float32)
    A[1:15, 1:15] = B
    A[2, 3, 4] += 0.25
    A[9, 9, 0] -= 2e-5
    m = L.output_checker(ptr(A), ptr(B), 14, 16, 1)
    want_m, want_c = O.output_checker(A, B, 14, 16, 1)
    assert abs(m - want_m) < 1e-7
    ctypes.CDLL(None).fflush(None)  # C stdio is block-buffered when captured
    out = capfd.readouterr().out
    assert out.strip() == "[max_error: %f][error_cnt: %d]" % (want_m, want_c)


def test_driver_configuration(pkg):
    L = pkg.lib()
    assert L.wino_driver_set_batch(0) != 0 and L.wino_driver_set_gpus(0) != 0
    assert L.wino_driver_set_batch(128) == 0 and L.wino_driver_get_batch() == 128
    assert L.wino_driver_set_gpus(8) == 0 and L.wino_driver_get_gpus() == 8
    L.wino_driver_set_batch(1); L.wino_driver_set_gpus(1)


def test_argument_checks_fail_loudly(pkg):
    L = pkg.lib()
    assert L.wino_filter_f2_elems(256, 256) == 16 * 256 * 256
    one = ctypes.c_void_p(16)  # never dereferenced: shape checks come first
    assert L.wino_conv3x3_bn_relu(None, None, None, None, None, 1, 128, 128, 1, None) == -3
    assert L.wino_conv3x3_bn_relu(one, one, one, one, one, 1, 100, 128, 1, None) == -2
    assert L.wino_conv3x3_bn_relu(one, one, one, one, one, 0, 128, 128, 1, None) == -2
    assert b"need C" in L.wino_last_error_string() or b"batch" in L.wino_last_error_string()
    assert L.wino_conv1x1_bn(one, one, one, one, one, 196, 100, 128, 1, None) == -2
    assert L.wino_conv1x1_bn(one, one, one, one, one, 196, 128, 100, 1, None) == -2


def test_python_ops_reject_cpu_tensors(pkg):
    """No CPU fallback: the operator wrappers refuse host tensors."""
    import torch
    x = torch.zeros(1, 16, 16, 128)
    with pytest.raises(pkg.WinoError):
        pkg.conv3x3_bn_relu(x, torch.zeros(16 * 128 * 128), torch.zeros(128), torch.zeros(128))
    with pytest.raises(pkg.WinoError):
        pkg.conv1x1_bn(torch.zeros(4, 128), torch.zeros(128, 128), torch.zeros(128), torch.zeros(128), True)


def test_shard_range_partitions(pkg):
    for N in (1, 7, 128, 1024):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                a, b = pkg.shard_range(N, r, world)
                cover += list(range(a, b))
            assert cover == list(range(N))
    with pytest.raises(ValueError):
        pkg.shard_range(8, 2, 2)


def test_packed_filter_index_is_a_bijection(pkg):
    """Every (point, in-channel, out-channel) owns exactly one slot of the packed F(2x2) buffer,
    and the 8 channels x 64 out-channels x 16 points one workgroup stages per pipeline step are
    contiguous (one 32 KB LDS-DMA slice)."""
    L = pkg.lib()
    C, K = 24, 128
    idx = np.array([[[L.wino_filter_f2_index(C, K, e, c, k) for k in range(K)] for c in range(C)]
                    for e in range(16)])
    assert idx.min() == 0 and idx.max() == 16 * C * K - 1
    assert np.unique(idx).size == idx.size
    for chunk in range(C // 8):
        for kb in range(K // 64):
            sl = idx[:, 8 * chunk:8 * chunk + 8, 64 * kb:64 * kb + 64]
            assert sl.max() - sl.min() == 16 * 8 * 64 - 1 and sl.min() % 8192 == 0
    assert L.wino_filter_f2_index(C, K, 16, 0, 0) == -1 and L.wino_filter_f2_index(C, K, 0, C, 0) == -1


# ------------------------------------------------------------------ launch plan (host logic, no GPU)
def _plan(pkg, N, H, W, C, K, cus):
    L = pkg.lib()
    g, r, it = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    t = ctypes.c_long()
    rc = L.wino_conv3x3_plan(N, H, W, C, K, cus, ctypes.byref(g), ctypes.byref(r), ctypes.byref(t), ctypes.byref(it))
    assert rc == 0, L.wino_last_error_string()
    return g.value, r.value, t.value, it.value


@pytest.mark.parametrize("N,H,W,C,K,cus", [
    (128, 14, 14, 256, 256, 256),   # the headline: 392 items on 256 CUs
    (128, 14, 14, 128, 128, 256),   # 196 items: fewer than CUs
    (1024, 14, 14, 256, 256, 256),  # many rounds
    (32, 14, 14, 256, 256, 256), (1, 14, 14, 8, 64, 256), (128, 56, 56, 64, 64, 256),
    (128, 7, 7, 512, 512, 304), (5, 28, 28, 128, 192, 64), (17, 14, 14, 128, 256, 8), (3, 2, 2, 8, 64, 1),
])
def test_launch_plan_covers_every_iteration_once(N, H, W, C, K, cus, pkg, knobs):
    """The throughput kernel's work decomposition, replayed on the host: `rounds` whole items per
    logical workgroup plus an evenly cut stream-K tail must cover every (item, chunk) iteration
    exactly once, never give one workgroup more than ceil(T/G) + a whole item's slack, and cut an
    item only inside the tail."""
    knobs.unset("WINO_SK_GRID")
    knobs.unset("WINO_SK_MIN_ITERS")
    G, rounds, tail_iters, nch = _plan(pkg, N, H, W, C, K, cus)
    tiles = N * ((H + 1) // 2) * ((W + 1) // 2)
    items = -(-tiles // 64) * (K // 64)
    assert nch == C // 8 and G >= 1 and rounds == items // G
    assert tail_iters == (items % G) * nch
    assert G <= max(cus, 1) or G == items          # at most one workgroup per CU, or one item each
    # the tail is cut per out-channel block when the grid is a multiple of K/64 (the k-blocks of a tile block then
    # walk the same patches in step): `groups` lists of tail_iters / groups iterations over G / groups ranges
    grp, per, inv, cop = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert pkg.lib().wino_conv3x3_plan_groups(N, H, W, C, K, cus, ctypes.byref(grp), ctypes.byref(per), ctypes.byref(inv),
                                              ctypes.byref(cop)) == 0
    kp, P, pinv, copies = grp.value, per.value, inv.value, cop.value
    assert kp in (1, K // 64) and G % kp == 0 and tail_iters % kp == 0
    assert (kp == K // 64) == (G % (K // 64) == 0)
    # position j of a group -> the tail range it runs: a permutation of the group's ranges (the identity for period 1)
    range_of = lambda j: (pinv * (j // copies)) % P + P * (j % copies)
    assert P * copies == G // kp and sorted(range_of(j) for j in range(G // kp)) == list(range(G // kp))
    Tg, Gp = tail_iters // kp, G // kp
    seen = np.zeros((items, nch), np.int32)
    loads = []
    slots = set()
    for l in range(G):
        mine = 0
        j, k = divmod(l, kp)
        lp = range_of(j)
        t0, t1 = lp * Tg // Gp, (lp + 1) * Tg // Gp     # the kernel's sk_start() in the group's tail space
        for t in range(t0, t1):
            item = rounds * G + (t // nch) * kp + k
            assert item % (K // 64) == k or kp == 1      # group k only ever touches k-block k
            seen[item, t % nch] += 1
        mine += t1 - t0
        for r in range(rounds):
            seen[r * G + l, :] += 1
            mine += nch
        loads.append(mine)
    assert (seen == 1).all()
    assert max(loads) - min(loads) <= 1
    # the two headline facts of DESIGN.md section 3.1
    if (N, H, C, K, cus) == (128, 14, 256, 256, 256):
        assert (G, rounds, tail_iters) == (256, 1, 136 * 32) and max(loads) == 49
        # 4 k-groups of 64 ranges of 17 iterations; phases 17 r mod 32 repeat with period 32; an XCD's 8 positions
        # hold 4 consecutive phases, each twice
        assert (kp, P, copies) == (4, 32, 2)
        for x in range(8):
            phases = sorted((17 * range_of(8 * x + i)) % 32 for i in range(8))
            assert phases == sorted(2 * [4 * x, 4 * x + 1, 4 * x + 2, 4 * x + 3])
    if (N, H, C, K, cus) == (128, 14, 128, 128, 256):
        assert (G, rounds, tail_iters) == (196, 1, 0)


def _plan1x1(pkg, M, Cin, Kout, cus):
    v = [ctypes.c_int() for _ in range(5)]
    rc = pkg.lib().wino_conv1x1_plan(M, Cin, Kout, cus, *[ctypes.byref(x) for x in v])
    assert rc == 0, pkg.lib().wino_last_error_string()
    return [x.value for x in v]


@pytest.mark.parametrize("M,Cin,Kout,cus,grid_env", [
    (25088, 1024, 256, 256, None), (25088, 512, 128, 256, None), (25088, 256, 1024, 256, None),
    (25088, 128, 512, 256, None), (196, 1024, 256, 256, "16"), (1000, 512, 128, 256, "104"),
    (588, 256, 1024, 256, "4096"), (2500, 128, 512, 304, "1000"), (113, 64, 64, 8, "8"),
    (40000, 2048, 192, 256, None), (5000, 96, 192, 64, "48"),
    (196, 1024, 256, 256, None), (196, 512, 128, 256, None), (196, 256, 1024, 256, None), (196, 128, 512, 256, None),
    (32 * 196, 1024, 256, 256, None), (64 * 196, 1024, 256, 256, None),
])
def test_one_by_one_plan_covers_every_step_once(M, Cin, Kout, cus, grid_env, pkg, knobs):
    """The 1x1 kernel's stream-K decomposition, replayed on the host the way the kernel walks it:
    every (tile, k-step) is computed exactly once, a range is cut into whole tiles plus at most one
    head and one tail segment, no two partial segments share a slab slot, every cut tile's
    segments come from consecutive ranges (what the finisher's gather assumes), and the ranges are
    balanced to one step.  The two facts DESIGN.md section 3.2 states for the reference layers."""
    knobs.unset("WINO_1X1_SK")
    if grid_env is None:
        knobs.unset("WINO_1X1_SK_GRID")
    else:
        knobs.set("WINO_1X1_SK", "1")
        knobs.set("WINO_1X1_SK_GRID", grid_env)
    G, nMB, nblk, nk, sk = _plan1x1(pkg, M, Cin, Kout, cus)
    assert nMB == -(-M // 112) and nk == Cin // 32 and Kout % nblk == 0 and Kout // nblk in (64, 128)
    tiles = nMB * nblk
    if (M, Cin, Kout, cus, grid_env) == (25088, 1024, 256, 256, None):
        assert (sk, G) == (1, 512)            # 448 tiles on 256 CUs: 28 of 32 k-steps per range
    if (M, Cin, Kout, cus, grid_env) in ((25088, 512, 128, 256, None), (25088, 256, 1024, 256, None), (25088, 128, 512, 256, None)):
        assert sk == 0                        # 14 steps per range: not worth the hand-over; exact rounds
    # small batches (fewer tiles than CUs): split-K when the K loop is long enough to pay for the hand-over
    # (4-step ranges for the 32- and 16-step layers, 2-step ranges for 256->1024's 8, never for 128->512's 4)
    small = {(196, 1024, 256): (1, 32), (196, 512, 128): (1, 16), (196, 256, 1024): (1, 64), (196, 128, 512): (0, None),
             (32 * 196, 1024, 256): (1, 256), (64 * 196, 1024, 256): (0, None)}
    if grid_env is None and cus == 256 and (M, Cin, Kout) in small:
        want_sk, want_G = small[(M, Cin, Kout)]
        assert sk == want_sk and (want_G is None or G == want_G)
    if not sk:
        assert G >= tiles and G % (8 * nblk) == 0
        return
    assert G % 8 == 0 and G % nblk == 0 and 0 < G <= 16384 and G <= tiles * nk
    R, T = G // nblk, nMB * nk
    seen = np.zeros((nMB, nk), np.int32)
    slots, loads, owners = set(), [], {}
    for r in range(R):
        u, uend = T * r // R, T * (r + 1) // R
        assert uend > u                       # no empty range: the finisher counts ranges, not segments
        loads.append(uend - u)
        partial = 0
        while u < uend:
            mb, k0 = divmod(u, nk)
            ln = min(uend - u, nk - k0)
            seen[mb, k0:k0 + ln] += 1
            if not (k0 == 0 and ln == nk):
                partial += 1
                slot = 2 * r + (1 if k0 == 0 else 0)
                assert slot not in slots
                slots.add(slot)
                owners.setdefault(mb, []).append((k0, r))
            u += ln
        assert partial <= 2
    assert (seen == 1).all()
    assert max(loads) - min(loads) <= 1
    for mb, segs in owners.items():
        segs.sort()
        rs = [r for _, r in segs]
        assert rs == list(range(rs[0], rs[0] + len(rs))) and len(rs) >= 2 and segs[0][0] == 0


def test_latency_plans_at_the_reference_point(pkg, knobs):
    """The reference's own protocol is ONE image (`./Test 0..5`).  On 256 CUs the policy must put those layers
    on (nearly) every CU: the 3x3 layers as 64 blocks x 4 C-splits (256 channels) / 32 blocks x 4 (128 channels), the 1x1 layers as 16 x 16 blocks with the K-split that fills the CUs;
    and hand over to the throughput / tiled kernels where the measurements (profiles/r3) say so."""
    for k in ("WINO_3X3_ALGO", "WINO_SMALL_SPLIT", "WINO_SMALL_CT", "WINO_1X1_ALGO", "WINO_1X1_SMALL_KS", "WINO_1X1_SK",
              "WINO_1X1_SK_GRID"):
        knobs.unset(k)
    # (use, point rows per task, C-split, block width / 16, workgroups)
    assert pkg.small_plan_3x3_full(1, 256, 256, cus=256) == (1, 2, 4, 1, 256)
    assert pkg.small_plan_3x3_full(1, 128, 128, cus=256) == (1, 2, 4, 1, 128)     # 4 S <= 2 C / 16: every wave has a task
    # beyond one image: wider blocks as the batch grows
    assert pkg.small_plan_3x3_full(2, 256, 256, cus=256) == (1, 2, 4, 2, 224)
    assert pkg.small_plan_3x3_full(5, 256, 256, cus=256) == (1, 2, 2, 2, 256)
    assert pkg.small_plan_3x3_full(8, 256, 256, cus=256) == (1, 2, 1, 2, 200)
    assert pkg.small_plan_3x3_full(16, 256, 256, cus=256) == (1, 2, 1, 4, 196)
    assert pkg.small_plan_3x3_full(20, 256, 256, cus=256)[:4] == (1, 2, 1, 4)
    assert pkg.small_plan_3x3(21, 256, 256, cus=256)[0] == 0               # no width fits one round: the throughput kernel
    assert pkg.small_plan_3x3_full(10, 128, 128, cus=256)[:4] == (1, 2, 1, 1)
    assert pkg.small_plan_3x3_full(41, 128, 128, cus=256)[:4] == (1, 2, 1, 4) and pkg.small_plan_3x3(42, 128, 128, cus=256)[0] == 0
    assert pkg.small_plan_3x3(10, 384, 384, cus=256)[0] == 0               # a full round of wide blocks at 384 channels loses
    assert pkg.small_plan_3x3(1, 24, 64, cus=256)[0] == 0                  # C % 16: throughput kernel only
    assert pkg.small_plan_3x3_full(1, 128, 128, cus=256, H=28, W=28)[:4] == (1, 2, 2, 1)   # other feature maps: 196 tiles = 4 images' worth
    assert pkg.small_plan_3x3_full(1, 512, 512, cus=256, H=7, W=7)[:4] == (1, 2, 8, 1)                 # one block of 16 tiles x 32 k-blocks x 8 splits
    # every wave of the S workgroups gets a task: 4 S <= (C / 16) * (4 / PR)
    for C in (16, 32, 48, 64, 96, 128, 192, 256, 384, 512):
        for N in (1, 2, 3, 5):
            use, pr, sp, ct, wgs = pkg.small_plan_3x3_full(N, C, 64, cus=256)
            assert ct in (1, 2, 4) and pr == 2
            if use and sp > 1:
                assert 4 * sp <= (C // 16) * 2 and 1 <= sp <= 8, (C, N, sp)
                assert wgs <= 256
    # (use, K-split, row tiles, column tiles, workgroups) at M = 196
    want = {(1024, 256): (1, 4, 1, 1, 208), (512, 128): (1, 4, 1, 1, 104), (128, 512): (1, 2, 1, 1, 208),
            (256, 1024): (1, 4, 2, 2, 224)}
    for (cin, kout), plan in want.items():
        assert pkg.small_plan_1x1_full(196, cin, kout, cus=256) == plan, (cin, kout)
    # where the tiled kernel takes over (images): measured crossovers, profiles/r3/latency_explore_1x1_forms.json
    for (cin, kout), (last_small, first_big) in {(1024, 256): (16, 24), (512, 128): (24, 48), (128, 512): (8, 12),
                                                 (256, 1024): (8, 12)}.items():
        assert pkg.small_plan_1x1(last_small * 196, cin, kout, cus=256)[0] == 1, (cin, kout, last_small)
        assert pkg.small_plan_1x1(first_big * 196, cin, kout, cus=256)[0] == 0, (cin, kout, first_big)
    # from a few images on a wave holds 2 x 2, then 2 x 4 MFMA tiles (16-byte filter loads on strided column tiles)
    assert pkg.small_plan_1x1_full(4 * 196, 1024, 256, cus=256)[2:4] == (2, 2)
    assert pkg.small_plan_1x1_full(8 * 196, 1024, 256, cus=256)[2:4] == (2, 4)
    # every plan is a legal launch: the K-split divides Cin into whole 16-channel super-chunks, the workgroup's
    # column span divides Kout
    for cin, kout in ((32, 64), (96, 64), (160, 192), (2048, 64), (64, 448), (1024, 256)):
        for M in (1, 17, 196, 1000, 5000):
            use, ks, rt, ct, wgs = pkg.small_plan_1x1_full(M, cin, kout, cus=256)
            assert cin % (16 * ks) == 0 and kout % ((4 // ks) * ct * 16) == 0 and rt in (1, 2) and ct in (1, 2, 4), (cin, kout, M)
    assert pkg.small_plan_1x1(128 * 196, 1024, 256, cus=256)[0] == 0
    # a developer forcing a form of the tiled kernel gets the tiled kernel
    knobs.set("WINO_1X1_SK", "1")
    assert pkg.small_plan_1x1(196, 1024, 256, cus=256)[0] == 0
    knobs.unset("WINO_1X1_SK")
    knobs.set("WINO_1X1_ALGO", "big")
    assert pkg.small_plan_1x1(196, 1024, 256, cus=256)[0] == 0
    knobs.set("WINO_1X1_ALGO", "small")
    knobs.set("WINO_1X1_SMALL_KS", "2")
    assert pkg.small_plan_1x1(50 * 196, 1024, 256, cus=256)[:2] == (1, 2)


def test_one_by_one_plan_rejects_bad_shapes(pkg):
    v = [ctypes.c_int() for _ in range(5)]
    a = [ctypes.byref(x) for x in v]
    L = pkg.lib()
    assert L.wino_conv1x1_plan(100, 48, 128, 256, *a) != 0       # Cin % 32
    assert L.wino_conv1x1_plan(100, 64, 96, 256, *a) != 0        # Kout % 64
    assert L.wino_conv1x1_plan(0, 64, 128, 256, *a) != 0         # no rows
    assert L.wino_conv1x1_plan(100, 64, 128, 0, *a) != 0         # no compute units


def test_launch_plan_rejects_bad_shapes(pkg):
    L = pkg.lib()
    g, r, it = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    t = ctypes.c_long()
    args = (ctypes.byref(g), ctypes.byref(r), ctypes.byref(t), ctypes.byref(it))
    assert L.wino_conv3x3_plan(1, 14, 14, 12, 64, 256, *args) != 0      # C % 8
    assert L.wino_conv3x3_plan(1, 14, 14, 16, 32, 256, *args) != 0      # K % 64
    assert L.wino_conv3x3_plan(1, 0, 14, 16, 64, 256, *args) != 0       # empty feature map
    assert L.wino_conv3x3_plan(1 << 20, 56, 56, 64, 64, 256, *args) != 0  # beyond 4 GiB


@pytest.mark.parametrize("header", ["winograd_mi355x.h", "util.h", "Kernel128_winograd.h", "Kernel256_winograd.h",
                                    "Kernel128_one.h", "Kernel256_one.h", "wino_data_files.h"])
def test_public_headers_compile_alone_as_c99_and_cxx(header, tmp_path):
    """A C host includes these headers with nothing else: each must be self-contained, strict C99
    (no HIP, no C++-isms) and equally valid C++ (extern "C" guards)."""
    inc = os.path.join(ROOT, "include")
    if not os.path.exists(os.path.join(inc, header)):
        pytest.skip(header + " not present")
    for compiler, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++17", "cpp")):
        cc = shutil.which(compiler)
        if cc is None:
            pytest.skip(compiler + " not available")
        src = tmp_path / ("use_header." + ext)
        src.write_text('#include "%s"\nint main(void) { return 0; }\n' % header)
        out = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I" + inc, str(src)],
                             capture_output=True, text=True)
        assert out.returncode == 0, (compiler, out.stderr[-1500:])
