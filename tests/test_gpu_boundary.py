"""GPU tests of the boundary (run with `-m gpu`): the reference-shaped caller of INTEGRATION.md section 1 RUN
against the library, the reference's stdout protocol on request, the entry points' outputs against
the fp64 ORACLE (not only the GPU comparator), the CPU-baseline line of ./Test, the threaded
multi-GPU path of the C driver exercised on one GPU through the device-alias knob, and the
library-owned scratch surviving a graph captured before it grew."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_bin
from test_boundary import build_reference_shaped_caller
from test_gpu_parity import REL, TIGHT, _rand_layer, _ring, _t, torch_dev  # noqa: F401

pytestmark = pytest.mark.gpu

LAYERS = ["kernel_128", "kernel_256", "kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in", "kernel_256_1_out"]


def _golden_for(O, golden_outputs, name):
    return golden_outputs[name]


@pytest.mark.parametrize("mode", [0, 5])
def test_reference_shaped_caller_runs_with_the_reference_stdout(mode, data_dir, tmp_path):
    """The caller of tests/test_boundary.py (reference-shaped main, built per INTEGRATION.md section 1) on the
    reference data set with WINO_STDOUT_COMPAT=1: exactly the reference's lines, in its order
    (Test.c:23,50-53; Kernel128_winograd.cu:270,275,404,409; util.c:62)."""
    exe = build_reference_shaped_caller(str(tmp_path))
    env = dict(os.environ, WINO_STDOUT_COMPAT="1")
    r = subprocess.run([exe, str(mode), "4"], cwd=data_dir, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 4 * 6 + 1, r.stdout
    for i in range(4):
        blk = lines[6 * i:6 * i + 6]
        assert blk[0] == "---- Iter: %d ----" % i
        assert re.fullmatch(r"TotalTime = \d+ us", blk[1])
        assert blk[2] == "cudaSuccess"
        assert re.fullmatch(r"cuDNN TotalTime = \d+ us", blk[3])
        assert blk[4] == "cudaSuccess"
        m = re.fullmatch(r"\[max_error: ([0-9.]+)\]\[error_cnt: (\d+)\]", blk[5])
        assert m, blk[5]
        if mode == 0:      # 3x3 outputs are O(1): the reference's absolute 1e-5 checker is meaningful
            assert float(m.group(1)) < 1e-4
    assert re.fullmatch(r"Average Total Time: \[Mine: \d+ us\], \[cuDNN: \d+ us\]", lines[-1])


def test_default_stdout_names_the_real_status_and_comparator(data_dir):
    exe = os.path.join(ROOT, "Test")
    env = {k: v for k, v in os.environ.items() if k != "WINO_STDOUT_COMPAT"}
    r = subprocess.run([exe, "2", "1", "1", "3"], cwd=data_dir, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert lines[2] == "hipSuccess" and lines[3].startswith("Direct TotalTime = ") and lines[4] == "hipSuccess"
    assert any(l.startswith("Average Total Time: [Mine: ") and "[Direct: " in l for l in lines)


@pytest.mark.parametrize("mode,N", [(1, 1), (0, 24), (4, 8)])
def test_test_binary_prints_the_cpu_baseline(mode, N, data_dir):
    """BASELINE.md section 4 / SURVEY.md section 8d: the same layer as a naive im2col + SGEMM on the host cores, timed
    in the same ./Test invocation, core count stated, and agreeing with the GPU output."""
    exe = os.path.join(ROOT, "Test")
    r = subprocess.run([exe, str(mode), str(N), "1", "3"], cwd=data_dir, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("CPU baseline (naive im2col+SGEMM+BN, ")]
    assert len(line) == 1, r.stdout[-2000:]
    m = re.search(r"(\d+) host threads, (\d+) reps\): (\d+) us, ([0-9.]+) GFLOP/s; max \|GPU - CPU\| = (\S+) \((\S+) relative\)", line[0])
    assert m, line[0]
    assert int(m.group(1)) >= 1 and int(m.group(2)) >= 1 and float(m.group(4)) > 0
    assert float(m.group(6)) < TIGHT, line[0]
    js = json.loads(r.stdout.splitlines()[-1])
    assert js["cpu_threads"] == int(m.group(1)) and js["cpu_baseline_us"] > 0 and js["N"] == N
    assert js["gpu_vs_cpu_max_rel_diff"] < TIGHT


@pytest.mark.parametrize("mode", range(6))
def test_entry_point_outputs_match_the_oracle(mode, data_dir, pkg, O, golden_outputs):
    """The six reference entry points at N = 1 on the reference's seed-0 files: the output the driver
    copied back against the fp64 golden vectors (tests/golden), not only against the GPU comparator."""
    L = pkg.lib()
    cwd = os.getcwd()
    os.chdir(data_dir)
    try:
        L.wino_driver_set_quiet(1)
        L.wino_driver_set_batch(1)
        L.wino_driver_set_gpus(1)
        getattr(L, LAYERS[mode])()
        n = ctypes.c_size_t()
        p = L.wino_driver_last_output(ctypes.byref(n))
        got = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    finally:
        L.wino_driver_set_quiet(0)
        os.chdir(cwd)
    want = golden_outputs[LAYERS[mode]]
    if mode < 2:
        K = 128 if mode == 0 else 256
        got = got.reshape(1, 16, 16, K)
        assert (got[:, _ring(), :] == 0).all()
        got = got[:, 1:15, 1:15, :]
    assert O.rel_error(got.reshape(want.shape), want) < TIGHT


def test_entry_points_from_two_host_threads(data_dir, pkg, O, golden_outputs):
    """The driver keeps the last call's result and tensors per calling thread (the reference is single-threaded,
    Test.c:13-56): two host threads call different entry points at the same time, each reads back ITS call's
    output, and both match the golden vectors."""
    import threading
    L = pkg.lib()
    cwd = os.getcwd()
    os.chdir(data_dir)
    got, errs = {}, []

    def work(mode):
        try:
            for _ in range(3):
                getattr(L, LAYERS[mode])()
                n = ctypes.c_size_t()
                p = L.wino_driver_last_output(ctypes.byref(n))
                got[mode] = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
                res = pkg.DriverResult()
                assert L.wino_driver_last_result(ctypes.byref(res)) == 0
                assert res.N == 1 and res.max_rel_err < TIGHT
        except Exception as e:   # noqa: BLE001 -- reported in the main thread
            errs.append((mode, repr(e)))

    try:
        L.wino_driver_set_quiet(1)
        L.wino_driver_set_batch(1)
        L.wino_driver_set_gpus(1)
        ts = [threading.Thread(target=work, args=(m,)) for m in (2, 5, 0)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        L.wino_driver_set_quiet(0)
        os.chdir(cwd)
    assert not errs, errs
    for mode in (2, 5):
        want = golden_outputs[LAYERS[mode]]
        assert O.rel_error(got[mode].reshape(want.shape), want) < TIGHT, mode
    g0 = got[0].reshape(1, 16, 16, 128)
    assert O.rel_error(g0[:, 1:15, 1:15, :], golden_outputs[LAYERS[0]].reshape(1, 14, 14, 128)) < TIGHT


@pytest.mark.parametrize("mode,N,G", [(1, 8, 2), (1, 9, 4), (4, 6, 2), (0, 128, 2), (1, 128, 8)])
def test_multi_gpu_driver_path_on_one_gpu(mode, N, G, data_dir, pkg, O):
    """layer_driver.c's batch split -- one host thread and one stream per job, common start barrier,
    per-job slices of the host tensors, (last finish - first start) timing -- run with G jobs aliased
    onto the one visible GPU (WINO_GPUS_ALIAS).  Every image must come out as in the G = 1 run (to
    fp32 summation order: a different per-job batch can take a different launch decomposition), the
    comparator diff must be clean, and the reported time must cover all jobs.  (1, 128, 8) is the headline
    layer as an 8-GPU node would split it: eight jobs of 16 images, eight threads and streams in one process.)"""
    L = pkg.lib()
    cwd = os.getcwd()
    os.chdir(data_dir)

    def run(g):
        L.wino_driver_set_gpus(g)
        getattr(L, LAYERS[mode])()
        res = pkg.DriverResult()
        assert L.wino_driver_last_result(ctypes.byref(res)) == 0
        n = ctypes.c_size_t()
        p = L.wino_driver_last_output(ctypes.byref(n))
        return res, np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    try:
        L.wino_driver_set_quiet(1)
        L.wino_driver_set_batch(N)
        L.wino_driver_set_gpu_alias(1)
        r1, out1 = run(1)
        rG, outG = run(G)
    finally:
        L.wino_driver_set_gpu_alias(0)
        L.wino_driver_set_gpus(1)
        L.wino_driver_set_batch(1)
        L.wino_driver_set_quiet(0)
        os.chdir(cwd)
    assert r1.gpus == 1 and rG.gpus == G and rG.N == N
    assert rG.max_rel_err < TIGHT and r1.max_rel_err < TIGHT
    assert out1.shape == outG.shape and np.isfinite(outG).all()
    assert np.abs(outG - out1).max() <= 4e-6 * np.abs(out1).max()
    assert rG.mine_us > 0 and rG.steady_us > 0


def test_multi_gpu_request_without_alias_fails_loudly(data_dir):
    """Without the alias knob a request for more GPUs than visible is an error, not a silent fallback."""
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("box has several GPUs")
    exe = os.path.join(ROOT, "Test")
    env = {k: v for k, v in os.environ.items() if k != "WINO_GPUS_ALIAS"}
    r = subprocess.run([exe, "0", "4", "2", "3"], cwd=data_dir, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) requested" in r.stdout


def test_test_binary_multi_gpu_alias(data_dir):
    """The same through the command line: ./Test 1 8 2 3 with WINO_GPUS_ALIAS=1."""
    exe = os.path.join(ROOT, "Test")
    env = dict(os.environ, WINO_GPUS_ALIAS="1", WINO_CPU_BASELINE="0")
    r = subprocess.run([exe, "1", "8", "2", "3"], cwd=data_dir, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    js = json.loads(r.stdout.splitlines()[-1])
    assert js["gpus"] == 2 and js["N"] == 8 and js["error_cnt_1e-5"] == 0 and js["max_rel_err"] < TIGHT


def test_graph_captured_before_the_scratch_grew_still_replays(pkg, O, torch_dev, knobs):
    """The library-owned stream-K scratch is never freed or moved while its stream lives: capture a
    small layer (its slab and ticket pointers are baked into the graph), then run a layer that needs
    more scratch on the same stream eagerly, then replay the graph -- bitwise the eager result, and
    the oracle's on a sample."""
    torch, dev = torch_dev
    rng = np.random.RandomState(91)
    N, C, K = 40, 64, 64
    x, w, s, b = _rand_layer(rng, N, C, K)
    xt, wt, st, bt = (_t(torch_dev, a) for a in (x, w, s, b))
    U = pkg.filter_transform_f2(wt)
    knobs.set("WINO_3X3_ALGO", "big")
    knobs.set("WINO_SK_GRID", "56")                  # everything is stream-K tail: the slabs are in use
    sg = torch.cuda.Stream()
    out = torch.empty((N, 16, 16, K), device=dev)
    with torch.cuda.stream(sg):
        pkg.conv3x3_prepare(N, C, K)
        eager = pkg.conv3x3_bn_relu(xt, U, bt, st).clone()
    sg.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=sg):
        pkg.conv3x3_bn_relu(xt, U, bt, st, out=out)
    # a shape whose slabs (2 * 2048 x 64 KB = 256 MiB) and ticket counters (> 4096) exceed the first allocation
    knobs.set("WINO_SK_GRID", "2048")
    N2, C2, K2 = 700, 64, 64
    x2 = (torch.rand(N2, 16, 16, C2, device=dev) - 0.5)
    w2 = (rng.rand(K2, C2, 3, 3) - 0.5).astype(np.float32)
    U2 = pkg.filter_transform_f2(_t(torch_dev, w2))
    with torch.cuda.stream(sg):
        big = pkg.conv3x3_bn_relu(x2, U2, bt, st)
    sg.synchronize()
    want2 = O.conv3x3_bn_relu_direct(x2[:2].cpu().numpy(), w2, s, b)
    assert O.rel_error(big[:2].cpu().numpy(), want2) < TIGHT
    for _ in range(3):
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
    idx = [0, 17, 39]
    assert O.rel_error(out[idx].cpu().numpy(), O.conv3x3_bn_relu_direct(x[idx], w, s, b)) < TIGHT
    # ... and eager launches of the small shape on the grown scratch agree too
    knobs.set("WINO_SK_GRID", "56")
    with torch.cuda.stream(sg):
        again = pkg.conv3x3_bn_relu(xt, U, bt, st)
    sg.synchronize()
    assert torch.equal(again, eager)


def test_residual_block_prepare_then_capture(pkg, O, torch_dev):
    """wino_residual_block_prepare allocates all three launches' scratch: the block captures right
    after it, with no per-layer prepare calls."""
    torch, dev = torch_dev
    rng = np.random.RandomState(12)
    N, C4, Cm = 20, 512, 128
    x = (rng.rand(N, 14, 14, C4) - 0.5).astype(np.float32)
    w1 = ((rng.rand(C4, Cm) - 0.5) / np.sqrt(C4) * 4).astype(np.float32)
    w2 = ((rng.rand(Cm, Cm, 3, 3) - 0.5) / np.sqrt(9 * Cm) * 4).astype(np.float32)
    w3 = ((rng.rand(Cm, C4) - 0.5) / np.sqrt(Cm) * 4).astype(np.float32)
    bn = [((rng.rand(c) - 0.5).astype(np.float32), (rng.rand(c) + 0.5).astype(np.float32)) for c in (Cm, Cm, C4)]
    t = lambda a: _t(torch_dev, a)
    bnt = [(t(a), t(b)) for a, b in bn]
    xt, w1t, w3t, U2 = t(x), t(w1), t(w3), pkg.filter_transform_f2(t(w2))
    sg = torch.cuda.Stream()
    out = torch.empty_like(xt)
    ws = torch.empty(pkg.lib().wino_residual_block_workspace_bytes(N, Cm) // 4, device=dev)
    with torch.cuda.stream(sg):
        pkg.residual_block_prepare(N, C4, Cm)
        # (kernel attributes are set at each kernel's first launch: one eager pass before the capture)
        eager = pkg.residual_block(xt, w1t, bnt[0], U2, bnt[1], w3t, bnt[2], workspace=ws).clone()
    sg.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=sg):
        pkg.residual_block(xt, w1t, bnt[0], U2, bnt[1], w3t, bnt[2], out=out, workspace=ws)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    assert O.rel_error(out.cpu().numpy(), O.residual_block(x, w1, bn[0], w2, bn[1], w3, bn[2])) < TIGHT


def test_caller_supplied_buffers_are_validated(pkg, torch_dev):
    """A wrong-sized, strided or wrong-dtype `out` / `workspace` would be an out-of-bounds GPU write:
    every wrapper rejects it before the launch."""
    torch, dev = torch_dev
    z = lambda *s: torch.zeros(*s, device=dev)
    A, B, v = z(196, 64), z(64, 128), z(128)
    with pytest.raises(pkg.WinoError):
        pkg.conv1x1_bn(A, B, v, v, True, out=z(196, 64))                      # wrong shape
    with pytest.raises(pkg.WinoError):
        pkg.conv1x1_bn(A, B, v, v, True, out=z(196, 256)[:, ::2])              # not contiguous
    with pytest.raises(pkg.WinoError):
        pkg.conv1x1_bn(A, B, v, v, True, out=torch.zeros(196, 128, device=dev, dtype=torch.float64))
    with pytest.raises(pkg.WinoError):
        pkg.conv1x1_bn_ex(A, B, v, v, pkg.RELU | pkg.C_PADDED, out=z(1, 16, 16, 64))
    with pytest.raises(pkg.WinoError):
        pkg.conv1x1_bn_ex(A, B, v, v, pkg.RELU | pkg.ADD_RESIDUAL, residual=z(196, 64))
    x = z(2, 14, 14, 128)
    w1, w3, U2 = z(128, 64), z(64, 128), z(16 * 64 * 64)
    bn = (z(64), z(64))
    bn3 = (z(128), z(128))
    with pytest.raises(pkg.WinoError):
        pkg.residual_block(x, w1, bn, U2, bn, w3, bn3, workspace=z(1000))    # too small
    with pytest.raises(pkg.WinoError):
        pkg.residual_block(x, w1, bn, U2, bn, w3, bn3, out=z(2, 14, 14, 64))
    with pytest.raises(pkg.WinoError):
        pkg.conv3x3_bn_relu(z(1, 16, 16, 64), U2, z(64), z(64), out=z(1, 16, 16, 128))
