"""Developer tool (GPU box): A/B timing of several builds of the library, interleaved on one box.
usage: python tools/ab_layers.py libA.so libB.so [...]   (each loaded RTLD_LOCAL: own state, own kernels)
Per layer: 0.4 s of clock ramp, then 5 rounds; in every round each library runs 200 launches between two events;
the median round per library is printed (us per launch).  WINO_* knobs apply to all of them."""
import ctypes, os, sys, statistics
from ctypes import c_void_p as P, c_int, c_long
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
libs = []
import shutil, tempfile
tmpd = tempfile.mkdtemp()
for ci, spec in enumerate(sys.argv[1:]):
    # "lib.so" or "lib.so:KNOB=VAL,KNOB=VAL": every column gets its own copy of the library (own cached knobs)
    path, _, envs = spec.partition(":")
    copy = os.path.join(tmpd, f"c{ci}_" + (envs.replace("=", "").replace(",", "_").replace("WINO_", "")[-14:] or "dflt"), "lib.so")
    os.makedirs(os.path.dirname(copy)); shutil.copy(path, copy); path = copy
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    L = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
    L.wino_debug_reload_knobs()
    for kv in filter(None, envs.split(",")):
        os.environ.pop(kv.split("=")[0])
    L.wino_conv1x1_bn.argtypes = [P, P, P, P, P, c_long, c_int, c_int, c_int, P]
    L.wino_conv1x1_bn_ex.argtypes = [P, P, P, P, P, P, c_long, c_int, c_int, c_int, P]
    L.wino_conv3x3_bn_relu.argtypes = [P, P, P, P, P, c_int, c_int, c_int, c_int, P]
    L.wino_filter_transform_f2.argtypes = [P, P, c_int, c_int, P]
    L.wino_residual_block.argtypes = [P] * 11 + [c_int, c_int, c_int, P, ctypes.c_size_t, P]
    L.wino_residual_block_workspace_bytes.restype = ctypes.c_size_t
    L.wino_residual_block_workspace_bytes.argtypes = [c_int, c_int]
    libs.append((path, L))
st = lambda: P(torch.cuda.current_stream().cuda_stream)
def bench(name, fns, reps=200, rounds=5):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.4:
        for f in fns:
            for _ in range(20): f()
        torch.cuda.synchronize()
    res = [[] for _ in fns]
    for r in range(rounds):
        for i, f in enumerate(fns):
            for _ in range(20): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 1e3 / reps)
    print(f"{name:34s}" + "  ".join(f"{statistics.median(r):14.2f}" for r in res), flush=True)
print(f"{'us per launch':34s}" + "  ".join(f"{os.path.basename(os.path.dirname(os.path.abspath(p)))[-14:]:>14s}" for p, _ in libs))
N = int(os.environ.get("AB_N", "128"))
M = N * 196
rnd = lambda *s: (torch.rand(*s, device=dev) - 0.5)
only = os.environ.get("AB_ONLY", "")   # "3x3" / "1x1": just those layers
for (Cin, Kout, relu) in ((512, 128, 1), (128, 512, 0), (1024, 256, 1), (256, 1024, 0), (64, 256, 0), (2048, 512, 1)) if only not in ("3x3", "block") else ():
    A, B, b, s, C = rnd(M, Cin), rnd(Cin, Kout), rnd(Kout), rnd(Kout), torch.empty(M, Kout, device=dev)
    fns = [(lambda L=L: L.wino_conv1x1_bn(A.data_ptr(), B.data_ptr(), b.data_ptr(), s.data_ptr(), C.data_ptr(), M, Cin, Kout, relu, st())) for _, L in libs]
    outs = []
    for f in fns:
        C.fill_(float("nan")); assert f() == 0; outs.append(C.clone())
    same = all(torch.equal(o, outs[0]) for o in outs)
    bench(f"1x1 {Cin}->{Kout} N={N}{'' if same else '  (DIFFER)'}", fns)
if only not in ("3x3", "block"):
    # the block's last layer: 256 -> 1024 + skip + ReLU, A padded
    Apad, B, b, s, R, C = rnd(N, 16, 16, 256), rnd(256, 1024), rnd(1024), rnd(1024), rnd(M, 1024), torch.empty(M, 1024, device=dev)
    fns = [(lambda L=L: L.wino_conv1x1_bn_ex(Apad.data_ptr(), B.data_ptr(), b.data_ptr(), s.data_ptr(), R.data_ptr(), C.data_ptr(), M, 256, 1024, 1 | 2 | 8, st())) for _, L in libs]
    bench("1x1 256->1024 + skip (A padded)", fns)
    # the block's first layer: 1024 -> 256, C padded
    A, B, b, s, Cp = rnd(M, 1024), rnd(1024, 256), rnd(256), rnd(256), torch.empty(N, 16, 16, 256, device=dev)
    fns = [(lambda L=L: L.wino_conv1x1_bn_ex(A.data_ptr(), B.data_ptr(), b.data_ptr(), s.data_ptr(), None, Cp.data_ptr(), M, 1024, 256, 1 | 4, st())) for _, L in libs]
    bench("1x1 1024->256 (C padded)", fns)
if only in ("", "block"):
    C4, Cm = 1024, 256
    x, w1, w3 = rnd(M, C4), rnd(C4, Cm), rnd(Cm, C4)
    w2 = rnd(Cm, Cm, 3, 3); U2 = torch.empty(16 * Cm * Cm, device=dev)
    libs[0][1].wino_filter_transform_f2(w2.data_ptr(), U2.data_ptr(), Cm, Cm, st())
    v = [rnd(Cm) for _ in range(4)] + [rnd(C4) for _ in range(2)]
    outb = torch.empty(M, C4, device=dev)
    wsb = libs[0][1].wino_residual_block_workspace_bytes(N, Cm)
    ws = torch.empty(wsb // 4 + 64, device=dev)
    fns = [(lambda L=L: L.wino_residual_block(x.data_ptr(), w1.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), U2.data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                              w3.data_ptr(), v[4].data_ptr(), v[5].data_ptr(), outb.data_ptr(), N, C4, Cm, ws.data_ptr(), wsb, st())) for _, L in libs]
    outs = []
    for f in fns:
        outb.fill_(float("nan")); assert f() == 0; outs.append(outb.clone())
    same = all(torch.equal(o, outs[0]) for o in outs)
    bench(f"bottleneck block 1024/256 N={N}{'' if same else '  (DIFFER)'}", fns, reps=100)
for Cc in (128, 256) if only not in ("1x1", "block") else ():
    x, w, b, s = rnd(N, 16, 16, Cc), rnd(Cc, Cc, 3, 3), rnd(Cc), rnd(Cc)
    U, out = torch.empty(16 * Cc * Cc, device=dev), torch.empty(N, 16, 16, Cc, device=dev)
    libs[0][1].wino_filter_transform_f2(w.data_ptr(), U.data_ptr(), Cc, Cc, st())
    fns = [(lambda L=L: L.wino_conv3x3_bn_relu(x.data_ptr(), U.data_ptr(), b.data_ptr(), s.data_ptr(), out.data_ptr(), N, Cc, Cc, 1, st())) for _, L in libs]
    bench(f"3x3 {Cc}->{Cc} N={N}", fns)
