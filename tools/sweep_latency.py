"""Developer tool: the latency forms against the throughput kernels at small batches (one process; the knobs
are switched with wino_debug_reload_knobs).  Prints one line per (layer, N, form): microseconds per launch
from HIP events over back-to-back launches, median of 3 bursts.
    python tools/sweep_latency.py [3x3|1x1|all] [full]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda:0")
L = pkg.lib()


def knob(**kw):
    for k in ("WINO_3X3_ALGO", "WINO_SMALL_PR", "WINO_SMALL_SPLIT", "WINO_1X1_ALGO", "WINO_1X1_SK", "WINO_1X1_SK_GRID"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)
    L.wino_debug_reload_knobs()


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(best)[1]


def sweep_3x3(full):
    for C in (128, 256) + ((64, 512) if full else ()):
        w = (torch.rand(C, C, 3, 3) - 0.5).to(dev)
        s, b = (torch.rand(C) - 0.5).to(dev), (torch.rand(C) - 0.5).to(dev)
        U = pkg.filter_transform_f2(w)
        for N in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16) if full else (1, 2, 4, 5, 8, 16):
            x = (torch.rand(N, 16, 16, C) - 0.5).to(dev)
            out = torch.empty(N, 16, 16, C, device=dev)
            fn = lambda: pkg.conv3x3_bn_relu(x, U, b, s, out=out)
            knob()
            use, pr, sp, wgs = pkg.small_plan_3x3(N, C, C)
            row = {"auto(%s)" % ("small pr%d s%d" % (pr, sp) if use else "big"): timeit(fn)}
            knob(WINO_3X3_ALGO="big")
            row["big"] = timeit(fn)
            nsuper = C // 16
            blocks = ((N * 49 + 15) // 16) * (C // 16)
            for pr in (4, 2, 1):
                for sp in (1, 2, 4, 8):
                    if 4 * sp > nsuper * (4 // pr) or blocks * sp > 600 or (sp == 1 and pr != 4):
                        continue
                    knob(WINO_3X3_ALGO="small", WINO_SMALL_PR=pr, WINO_SMALL_SPLIT=sp)
                    row["pr%d s%d" % (pr, sp)] = timeit(fn)
            print("3x3 C=%d N=%d  " % (C, N) + "  ".join("%s %.1f" % kv for kv in row.items()), flush=True)
    knob()


def sweep_1x1(full):
    for Cin, Kout in ((1024, 256), (512, 128), (128, 512), (256, 1024)) + (((2048, 512), (64, 256)) if full else ()):
        Bm = ((torch.rand(Cin, Kout) - 0.5) * 4).to(dev)
        s, b = (torch.rand(Kout) - 0.5).to(dev), (torch.rand(Kout) - 0.5).to(dev)
        for N in (1, 2, 3, 4, 6, 8, 16) if full else (1, 2, 4, 8):
            A = ((torch.rand(N * 196, Cin) - 0.5) * 4).to(dev)
            out = torch.empty(N * 196, Kout, device=dev)
            fn = lambda: pkg.conv1x1_bn(A, Bm, b, s, True, out=out)
            knob()
            use, ks, wgs = pkg.small_plan_1x1(N * 196, Cin, Kout)
            row = {"auto(%s)" % ("small ks%d %d wgs" % (ks, wgs) if use else "big"): timeit(fn)}
            knob(WINO_1X1_ALGO="big")
            row["big"] = timeit(fn)
            knob(WINO_1X1_ALGO="small")
            use, ks, wgs = pkg.small_plan_1x1(N * 196, Cin, Kout)
            row["small ks%d %d wgs" % (ks, wgs)] = timeit(fn)
            print("1x1 %d->%d N=%d  " % (Cin, Kout, N) + "  ".join("%s %.1f" % kv for kv in row.items()), flush=True)
    knob()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    full = len(sys.argv) > 2
    if what in ("3x3", "all"):
        sweep_3x3(full)
    if what in ("1x1", "all"):
        sweep_1x1(full)
