"""Developer tool, run under `rocprofv3 --kernel-trace`: every small-batch case launches its kernel REPS times,
in a fixed order that is printed as JSON (one line per case: name + launches), so that tools/trace_split.py can
cut the kernel trace into cases by counting dispatches.  Kernel durations come from the trace, not from here
(back-to-back launches through Python are host-bound at ~10 us each).
    python tools/latency_cases.py [quick|full]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda:0")
L = pkg.lib()
REPS = 60
mode = sys.argv[1] if len(sys.argv) > 1 else "quick"


def knob(**kw):
    for k in ("WINO_3X3_ALGO", "WINO_SMALL_SPLIT", "WINO_SMALL_CT", "WINO_1X1_ALGO", "WINO_1X1_SMALL_KS", "WINO_1X1_SMALL_RT",
              "WINO_1X1_SMALL_CT"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)
    L.wino_debug_reload_knobs()


cases = []


def run(name, fn):
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    cases.append({"case": name, "launches": REPS})


def conv3(C, N, HW=14, **kw):
    w = (torch.rand(C, C, 3, 3) - 0.5).to(dev)
    s, b = (torch.rand(C) - 0.5).to(dev), (torch.rand(C) - 0.5).to(dev)
    U = pkg.filter_transform_f2(w)
    x = (torch.rand(N, HW + 2, HW + 2, C) - 0.5).to(dev)
    out = torch.empty(N, HW + 2, HW + 2, C, device=dev)
    torch.cuda.synchronize()
    knob(**kw)
    tag = " ".join("%s=%s" % (k.replace("WINO_", "").lower(), v) for k, v in kw.items()) or "auto"
    use, pr, sp, ct, wgs = pkg.small_plan_3x3_full(N, C, C, H=HW, W=HW)
    form = "small 1x%d pr%d s%d %d wgs" % (ct, pr, sp, wgs) if use else "big"
    size = "" if HW == 14 else " %dx%d" % (HW, HW)
    run("3x3%s C=%d N=%d [%s -> %s]" % (size, C, N, tag, form), lambda: pkg.conv3x3_bn_relu(x, U, b, s, out=out))
    knob()


def conv1(Cin, Kout, N, **kw):
    Bm = ((torch.rand(Cin, Kout) - 0.5) * 4).to(dev)
    s, b = (torch.rand(Kout) - 0.5).to(dev), (torch.rand(Kout) - 0.5).to(dev)
    A = ((torch.rand(N * 196, Cin) - 0.5) * 4).to(dev)
    out = torch.empty(N * 196, Kout, device=dev)
    torch.cuda.synchronize()
    knob(**kw)
    tag = " ".join("%s=%s" % (k.replace("WINO_", "").lower(), v) for k, v in kw.items()) or "auto"
    use, ks, rt, ct, wgs = pkg.small_plan_1x1_full(N * 196, Cin, Kout)
    form = "small %dx%d ks%d %d wgs" % (rt, ct, ks, wgs) if use else "big"
    run("1x1 %d->%d N=%d [%s -> %s]" % (Cin, Kout, N, tag, form), lambda: pkg.conv1x1_bn(A, Bm, b, s, True, out=out))
    knob()


def block(N, **kw):
    """The bottleneck block 1024 -> 256 -> 256 -> 1024 (3 launches per call)."""
    C4, Cm = 1024, 256
    x = (torch.rand(N, 14, 14, C4) - 0.5).to(dev)
    w1, w3 = ((torch.rand(C4, Cm) - 0.5) / 8).to(dev), ((torch.rand(Cm, C4) - 0.5) / 4).to(dev)
    U2 = pkg.filter_transform_f2(((torch.rand(Cm, Cm, 3, 3) - 0.5) / 12).to(dev))
    bn = [((torch.rand(c) - 0.5).to(dev), (torch.rand(c) + 0.5).to(dev)) for c in (Cm, Cm, C4)]
    out = torch.empty(N, 14, 14, C4, device=dev)
    ws = torch.empty(pkg.lib().wino_residual_block_workspace_bytes_hw(N, 14, 14, Cm) // 4, device=dev)
    torch.cuda.synchronize()
    knob(**kw)
    tag = " ".join("%s=%s" % (k.replace("WINO_", "").lower(), v) for k, v in kw.items()) or "auto"
    for _ in range(REPS):
        pkg.residual_block(x, w1, bn[0], U2, bn[1], w3, bn[2], out=out, workspace=ws)
    torch.cuda.synchronize()
    cases.append({"case": "block 1024->256->256->1024 N=%d [%s]" % (N, tag), "launches": 3 * REPS, "per_call": 3})
    knob()


small = lambda ct, sp: dict(WINO_3X3_ALGO="small", WINO_SMALL_CT=ct, WINO_SMALL_SPLIT=sp)
if mode == "s8":   # more workgroups than CUs at one image: two co-resident workgroups per CU, one round each
    for C, N in ((256, 1), (256, 2), (512, 1), (384, 1)):
        conv3(C, N)
        for ct, sp in ((1, 4), (1, 8), (2, 8)):
            conv3(C, N, **small(ct, sp))
    for c in cases:
        print(json.dumps(c), flush=True)
    sys.exit(0)
if mode == "policy":   # the automatic choice against the forced throughput kernel over a grid of shapes the model was not fitted on
    for C in (64, 128, 192, 256, 320, 384, 448, 512):
        for N in (1, 3, 7, 13, 21, 33, 47):
            conv3(C, N)
            conv3(C, N, WINO_3X3_ALGO="big")
    for c in cases:
        print(json.dumps(c), flush=True)
    sys.exit(0)
if mode == "stages":   # ResNet's four 3x3 stages at small batches: the latency kernel against the throughput kernel
    for HW, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
        for N in (1, 2, 4, 8):
            conv3(C, N, HW)
            conv3(C, N, HW, WINO_3X3_ALGO="big")
    for c in cases:
        print(json.dumps(c), flush=True)
    sys.exit(0)
if mode == "block":
    for N in (1, 2, 4, 8, 16, 32):
        block(N)
        block(N, WINO_1X1_ALGO="big")
        block(N, WINO_1X1_ALGO="big", WINO_3X3_ALGO="big")
    for c in cases:
        print(json.dumps(c), flush=True)
    sys.exit(0)
if mode == "explore1":   # the 1x1 latency form: every block shape and K-split against the tiled kernel
    for Cin, Kout in ((1024, 256), (512, 128), (128, 512), (256, 1024)):
        for N in (1, 2, 3, 4, 6, 8, 12, 16, 24):
            conv1(Cin, Kout, N)
            conv1(Cin, Kout, N, WINO_1X1_ALGO="big")
            for rt, ct in ((1, 1), (2, 2), (1, 2), (2, 1), (1, 4), (2, 4)):
                for ks in (4, 2, 1):
                    if Cin % (16 * ks) or Kout % ((4 // ks) * ct * 16) or (Cin // ks < 64 and ks > 1):
                        continue
                    if N >= 12 and (rt, ct) == (1, 1):
                        continue
                    wgs = -(-N * 196 // (16 * rt)) * (Kout // ((4 // ks) * ct * 16))
                    if wgs > 2048:
                        continue
                    conv1(Cin, Kout, N, WINO_1X1_ALGO="small", WINO_1X1_SMALL_KS=ks, WINO_1X1_SMALL_RT=rt, WINO_1X1_SMALL_CT=ct)
    for c in cases:
        print(json.dumps(c), flush=True)
    sys.exit(0)
if mode == "explore3":   # the 3x3 latency kernel's block widths and splits, one image up to the throughput kernel's range
    for C, Ns in ((256, (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 21, 24)), (128, (1, 2, 3, 4, 6, 8, 10, 12, 16, 20, 24, 32, 41, 42)),
                  (64, (1, 2, 4, 16, 41, 83, 84)), (192, (1, 3, 6, 10, 14, 20, 27, 28)), (384, (1, 2, 4, 6, 10, 13, 14)),
                  (512, (1, 2, 3, 5, 8, 10, 11))):
        nsuper = C // 16
        for N in Ns:
            conv3(C, N)
            conv3(C, N, WINO_3X3_ALGO="big")
            if C not in (128, 256) and N not in (Ns[0], Ns[1], Ns[len(Ns) // 2]):
                continue
            for ct in (1, 2, 4):
                if C % (16 * ct):
                    continue
                blocks = -(-N * 49 // 16) * (C // (16 * ct))
                if blocks > 300:
                    continue
                smax = max(1, min(256 // blocks, nsuper // 2, 8))
                for sp in sorted({smax, max(1, smax // 2)}):
                    conv3(C, N, WINO_3X3_ALGO="small", WINO_SMALL_CT=ct, WINO_SMALL_SPLIT=sp)
    for c in cases:
        print(json.dumps(c), flush=True)
    sys.exit(0)
if mode == "explore":   # beyond one round of blocks / workgroups: where do the latency forms stop paying?
    for N in (6, 8, 10, 12, 16, 20, 24):
        conv3(128, N, WINO_3X3_ALGO="big")
        for ct, sp in ((1, 1), (2, 1), (4, 1), (1, 2), (2, 2)):
            conv3(128, N, **small(ct, sp))
    for N in (3, 4, 5, 6, 8):
        conv3(256, N, WINO_3X3_ALGO="big")
        for ct, sp in ((1, 1), (2, 1), (4, 1), (2, 2), (4, 2), (4, 4)):
            conv3(256, N, **small(ct, sp))
    for Cin, Kout in ((1024, 256), (512, 128), (128, 512), (256, 1024)):
        for N in (1, 2, 3, 4, 6, 8, 12, 16):
            conv1(Cin, Kout, N, WINO_1X1_ALGO="big")
            for ks in (1, 2, 4):
                conv1(Cin, Kout, N, WINO_1X1_ALGO="small", WINO_1X1_SMALL_KS=ks)
    for c in cases:
        print(json.dumps(c), flush=True)
    sys.exit(0)
for C in (256, 128):
    for N in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32) if mode == "full" else (1, 2, 4, 16):
        conv3(C, N)
        conv3(C, N, WINO_3X3_ALGO="big")
    forms = [(1, 1), (1, 2), (1, 4), (1, 8), (2, 2), (2, 4), (2, 8), (4, 4), (4, 8)]
    for N in (1, 2, 3, 4) if mode == "full" else (1, 2):
        for ct, sp in forms:
            if 4 * sp <= (C // 16) * 2 and ((N * 49 + 15) // 16) * (C // (16 * ct)) * sp <= 512:
                conv3(C, N, **small(ct, sp))
for Cin, Kout in ((1024, 256), (512, 128), (128, 512), (256, 1024)):
    for N in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 40) if mode == "full" else (1, 2, 4):
        conv1(Cin, Kout, N)
        conv1(Cin, Kout, N, WINO_1X1_ALGO="big")
        conv1(Cin, Kout, N, WINO_1X1_ALGO="small")
for c in cases:
    print(json.dumps(c), flush=True)
