"""Developer tool (GPU box): soak of the stream-K hand-offs of both kernels.  For several shapes and forced
grids, launches the layer back to back and compares every result bitwise with the first of its
configuration (stream-K sums are added in segment order, so they must not move); under uneven load
(a second stream runs another layer concurrently).  usage: python tools/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("SOAK_LIB"): pkg.LIB_PATH = os.path.abspath(os.environ["SOAK_LIB"])   # A/B against another build of the library
L = pkg.lib()
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
os.environ["WINO_3X3_ALGO"] = "big"
os.environ["WINO_1X1_ALGO"] = "big"    # the tiled kernel's stream-K / split-K hand-offs are what is soaked (N = 2 would take the latency form)
torch.manual_seed(0)
cfgs = []
for (N, C, K, grid) in [(128, 256, 256, 0), (128, 128, 128, 0), (50, 64, 128, 56), (37, 128, 192, 200), (96, 256, 64, 97),
                        (20, 512, 256, 0), (7, 64, 64, 13), (128, 256, 256, 333), (3, 256, 256, 48)]:
    x = torch.rand(N, 16, 16, C, device=dev) - 0.5
    U = pkg.filter_transform_f2(torch.rand(K, C, 3, 3, device=dev) - 0.5)
    b, s = torch.rand(K, device=dev) - 0.5, torch.rand(K, device=dev) - 0.5
    cfgs.append([N, C, K, grid, x, U, b, s, None])
# 1x1 stream-K / split-K forms (M = N * 196 rows), automatic and forced grids (multiples of 8 and of the column blocks)
cfg1 = []
for (N, Cin, Kout, grid) in [(128, 1024, 256, 0), (128, 1024, 256, 512), (100, 512, 128, 0), (100, 512, 128, 320), (2, 1024, 256, 0),
                             (160, 1024, 256, 0), (60, 2048, 512, 768)]:
    A = torch.rand(N * 196, Cin, device=dev) - 0.5
    Bm = torch.rand(Cin, Kout, device=dev) - 0.5
    b, s = torch.rand(Kout, device=dev) - 0.5, torch.rand(Kout, device=dev) - 0.5
    cfg1.append([N, Cin, Kout, grid, A, Bm, b, s, None])
# 3x3 latency kernel: blocks shared by S workgroups through slabs + one ticket per workgroup (round 3); forms forced
cfgs_small = []
for (N, C, K, sp, ct) in [(1, 256, 256, 4, 1), (1, 128, 128, 4, 1), (2, 256, 256, 2, 1), (1, 256, 256, 8, 2), (3, 128, 128, 3, 1),
                          (2, 128, 128, 4, 2), (1, 64, 192, 2, 1), (4, 128, 128, 2, 1), (4, 256, 256, 2, 2), (6, 256, 256, 3, 4),
                          (2, 256, 256, 4, 2), (3, 128, 128, 4, 2), (1, 64, 192, 2, 4), (1, 512, 512, 8, 4)]:
    x = torch.rand(N, 16, 16, C, device=dev) - 0.5
    U = pkg.filter_transform_f2(torch.rand(K, C, 3, 3, device=dev) - 0.5)
    b, s = torch.rand(K, device=dev) - 0.5, torch.rand(K, device=dev) - 0.5
    cfgs_small.append([N, C, K, (sp, ct), x, U, b, s, None])
side = torch.cuda.Stream()
xs = torch.rand(64, 16, 16, 128, device=dev); Us = pkg.filter_transform_f2(torch.rand(128, 128, 3, 3, device=dev)); vs = torch.rand(128, device=dev)
use_side = os.environ.get("SOAK_SIDE", "1") != "0"
t0, launches, bad = time.time(), 0, 0
stats = {}
last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 60:   # (a GPU box takes seven silent minutes for a hang)
        last_note = time.time()
        print("soak: %d launches, %d differ, %.0f s" % (launches, bad, last_note - t0), flush=True)
    for c in cfgs_small:
        N, C, K, (sp, ct), x, U, b, s, ref = c
        os.environ.update(WINO_3X3_ALGO="small", WINO_SMALL_SPLIT=str(sp), WINO_SMALL_CT=str(ct))
        os.environ.pop("WINO_SK_GRID", None)
        L.wino_debug_reload_knobs()
        if use_side:
            with torch.cuda.stream(side):
                for _ in range(2): pkg.conv3x3_bn_relu(xs, Us, vs, vs)
        outs = [pkg.conv3x3_bn_relu(x, U, b, s) for _ in range(40)]
        launches += 40
        if ref is None:
            c[8] = outs[0].clone(); ref = c[8]
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                st = stats.setdefault(("small", N, C, K, sp, ct), [0, 0.0, 0, 0])
                d = (o - ref).abs()
                st[0] += 1; st[1] = max(st[1], float(d.max())); st[2] = max(st[2], int((d > 0).sum())); st[3] = max(st[3], int(torch.isnan(o).sum()))
    os.environ["WINO_3X3_ALGO"] = "big"
    for k in ("WINO_SMALL_SPLIT", "WINO_SMALL_CT"): os.environ.pop(k, None)
    for c in cfgs:
        N, C, K, grid, x, U, b, s, ref = c
        if grid: os.environ["WINO_SK_GRID"] = str(grid)
        else: os.environ.pop("WINO_SK_GRID", None)
        L.wino_debug_reload_knobs()
        if use_side:
            with torch.cuda.stream(side):          # uneven load from another stream (its own scratch)
                for _ in range(3): pkg.conv3x3_bn_relu(xs, Us, vs, vs)
        outs = [pkg.conv3x3_bn_relu(x, U, b, s) for _ in range(20)]
        launches += 20
        if ref is None:
            c[8] = outs[0].clone(); ref = c[8]
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                d = (o - ref).abs()
                key = (N, C, K, grid)
                st = stats.setdefault(key, [0, 0.0, 0, 0])
                st[0] += 1; st[1] = max(st[1], float(d.max())); st[2] = max(st[2], int((d > 0).sum())); st[3] = max(st[3], int(torch.isnan(o).sum()))
    for c in cfg1:
        N, Cin, Kout, grid, A, Bm, b, s, ref = c
        if grid: os.environ["WINO_1X1_SK_GRID"] = str(grid)
        else: os.environ.pop("WINO_1X1_SK_GRID", None)
        L.wino_debug_reload_knobs()
        if use_side:
            with torch.cuda.stream(side):
                for _ in range(3): pkg.conv3x3_bn_relu(xs, Us, vs, vs)
        outs = [pkg.conv1x1_bn(A, Bm, b, s, True) for _ in range(10)]
        launches += 10
        if ref is None:
            c[8] = outs[0].clone(); ref = c[8]
            want = torch.relu((A.double() @ Bm.double()) * s.double() + b.double())
            err = float((ref.double() - want).abs().max() / want.abs().max())
            assert err < 2e-5, ("1x1 first result wrong", N, Cin, Kout, grid, err)
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                st = stats.setdefault(("1x1", N, Cin, Kout, grid), [0, 0.0, 0, 0])
                d = (o - ref).abs()
                st[0] += 1; st[1] = max(st[1], float(d.max())); st[2] = max(st[2], int((d > 0).sum())); st[3] = max(st[3], int(torch.isnan(o).sum()))
    os.environ.pop("WINO_1X1_SK_GRID", None)
    torch.cuda.synchronize()
for k, v in stats.items(): print("  differs", k, "times", v[0], "max |diff|", v[1], "max elements", v[2], "nan", v[3])
print("ticket counters in use at the end:", pkg.tickets_in_use())
print("soak: %d launches in %.0f s, %d results differ from the first of their configuration" % (launches, time.time() - t0, bad))
sys.exit(1 if bad else 0)
