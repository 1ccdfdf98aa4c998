"""Developer tool (run on the GPU box): does the 3x3 launch policy still pick the fastest grid?
For each (C, N): the automatic choice against forced grids (WINO_SK_GRID through the knob reload)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
L = pkg.lib()
dev = torch.device("cuda:0")

def t_us(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps

def plan(N, C):
    g, r, it = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(); t = ctypes.c_long()
    L.wino_conv3x3_plan(N, 14, 14, C, C, 256, ctypes.byref(g), ctypes.byref(r), ctypes.byref(t), ctypes.byref(it))
    return g.value

os.environ["WINO_3X3_ALGO"] = "big"
# clock ramp
x = torch.rand(128, 16, 16, 256, device=dev) - 0.5
U = pkg.filter_transform_f2(torch.rand(256, 256, 3, 3, device=dev) - 0.5)
bv = torch.rand(256, device=dev)
out = torch.empty(128, 16, 16, 256, device=dev)
for _ in range(3000): pkg.conv3x3_bn_relu(x, U, bv, bv, out=out)
torch.cuda.synchronize()
for C in ([int(a) for a in sys.argv[1:]] or [256, 128]):
    U = pkg.filter_transform_f2(torch.rand(C, C, 3, 3, device=dev) - 0.5)
    bv = torch.rand(C, device=dev)
    for N in [int(a) for a in os.environ.get("SWEEP_N", "8,12,16,24,32,40,48,64,80,96,112,160,192").split(",")]:
        x = torch.rand(N, 16, 16, C, device=dev) - 0.5
        out = torch.empty(N, 16, 16, C, device=dev)
        fn = lambda: pkg.conv3x3_bn_relu(x, U, bv, bv, out=out)
        os.environ.pop("WINO_SK_GRID", None); L.wino_debug_reload_knobs()
        auto_g = plan(N, C)
        res = {"auto(%d)" % auto_g: t_us(fn)}
        items = -(-N * 49 // 64) * (C // 64)
        cands = {256, items if items <= 256 else 256, max(8, items // 2) if items // 2 <= 256 else 256}
        nch = C // 8
        for s in (4, 8, 16):
            if nch % s == 0 and items * nch // s <= 256: cands.add(items * nch // s)
        for g in sorted(cands):
            if g == auto_g or g < 8: continue
            os.environ["WINO_SK_GRID"] = str(g); L.wino_debug_reload_knobs()
            res["G=%d" % g] = t_us(fn)
        best = min(res, key=res.get)
        print("C=%d N=%3d items=%3d  " % (C, N, items) + "  ".join("%s %.1f" % kv for kv in res.items()) + ("   <-- policy loses %.1f%%" % (100 * (res["auto(%d)" % auto_g] / res[best] - 1)) if not best.startswith("auto") and res["auto(%d)" % auto_g] > 1.02 * res[best] else ""))
