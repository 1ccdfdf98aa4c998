import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
if os.environ.get("SOAK_LIB"): pkg.LIB_PATH = os.path.abspath(os.environ["SOAK_LIB"])
L = pkg.lib()
dev = torch.device("cuda:0")
os.environ["WINO_3X3_ALGO"] = "big"; L.wino_debug_reload_knobs()
torch.manual_seed(0)
N, C, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.rand(N, 16, 16, C, device=dev) - 0.5
w = torch.rand(K, C, 3, 3, device=dev) - 0.5
U = pkg.filter_transform_f2(w)
b, s = torch.rand(K, device=dev) - 0.5, torch.rand(K, device=dev) - 0.5
direct = pkg.conv3x3_direct(x, w, b, s)
for grid in [int(g) for g in sys.argv[4:]]:
    if grid: os.environ["WINO_SK_GRID"] = str(grid)
    else: os.environ.pop("WINO_SK_GRID", None)
    L.wino_debug_reload_knobs()
    res = []
    for rep in range(4):
        out = torch.full((N, 16, 16, K), float("nan"), device=dev)
        pkg.conv3x3_bn_relu(x, U, b, s, out=out)
        torch.cuda.synchronize()
        nn = int(torch.isnan(out).sum())
        res.append(nn if nn else round(float((out - direct).abs().max()), 7))
    print("grid", grid, res)
