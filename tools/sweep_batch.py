"""Developer tool (GPU box): the six reference layers, the bottleneck block and ResNet's other stages against batch size, automatic
launch forms, clock ramped; one JSON line per point (us per launch from HIP events, effective TFLOP/s).
usage: python tools/sweep_batch.py > gpurun_out/.../bench_batch.jsonl"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
rnd = lambda *s: torch.rand(*s, device=dev) - 0.5

def t_us(fn, reps):
    t0 = time.time()
    while time.time() - t0 < 0.25:            # clock ramp on this very shape
        for _ in range(20): fn()
        torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(best)[1]

Ns = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512)
for C in (256, 128):
    w, b, s = rnd(C, C, 3, 3), rnd(C), rnd(C)
    U = pkg.filter_transform_f2(w)
    for N in Ns:
        x, out = rnd(N, 16, 16, C), torch.empty(N, 16, 16, C, device=dev)
        us = t_us(lambda: pkg.conv3x3_bn_relu(x, U, b, s, out=out), 200 if N <= 128 else 60)
        fl = 2.0 * N * 196 * C * C * 9
        print(json.dumps({"layer": f"conv3x3_{C}", "N": N, "us": round(us, 2), "eff_tflops": round(fl / us / 1e6, 1),
                          "executed_mfma_frac": round(fl / 2.25 / us / 1e6 / 157.3, 3)}), flush=True)
for (Cin, Kout, relu) in ((1024, 256, True), (256, 1024, False), (512, 128, True), (128, 512, False)):
    Bm, b, s = rnd(Cin, Kout), rnd(Kout), rnd(Kout)
    for N in Ns:
        A, out = rnd(N * 196, Cin), torch.empty(N * 196, Kout, device=dev)
        us = t_us(lambda: pkg.conv1x1_bn(A, Bm, b, s, relu, out=out), 200 if N <= 128 else 60)
        fl = 2.0 * N * 196 * Cin * Kout
        print(json.dumps({"layer": f"conv1x1_{Cin}_{Kout}", "N": N, "us": round(us, 2), "eff_tflops": round(fl / us / 1e6, 1),
                          "frac_of_peak": round(fl / us / 1e6 / 157.3, 3)}), flush=True)
