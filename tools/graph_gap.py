"""Developer tool: why a HIP-graph replay of the bottleneck block takes longer than its three eager launches.
Run under rocprofv3 --kernel-trace (tools/graph_gap.sh): 60 eager blocks back to back, then 60 replays of the captured
block; with a trace directory as argument it reads the kernel trace back and prints, per phase, the three kernels'
durations and the idle gaps between consecutive kernels (inside a block and from one block to the next).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/graph_gap.py
    python3 tools/graph_gap.py DIR"""
import csv
import glob
import os
import statistics
import sys

REPS = 60
HOT = ("wino_f2_fused_kernel", "conv1x1_bn_kernel")

if len(sys.argv) > 1:
    f = max(glob.glob(os.path.join(sys.argv[1], "**/*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))
                  if any(h in r["Kernel_Name"] for h in HOT))
    rows = rows[-6 * REPS:]                     # the two measured phases (warm-up launches come first)
    for name, chunk in (("eager", rows[:3 * REPS]), ("graph replay", rows[3 * REPS:])):
        chunk = chunk[30:]                      # drop the first ten blocks of a phase
        dur = [[], [], []]
        gap = [[], [], []]                      # gap AFTER kernel i (i = 2: to the next block's first kernel)
        for j in range(len(chunk) - 1):
            i = j % 3
            dur[i].append((chunk[j][1] - chunk[j][0]) / 1e3)
            gap[i].append((chunk[j + 1][0] - chunk[j][1]) / 1e3)
        per_block = (chunk[-1][0] - chunk[0][0]) / 1e3 / ((len(chunk) - 1) / 3)
        print("%-13s block period %.1f us | kernels %s us | gaps after k1 / k2 / k3(to next block) %s us" % (
            name, per_block, " / ".join("%.1f" % statistics.median(d) for d in dur),
            " / ".join("%.2f" % statistics.median(g) for g in gap)))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda:0")
N, C, K, H = 128, 1024, 256, 14
g = torch.Generator(device="cpu").manual_seed(1)
rnd = lambda *shape, scale=1.0: ((torch.rand(*shape, generator=g) - 0.5) * scale).to(dev)
x = rnd(N, H, H, C)
w1, w3 = rnd(C, K, scale=4.0 / C ** 0.5), rnd(K, C, scale=4.0 / K ** 0.5)
U2 = pkg.filter_transform_f2(rnd(K, K, 3, 3, scale=4.0 / (9 * K) ** 0.5))
bn1, bn2, bn3 = (rnd(K), rnd(K) + 1.0), (rnd(K), rnd(K) + 1.0), (rnd(C), rnd(C) + 1.0)
out = torch.empty_like(x)
ws = torch.empty(pkg.lib().wino_residual_block_workspace_bytes_hw(N, H, H, K) // 4, device=dev)
step = lambda: pkg.residual_block(x, w1, bn1, U2, bn2, w3, bn3, out=out, workspace=ws)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    pkg.residual_block_prepare(N, C, K)
    for _ in range(200):                        # clock ramp + warm-up
        step()
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        step()
    for _ in range(20):
        graph.replay()
    stream.synchronize()
    for _ in range(REPS):
        step()
    stream.synchronize()
    for _ in range(REPS):
        graph.replay()
    stream.synchronize()
