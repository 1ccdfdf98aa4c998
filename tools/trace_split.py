"""Cut a rocprofv3 kernel trace into the cases tools/latency_cases.py printed (in order, by counting the
dispatches of the product kernels) and print mean / median / min kernel duration per case.
    python tools/trace_split.py <trace dir> <cases.jsonl> [out.json]"""
import csv
import glob
import json
import os
import statistics
import sys

HOT = ("wino_f2_fused_kernel", "wino_f2_small_kernel", "conv1x1_bn_kernel", "conv1x1_small_kernel")
trace_dir, cases_path = sys.argv[1], sys.argv[2]
f = max(glob.glob(os.path.join(trace_dir, "**/*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    if any(h in r["Kernel_Name"] for h in HOT):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")),
                     r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
rows.sort()
cases = [json.loads(l) for l in open(cases_path) if l.startswith("{")]
need = sum(c["launches"] for c in cases)
if need != len(rows):
    print("WARNING: %d hot dispatches in the trace, %d launches in the case list" % (len(rows), need))
out, i = [], 0
for c in cases:
    chunk = rows[i:i + c["launches"]]
    i += c["launches"]
    if not chunk:
        break
    pc = c.get("per_call", 1)                              # launches per call (the bottleneck block: 3): durations summed per call
    d = [sum((e - s) / 1e3 for s, e, *_ in chunk[j:j + pc]) for j in range(10 * pc, len(chunk) - pc + 1, pc)]   # the first ten calls of a case are its warm-up
    gap = [(chunk[j + 1][0] - chunk[j][1]) / 1e3 for j in range(10, len(chunk) - 1)]
    name = chunk[-1][2]
    short = next(h for h in HOT if h in name) + (name[name.index("<"):name.index(">") + 1] if "<" in name else "")
    e = {"case": c["case"], "kernel": short[:60], "grid": "x".join(x for x in chunk[-1][3:6] if x), "wg": chunk[-1][6],
         "mean_us": round(statistics.mean(d), 2), "median_us": round(statistics.median(d), 2), "min_us": round(min(d), 2),
         "median_gap_us": round(statistics.median(gap), 2) if gap else None}
    out.append(e)
    print("%-52s %-42s grid %-12s mean %6.2f  median %6.2f  min %6.2f  gap %s" % (e["case"], e["kernel"], e["grid"], e["mean_us"], e["median_us"], e["min_us"], e["median_gap_us"]))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
