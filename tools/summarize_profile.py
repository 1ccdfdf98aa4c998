#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into profiles/<tag>/:
  kernel_stats.csv   per config and kernel: calls, avg/min/max ns from the kernel trace (rocprofv3 --stats),
                     next to bench.py's HIP-event kernel time in the same (profiled) run and in an unprofiled run
  summary.json       per config and kernel: PMC means per launch and the figures derived from them --
                     HBM bytes (FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md section HBM
                     prescribes for gfx950, WRITE_SIZE as is), achieved GB/s against 8 TB/s, L2 hit rate,
                     matrix-pipe busy fraction, executed fp32 MFMA FLOPs, shader clock under the profiler
  pmc_traffic.json   (profiles/) HBM bytes per launch of each config's dominant kernel, read by bench.py
Kernels that only prepare data (filter transforms, torch fills) are left out."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = f"gpurun_out/prof_{tag}"
dst = f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
HOT = ("wino_f2_fused_kernel", "wino_f2_small_kernel", "conv1x1_bn_kernel", "conv1x1_small_kernel",
       "f4_input_transform_kernel", "f4_output_transform_kernel", "f4_ring_kernel")


def short(name):
    for h in HOT:
        if h in name:
            if h == "conv1x1_bn_kernel":      # keep the variant: <BK, NW, ABLATE, SK>
                m = re.search(r"conv1x1_bn_kernel<([^>]*)>", name)
                if m:
                    a = [x.strip() for x in m.group(1).split(",")]
                    return "conv1x1_bn_kernel<%sw%s>" % (a[1] if len(a) > 1 else "?", ",streamK" if len(a) > 3 and a[3] in ("true", "1") else "")
            if h == "wino_f2_small_kernel":   # keep the form: <CT>
                m = re.search(h + r"<\s*(\d+)", name)
                return h + ("<%s>" % m.group(1) if m else "")
            if h == "conv1x1_small_kernel":   # <KS, RT, CT>
                m = re.search(h + r"<\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)", name)
                return h + ("<%s,%s,%s>" % m.groups() if m else "")
            if h == "wino_f2_fused_kernel" and re.search(r"wino_f2_fused_kernel<\s*16\s*,", name):
                return "wino_f2_fused_kernel<stamped diagnostic build: bench.py's clock probe>"
            return h
    return None


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def bench_line(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return {}


summary, traffic, rows = {}, {}, []
for d in sorted(glob.glob(os.path.join(src, "*/"))):
    cfg = os.path.basename(d.rstrip("/"))
    if os.path.exists(os.path.join(d, "FAILED")):   # tools/profile.sh: a pass of this config failed
        print("skipping %s: failed passes %s" % (cfg, open(os.path.join(d, "FAILED")).read().split()))
        continue
    S = summary.setdefault(cfg, {"kernels": {}})
    bt, bu = bench_line(os.path.join(d, "bench_trace.json")), bench_line(os.path.join(d, "bench_unprofiled.json"))
    S["bench_us_per_step_under_rocprof"] = bt.get("us_per_layer")
    S["bench_us_per_step_unprofiled"] = bu.get("us_per_layer")
    S["bench_kernel_us_unprofiled"] = bu.get("roofline", {}).get("kernel_us")
    S["clock_ghz_unprofiled"] = bu.get("roofline", {}).get("clock_ghz")
    S["workload"] = bu.get("config", {}).get("workload") or bt.get("config", {}).get("workload")
    f = newest(os.path.join(d, "trace/**/*_kernel_stats.csv"))
    if f:
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if not k:
                continue
            e = S["kernels"].setdefault(k, {})
            e["calls"] = e.get("calls", 0) + int(r["Calls"])
            e["trace_avg_us"] = float(r["AverageNs"]) / 1e3
            rows.append((cfg, k, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"],
                         bt.get("roofline", {}).get("kernel_us", ""), bu.get("roofline", {}).get("kernel_us", "")))
    for kind in ("fetch", "write", "sq", "l2"):
        f = newest(os.path.join(d, kind, "**/*_counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                S["kernels"].setdefault(k, {}).setdefault("pmc", {})[c] = sum(v) / len(v)
    for k, e in S["kernels"].items():
        p, t_us = e.get("pmc", {}), e.get("trace_avg_us")
        der = e.setdefault("derived", {})
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced read -> doubled
            der["hbm_fetch_bytes_x2"] = p["FETCH_SIZE"] * 1024 * 2
            der["hbm_write_bytes"] = p["WRITE_SIZE"] * 1024
            der["hbm_bytes_per_launch"] = der["hbm_fetch_bytes_x2"] + der["hbm_write_bytes"]
            if t_us:
                der["hbm_GBps"] = der["hbm_bytes_per_launch"] / (t_us * 1e-6) / 1e9
                der["hbm_fraction_of_8TBps"] = der["hbm_GBps"] / 8000.0
        if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p and p["TCC_HIT_sum"] + p["TCC_MISS_sum"] > 0:
            der["l2_hit_rate"] = p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"])
        if "GRBM_GUI_ACTIVE" in p and t_us:
            # summed over the 8 XCDs: / 8 = shader cycles of the dispatch (reads high below ~0.3 ms, see the guide)
            der["shader_clock_GHz_under_profiler"] = p["GRBM_GUI_ACTIVE"] / 8.0 / (t_us * 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p and "GRBM_GUI_ACTIVE" in p and p["GRBM_GUI_ACTIVE"] > 0:
            der["mfma_busy_fraction_of_1024_SIMDs"] = p["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * p["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in p:
            der["executed_mfma_gflop"] = p["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / 1e9
            if t_us:
                der["executed_mfma_frac_of_157.3TF_under_profiler"] = der["executed_mfma_gflop"] * 1e9 / (t_us * 1e-6) / 157.3e12
    # bench.py's roofline.traffic = HBM bytes of ONE STEP of the config: every hot kernel's bytes per launch x its
    # launches per step (the block is three launches, the F(4x4) path four), with the kernels named
    if "@" not in cfg and S["kernels"]:
        ks = {k: e for k, e in S["kernels"].items() if "stamped" not in k and "hbm_bytes_per_launch" in e.get("derived", {})}
        if ks:
            # launches per step from the trace: a kernel called c times where the most-called hot kernel is called
            # c_max times runs c / c_max times per step (all hot kernels of a config run once per step today)
            cmax = max(e.get("calls", 1) for e in ks.values())
            total = sum(e["derived"]["hbm_bytes_per_launch"] * (e.get("calls", cmax) / cmax) for e in ks.values())
            dom = max(ks, key=lambda k: ks[k].get("trace_avg_us", 0) * ks[k].get("calls", 0))
            traffic[cfg] = {"kernels": sorted(ks), "dominant_kernel": dom, "hbm_bytes_per_launch": total,
                            "dominant_kernel_hbm_bytes": ks[dom]["derived"]["hbm_bytes_per_launch"],
                            "fetch_bytes_x2": sum(e["derived"]["hbm_fetch_bytes_x2"] * (e.get("calls", cmax) / cmax) for e in ks.values()),
                            "write_bytes": sum(e["derived"]["hbm_write_bytes"] * (e.get("calls", cmax) / cmax) for e in ks.values()),
                            "source": f"profiles/{tag}/summary.json"}

with open(os.path.join(dst, "kernel_stats.csv"), "w") as out:
    out.write("config,kernel,calls,avg_ns,min_ns,max_ns,bench_kernel_us_same_profiled_run,bench_kernel_us_unprofiled_run\n")
    for r in rows:
        out.write("%s,%s,%s,%.0f,%s,%s,%s,%s\n" % r)
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1, sort_keys=True)
if traffic:
    old = {}
    try:
        old = json.load(open("profiles/pmc_traffic.json"))
    except Exception:
        pass
    old.update(traffic)
    json.dump(old, open("profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
with open(os.path.join(dst, "bench_layers_unprofiled.jsonl"), "w") as out:
    for d in sorted(glob.glob(os.path.join(src, "*/"))):
        p = os.path.join(d, "bench_unprofiled.json")
        if os.path.exists(p):
            out.write(open(p).read().strip().splitlines()[-1] + "\n")
# compact table
for cfg, S in summary.items():
    for k, e in S["kernels"].items():
        d_ = e.get("derived", {})
        print("%-24s %-34s %8.1f us  hbm %7.1f MB %6.0f GB/s  L2 %4.2f  mfma_busy %4.2f  exec %6.2f GF" % (
            cfg, k, e.get("trace_avg_us", 0), d_.get("hbm_bytes_per_launch", 0) / 1e6, d_.get("hbm_GBps", 0),
            d_.get("l2_hit_rate", 0), d_.get("mfma_busy_fraction_of_1024_SIMDs", 0), d_.get("executed_mfma_gflop", 0)))
