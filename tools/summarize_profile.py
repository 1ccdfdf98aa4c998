#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into
profiles/<tag>/: per-kernel stats from the kernel traces, PMC means per launch, and
profiles/pmc_traffic.json (HBM bytes per launch, FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950)."""
import csv, glob, json, os, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = f"gpurun_out/prof_{tag}"
dst = f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
HOT = ("wino_f2_fused_kernel", "conv1x1_bn_kernel")

def find(pattern):
    """newest file per profiling directory (gpurun merges successive runs into the same tree)"""
    best = {}
    for f in glob.glob(os.path.join(src, pattern), recursive=True):
        d = f.split(src + "/")[1].split("/")[0]
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())

summary = {}
with open(os.path.join(dst, "kernel_stats.csv"), "w") as out:
    out.write("layer,kernel,calls,avg_ns,min_ns,max_ns,bench_kernel_us_same_run\n")
    for f in find("trace_*/**/*_kernel_stats.csv"):
        layer = f.split("trace_")[1].split("/")[0]
        bench = {}
        try:
            bench = json.loads(open(os.path.join(src, f"bench_trace_{layer}.json")).read().strip().splitlines()[-1])
        except Exception:
            pass
        for r in csv.DictReader(open(f)):
            if any(h in r["Name"] for h in HOT):
                name = next(h for h in HOT if h in r["Name"])
                out.write(f'{layer},{name},{r["Calls"]},{float(r["AverageNs"]):.0f},{r["MinNs"]},{r["MaxNs"]},'
                          f'{bench.get("roofline", {}).get("kernel_us", "")}\n')
                summary.setdefault(layer, {})["trace_avg_us"] = float(r["AverageNs"]) / 1e3

traffic = {}
for f in find("pmc_*/**/*_counter_collection.csv"):
    d = f.split(src + "/")[1].split("/")[0]            # pmc_fetch_conv3x3_256
    _, kind, layer = d.split("_", 2)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(h in r["Kernel_Name"] for h in HOT):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summary.setdefault(layer, {}).setdefault("pmc", {})[k] = sum(v) / len(v)

for layer, s in summary.items():
    p = s.get("pmc", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        # rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reads 1/2 of a
        # wide coalesced streaming read -> doubled (guide: MI355X_MICROARCH.md, HBM section)
        fetch = p["FETCH_SIZE"] * 1024 * 2
        write = p["WRITE_SIZE"] * 1024
        s["hbm_bytes_per_launch"] = fetch + write
        s["fetch_bytes_corrected"] = fetch
        s["write_bytes"] = write
        traffic[layer] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_x2": fetch, "write_bytes": write,
                          "source": f"profiles/{tag}/summary.json"}
# derived figures (per launch): matrix-pipe utilisation and HBM rate against the gfx950 peaks
for layer, s in summary.items():
    p = s.get("pmc", {})
    t_us = s.get("trace_avg_us")
    if not t_us:
        continue
    der = s.setdefault("derived", {})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in p and "GRBM_GUI_ACTIVE" in p:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = shader cycles of the dispatch
        cycles = p["GRBM_GUI_ACTIVE"] / 8.0
        der["shader_clock_GHz_under_profiler"] = cycles / (t_us * 1e3)
        der["mfma_busy_fraction_of_1024_SIMDs"] = p["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles)
    if "hbm_bytes_per_launch" in s:
        der["hbm_GBps"] = s["hbm_bytes_per_launch"] / (t_us * 1e-6) / 1e9
        der["hbm_fraction_of_8TBps"] = der["hbm_GBps"] / 8000.0
    if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p:
        der["l2_hit_rate"] = p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"])
json.dump(summary, open(os.path.join(dst, "summary.json"), "w"), indent=1, sort_keys=True)
if traffic:
    json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
for f in ("bench_unprofiled.json",):
    p = os.path.join(src, f)
    if os.path.exists(p):
        open(os.path.join(dst, f), "w").write(open(p).read())
print(json.dumps(summary, indent=1, sort_keys=True))
