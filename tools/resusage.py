#!/usr/bin/env python3
"""Developer tool: hipcc's per-kernel resource usage of a csrc/*.hip translation unit, one line per kernel.
    python tools/resusage.py wino_f2_fused.hip [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-winograd_amd", "csrc")
src = sys.argv[1]
path = src if os.path.exists(src) else os.path.join(CSRC, src)
with tempfile.TemporaryDirectory() as d:
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                          "-I" + CSRC, "-c", path, "-o", os.path.join(d, "x.o"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:],
                         capture_output=True, text=True, cwd=d)
if out.returncode:
    sys.exit(out.stderr[-3000:])
cur = None
rows = []
for line in out.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                     ("sspill", r"SGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur.setdefault(key, int(m.group(1)))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    print(f"{name[:90]:90s} vgpr {r.get('vgpr')} agpr {r.get('agpr')} sgpr {r.get('sgpr')} spill v{r.get('vspill')}/s{r.get('sspill')} "
          f"scratch {r.get('scratch')} occ {r.get('occ')} lds {r.get('lds')}")
