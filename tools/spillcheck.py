"""Developer tool: where the fused kernel's spill code sits (per basic block that holds MFMAs).
usage: python tools/spillcheck.py [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-winograd_amd", "csrc")
src = "wino_f2_fused.hip"
with tempfile.TemporaryDirectory() as d:
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                          "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", os.path.join(d, "x.o"), "-save-temps",
                          "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:], capture_output=True, text=True, cwd=d)
    if out.returncode: sys.exit(out.stderr[-3000:])
    cur = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m: cur = m.group(1)
        if cur and "fused_kernel" in cur and re.search(r"VGPRs:|SGPRs Spill|VGPRs Spill|Occupancy", line):
            print(cur[-40:], line.split("remark:")[-1].strip())
    text = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f][0]).read()
for name in re.findall(r"^(_ZN4wino5fused20wino_f2_fused_kernel\w+):", text, re.M):
    i = text.index(name + ":")
    body = text[i:text.index(".Lfunc_end", i)].splitlines()
    blocks, blk = [], []
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l): blocks.append(blk); blk = []
        blk.append(l)
    blocks.append(blk)
    for b in blocks:
        if any("v_mfma" in x for x in b):
            sp = [(k, x) for k, x in enumerate(b) if any(p in x for p in ("v_readlane", "v_writelane", "scratch_"))]
            print(name[-30:], b[0].split(":")[0], "len", len(b), "mfma", sum("v_mfma" in x for x in b), "spill instrs", len(sp))
            for k, x in sp: print("    ", k, x.strip())
