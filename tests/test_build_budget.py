"""Register budgets of the hot kernels, checked at build time (no GPU needed: hipcc cross-compiles).

Both kernels are tuned to a resident-wave count -- the 1x1 GEMM to what its LDS footprint allows
(two 8-wave workgroups per CU: 4 waves per SIMD, at most 128 VGPRs; three 4-wave workgroups: 3
waves per SIMD, at most 168), the fused 3x3 throughput kernel to 2 (at most 256) -- and a few extra live registers in an epilogue silently halve that (it happened three times
while these kernels were written; the only symptom is a slower launch).  hipcc reports the
allocation per kernel with -Rpass-analysis=kernel-resource-usage; the emitted ISA says where
spill code sits.

What is asserted: the VGPR budget and occupancy; no VGPR spill (scratch memory) in the fused kernel
and at most a handful, outside the loops, in the GEMM; and no spill code of either kind -- v_readlane / v_writelane for SGPRs parked in VGPR lanes, scratch_load /
scratch_store -- inside any basic block that holds MFMAs, i.e. the main loops, where every extra
instruction is paid for (in-order issue).  SGPR spills outside the loops
(prologue, epilogue, stream-K bookkeeping: 36-57 in the fused kernel, up to 37 in the GEMM's stream-K variants) cost a
lane move each and are only bounded."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "cuda-winograd_amd", "csrc")


def _compile_report(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(CSRC, src),
                          "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage", "-save-temps"],
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                         ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    isa = [f for f in os.listdir(tmp_path) if f.endswith(".s") and "gfx950" in f]
    assert len(isa) == 1, isa
    text = (tmp_path / isa[0]).read_text()
    for name, v in kernels.items():
        i = text.index(name + ":")
        body = text[i:text.index(".Lfunc_end", i)].splitlines()
        blocks, blk = [], []
        for line in body:
            if re.match(r"^\.LBB\d+_\d+:", line):
                blocks.append(blk)
                blk = []
            blk.append(line)
        blocks.append(blk)
        hot = [b for b in blocks if any("v_mfma" in x for x in b)]
        v["mfma"] = sum("v_mfma" in x for b in hot for x in b)
        v["spill_code_in_mfma_blocks"] = sum(any(p in x for p in ("v_readlane", "v_writelane", "scratch_load", "scratch_store"))
                                             for b in hot for x in b)
    return kernels


def test_gemm_kernel_keeps_four_waves_per_simd(tmp_path):
    k = {n: v for n, v in _compile_report("conv1x1.hip", tmp_path).items() if "conv1x1_bn_kernel" in n}
    assert len(k) == 8, sorted(k)          # {4, 8 waves} x {plain, stream-K} x {no residual, residual}
    for name, v in k.items():
        # 8-wave form: 60 KB of LDS -> two workgroups per CU -> 4 waves per SIMD -> 128 VGPRs.
        # 4-wave form: 44 KB -> three workgroups (a fourth does not fit) -> 3 waves per SIMD -> 168.
        eight = "ILi32ELi8E" in name
        assert eight or "ILi32ELi4E" in name, name
        budget, waves = (128, 4) if eight else (168, 3)
        # Held to its residency, the 8-wave stream-K variant sends 4 VGPRs to scratch in the
        # once-per-segment address setup and the gather path (13-17 SGPRs go to lanes in the
        # stream-K variants): a few accesses per tile, bounded here; none of it may sit in a
        # block with MFMAs.
        assert v["vgprs"] <= budget and v["occupancy"] >= waves and v["spill"] <= 8, (name, v)
        assert v["mfma"] >= 56 and v["spill_code_in_mfma_blocks"] == 0 and v["sgpr_spill"] <= 40, (name, v)


def test_fused_kernel_keeps_two_waves_per_simd(tmp_path):
    k = {n: v for n, v in _compile_report("wino_f2_fused.hip", tmp_path).items() if "wino_f2_fused_kernel" in n}
    # the 14x14 specialisation, the general H x W form, and the stamped diagnostic build of the
    # first (ILi16E: wino_diag_conv3x3_clock), which must stay within the same budget to be a
    # faithful probe of the product kernel's clock
    # (x {with a stream-K tail, whole items only} for the two product forms)
    assert len(k) == 5 and sum("ILi16E" in n for n in k) == 1, sorted(k)
    for name, v in k.items():
        assert v["vgprs"] <= 256 and v["occupancy"] >= 2 and v["spill"] == 0, (name, v)
        assert v["mfma"] == 128 and v["spill_code_in_mfma_blocks"] == 0 and v["sgpr_spill"] <= 80, (name, v)


def test_latency_kernels_spill_nothing(tmp_path):
    """The latency kernels run one wave per SIMD and may use the whole 512-register file -- the 16 x 64 form of the 3x3
    kernel (128 accumulators + 128 filter-fragment registers + the staging buffer) does -- but a spill there is a
    round trip to memory inside a loop that is bound by memory round trips: the first build of that form spilled 55
    registers (DESIGN_NOTEBOOK section 3).  None may spill, and no spill code may sit beside MFMAs."""
    k3 = {n: v for n, v in _compile_report("wino_f2_fused.hip", tmp_path).items() if "wino_f2_small_kernel" in n}
    assert len(k3) == 6, sorted(k3)                # block widths 16 / 32 / 64 x {14x14, any feature map}
    for name, v in k3.items():
        assert v["spill"] == 0 and v["sgpr_spill"] == 0 and v["spill_code_in_mfma_blocks"] == 0, (name, v)
        assert v["mfma"] >= 32, (name, v)
    sub = tmp_path / "one"
    sub.mkdir()
    k1 = {n: v for n, v in _compile_report("conv1x1.hip", sub).items() if "conv1x1_small_kernel" in n}
    assert len(k1) == 18, sorted(k1)               # KS {1, 2, 4} x RT {1, 2} x CT {1, 2, 4}
    for name, v in k1.items():
        assert v["spill"] == 0 and v["sgpr_spill"] == 0 and v["spill_code_in_mfma_blocks"] == 0, (name, v)
