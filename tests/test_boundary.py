"""The drop-in boundary to the letter: a reference-shaped caller builds against
include/ + the shared library exactly as INTEGRATION.md section 1 says, host code that names the reference's
file-name objects compiles, the packed return value survives the reference's signed decode, and the
stdout protocol carries the reference's labels on request.  CPU only; the same caller is RUN on the
GPU box by tests/test_gpu_parity.py."""
import ctypes
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "cuda-winograd_amd")

# A caller shaped like the reference's Test.c:13-56 (own text): unprototyped use of the six entry
# points through the reference-named headers, `res >> 16` / `res & 0xFFFF`, first two calls
# discarded, integer means over nTest - 2.  The one edit INTEGRATION.md section 1 prescribes is made:
# cudaSetDevice(0) -> wino_set_device(0).
REFERENCE_SHAPED_CALLER = r"""
#include <stdio.h>
#include <stdlib.h>
#include "Kernel128_one.h"
#include "Kernel128_winograd.h"
#include "Kernel256_one.h"
#include "Kernel256_winograd.h"
#include "util.h"
#include "winograd_mi355x.h"

int main(int argc, char** argv) {
  int nTest = 5, sum = 0, sum_other = 0, i, mode = 0;
  wino_set_device(0);
  if (argc >= 2) mode = atoi(argv[1]);
  if (argc >= 3) nTest = atoi(argv[2]);
  for (i = 0; i < nTest; i++) {
    int res = -1;
    printf("---- Iter: %d ----\n", i);
    switch (mode) {
      case 0: res = kernel_128(); break;
      case 1: res = kernel_256(); break;
      case 2: res = kernel_128_1_in(); break;
      case 3: res = kernel_128_1_out(); break;
      case 4: res = kernel_256_1_in(); break;
      case 5: res = kernel_256_1_out(); break;
    }
    if (i > 1) { sum += res >> 16; sum_other += res & 0xFFFF; }
  }
  printf("Average Total Time: [Mine: %d us], [cuDNN: %d us]\n", sum / (nTest - 2), sum_other / (nTest - 2));
  return 0;
}
"""


def build_reference_shaped_caller(workdir):
    """Compile + link per INTEGRATION.md section 1; returns the executable's path."""
    src = os.path.join(workdir, "RefShapedTest.c")
    with open(src, "w") as f:
        f.write(REFERENCE_SHAPED_CALLER)
    exe = os.path.join(workdir, "RefShapedTest")
    obj = os.path.join(workdir, "RefShapedTest.o")
    subprocess.check_call(["gcc", "-Wall", "-Werror", "-I" + INC, "-c", src, "-o", obj])
    subprocess.check_call(["gcc", "-o", exe, obj, "-L" + LIBDIR, "-lwinograd_mi355x",
                           "-Wl,-rpath," + LIBDIR, "-lpthread", "-lm"])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_reference_shaped_caller_builds_against_include_and_so(tmp_path):
    exe = build_reference_shaped_caller(str(tmp_path))
    needed = subprocess.check_output(["readelf", "-d", exe]).decode()
    assert "libwinograd_mi355x.so" in needed
    undefined = subprocess.check_output(["nm", "-u", exe]).decode()
    for sym in ("kernel_128", "kernel_256", "kernel_128_1_in", "kernel_128_1_out", "kernel_256_1_in",
                "kernel_256_1_out", "wino_set_device"):
        assert sym in undefined, sym


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="no gcc/g++")
def test_reference_file_name_objects_are_source_compatible(tmp_path):
    """Host code written against the reference names inputName128 & co (Kernel128_winograd.h:8-18,
    Kernel128_one.h:8-16 and the 256 headers); it must compile as C and as C++, and two translation
    units that include the headers must link (the objects have internal linkage here)."""
    names = {
        "Kernel128_winograd.h": ["inputName128", "biasName128", "weight_winograd_Name128", "weight_NCHW_Name128",
                                 "bnBiasName128", "bnScaleName128", "bnBias_winograd_Name128",
                                 "bnScale_winograd_Name128", "eMeanName128", "eVarName128"],
        "Kernel256_winograd.h": ["inputName256", "biasName256", "weight_winograd_Name256", "weight_NCHW_Name256",
                                 "bnBiasName256", "bnScaleName256", "bnBias_winograd_Name256",
                                 "bnScale_winograd_Name256", "eMeanName256", "eVarName256"],
        "Kernel128_one.h": ["inputName128one", "weightName128one", "bnBiasName128one", "bnScaleName128one",
                            "bnBias_myKernel_Name128one", "bnScale_myKernel_Name128one", "eMeanName128one",
                            "eVarName128one"],
        "Kernel256_one.h": ["inputName256one", "weightName256one", "bnBiasName256one", "bnScaleName256one",
                            "bnBias_myKernel_Name256one", "bnScale_myKernel_Name256one", "eMeanName256one",
                            "eVarName256one"],
    }
    body = "".join('#include "%s"\n' % h for h in names) + "#include <stdio.h>\n#include <string.h>\n"
    uses = "".join('  n += (int)strlen(%s);\n' % s for v in names.values() for s in v)
    (tmp_path / "a.c").write_text(body + "int count_a(void) { int n = 0;\n" + uses + "  return n; }\n")
    (tmp_path / "b.cpp").write_text(body + 'extern "C" int count_a(void);\nint main() { int n = 0;\n' + uses +
                                    '  printf("%d %d %s %s\\n", n, count_a(), inputName128, weightName256one);\n'
                                    "  return n == count_a() ? 0 : 1; }\n")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-c", str(tmp_path / "a.c"),
                           "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-c", str(tmp_path / "b.cpp"),
                           "-o", str(tmp_path / "b.o")])
    subprocess.check_call(["g++", "-o", str(tmp_path / "ab"), str(tmp_path / "a.o"), str(tmp_path / "b.o")])
    out = subprocess.check_output([str(tmp_path / "ab")]).decode().split()
    assert out[0] == out[1] and out[2] == "data/input_14_1_128.bin" and out[3] == "data/weight_one_1024.bin"


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="no gcc/g++")
def test_util_h_alone_brings_the_libc_headers_the_reference_hosts_rely_on(tmp_path):
    """The reference's util.h:9-19 includes <stdio.h> <stdlib.h> <string.h> <math.h> <inttypes.h> <assert.h>
    <errno.h> <float.h> for everyone (its util.c:1-3 includes nothing else for printf / malloc / exit): a
    reference-side translation unit that includes ONLY util.h must compile unchanged, as C and as C++."""
    body = r'''
#include "util.h"
int only_util_h(int n) {
  float* p = (float*)malloc((size_t)n * sizeof(float));
  uint64_t t = getTimeMicroseconds64();
  assert(n > 0);
  if (!p) { printf("malloc: %s\n", strerror(errno)); exit(0); }
  memset(p, 0, (size_t)n * sizeof(float));
  p[0] = (float)fabs(-1.5) + FLT_EPSILON;
  printf("%" PRIu64 " %f\n", t, p[0]);
  free(p);
  return 0;
}
'''
    (tmp_path / "u.c").write_text(body)
    (tmp_path / "u.cpp").write_text(body)
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-c", str(tmp_path / "u.c"),
                           "-o", str(tmp_path / "u_c.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-c", str(tmp_path / "u.cpp"),
                           "-o", str(tmp_path / "u_cpp.o")])


def test_file_names_are_the_ones_the_generator_writes(gen_mod, tmp_path):
    """Every name object of the four headers is a file this repo's generator (byte-identical to the
    reference's, tests/test_generator.py) actually writes."""
    import re
    gen_mod.generate_reference_set(seed=0, out_dir=str(tmp_path / "data"), with_256=True)
    have = set(os.listdir(tmp_path / "data"))
    for h in ("Kernel128_winograd.h", "Kernel256_winograd.h", "Kernel128_one.h", "Kernel256_one.h"):
        found = re.findall(r'static const char (\w+)\[\] WINO_UNUSED = "data/([^"]+)";', open(os.path.join(INC, h)).read())
        assert len(found) in (8, 10), h
        for obj, fname in found:
            assert fname in have, (h, obj, fname)


@pytest.mark.parametrize("mine,cmp_", [(0, 0), (59, 95), (0x7FFF, 0xFFFF), (0x8000, 5), (0xFFFF, 0x10000),
                                       (40_000_000, 3_000_000)])
def test_packed_return_survives_the_signed_decode(pkg, mine, cmp_):
    """Test.c:46-47 decodes with a SIGNED `res >> 16` and `res & 0xFFFF`: the custom half is clamped to
    0x7FFF (it used to reach 0xFFFF: negative averages from 32.8 ms on), the comparator half to 0xFFFF."""
    L = pkg.lib()
    L.wino_driver_pack_times.restype = ctypes.c_int
    res = L.wino_driver_pack_times(mine, cmp_)
    assert res >= 0
    assert res >> 16 == min(mine, 0x7FFF) and res & 0xFFFF == min(cmp_, 0xFFFF)


def test_stdout_compat_switch(pkg):
    L = pkg.lib()
    L.wino_driver_set_stdout_compat(1)
    assert L.wino_driver_get_stdout_compat() == 1
    L.wino_driver_set_stdout_compat(0)
    assert L.wino_driver_get_stdout_compat() == 0


def test_status_name_and_cpu_baseline_need_a_prior_call(pkg):
    L = pkg.lib()
    assert L.wino_last_status_name() == b"hipSuccess"
    r = pkg.CpuBaselineResult()
    assert L.wino_driver_cpu_baseline(ctypes.byref(r)) != 0      # no kernel_*() call yet: nothing to time
    assert L.wino_driver_cpu_baseline(None) != 0


def test_knobs_are_cached_until_reloaded(pkg, monkeypatch):
    """The WINO_* developer knobs are read once per process, not per launch: an environment change
    shows only after wino_debug_reload_knobs()."""
    L = pkg.lib()

    def grid():
        g, r, it = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        t = ctypes.c_long()
        assert L.wino_conv3x3_plan(128, 14, 14, 256, 256, 256, ctypes.byref(g), ctypes.byref(r), ctypes.byref(t),
                                   ctypes.byref(it)) == 0
        return g.value

    monkeypatch.delenv("WINO_SK_GRID", raising=False)
    L.wino_debug_reload_knobs()
    assert grid() == 256
    monkeypatch.setenv("WINO_SK_GRID", "100")
    assert grid() == 256            # cached
    L.wino_debug_reload_knobs()
    assert grid() == 100
    monkeypatch.delenv("WINO_SK_GRID")
    L.wino_debug_reload_knobs()
    assert grid() == 256
