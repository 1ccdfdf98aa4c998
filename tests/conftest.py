"""Shared fixtures.  `-m "not gpu"` runs everything that needs no GPU (oracle vs golden
vectors, generator vs reference hashes, host logic, ABI surface, gloo sharding);
`-m gpu` runs the parity tests proper, through the C-ABI, on a real MI355X."""
import ctypes
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


class _Knobs:
    """The library reads its WINO_* developer knobs once per process; tests that sweep them change the
    environment through this helper, which makes the library re-read it (wino_debug_reload_knobs)."""

    def __init__(self, monkeypatch, lib):
        self._mp, self._lib = monkeypatch, lib

    def set(self, name, value):
        self._mp.setenv(name, str(value))
        self._lib.wino_debug_reload_knobs()

    def unset(self, name):
        self._mp.delenv(name, raising=False)
        self._lib.wino_debug_reload_knobs()


@pytest.fixture
def knobs(pkg, monkeypatch):
    k = _Knobs(monkeypatch, pkg.lib())
    yield k
    monkeypatch.undo()
    pkg.lib().wino_debug_reload_knobs()


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="session")
def gen_mod(pkg):
    import importlib
    return importlib.import_module("cuda_winograd_amd.data_generator")


@pytest.fixture(scope="session")
def golden_outputs():
    return dict(np.load(os.path.join(GOLDEN, "outputs_seed0.npz")))


@pytest.fixture(scope="session")
def golden_files():
    with open(os.path.join(GOLDEN, "reference_files.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def data_dir(tmp_path_factory, gen_mod):
    """The reference's data set (seed 0: 128 set, 1x1 set, 256 set) written by this repo's
    own generator into <tmp>/data; byte-identity with the reference generator is what
    tests/test_generator.py pins."""
    base = tmp_path_factory.mktemp("wino")
    gen_mod.generate_reference_set(seed=0, out_dir=str(base / "data"), with_256=True)
    return str(base)


def load_bin(base, name, n=None):
    a = np.fromfile(os.path.join(base, "data", name), dtype="<f4")
    return a if n is None else a[:n]


@pytest.fixture(scope="session")
def c_oracle():
    """ctypes handle of oracle/liboracle.so (built by `make oracle`)."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "oracle"], cwd=ROOT)
    L = ctypes.CDLL(path)
    fp = ctypes.c_void_p
    L.oracle_conv3x3_im2col.argtypes = [fp, fp, fp, fp, fp] + [ctypes.c_int] * 5
    L.oracle_conv1x1.argtypes = [fp, fp, fp, fp, fp, ctypes.c_long] + [ctypes.c_int] * 4
    L.oracle_winograd_f4.argtypes = [fp, fp, fp, fp, fp] + [ctypes.c_int] * 3
    L.oracle_now_us.restype = ctypes.c_double
    return L


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


requires_gpu = pytest.mark.gpu
