"""Input-contract generator for the Winograd / 1x1 conv(+BN+ReLU) drivers.

Host-side mirror of the reference's offline data generator (reference:
data_generator.py:20-127).  It writes the same raw little-endian fp32 ``.bin``
files, with the same names, element order, value distributions and -- for a
given ``seed`` and the reference's call order -- the same bytes (pinned by
SHA-256 in tests/golden/reference_files.json, see tests/test_generator.py).

What is new relative to the reference:
  * everything is seedable and uses a private ``RandomState`` (the reference
    draws from numpy's global legacy RNG, data_generator.py:13);
  * ``input_generator`` / ``onebyone_generator`` take a batch size ``N`` and
    write ``[N][16][16][C]`` / ``[N*196][C]`` tensors (N=1 is byte-identical
    to the reference);
  * ``weight_generator`` also writes the F(2x2,3x3) Winograd-domain weights
    this framework's HIP path can consume (``weight_winograd_f2_C_K.bin``).

Run as a script it reproduces the reference's ``__main__`` (C=K=128 set plus
the 1x1 set) and additionally the C=K=256 set that ``./Test 1`` needs, which
the reference obtains by hand-editing the calls (README.md:17).
"""
from __future__ import annotations

import argparse
import os

import numpy as np

EPS = 1e-5  # reference: data_generator.py:41,106

# F(4x4,3x3) filter transform, reference: data_generator.py:65
G_F4 = np.array(
    [
        [0.25, 0, 0],
        [-1.0 / 6, -1.0 / 6, -1.0 / 6],
        [-1.0 / 6, 1.0 / 6, -1.0 / 6],
        [1.0 / 24, 1.0 / 12, 1.0 / 6],
        [1.0 / 24, -1.0 / 12, 1.0 / 6],
        [0, 0, 1],
    ]
)
# F(2x2,3x3) filter transform (Lavin & Gray); not in the reference.
G_F2 = np.array([[1.0, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1.0]])


class Generator:
    """Stateful generator: one RandomState, one output directory."""

    def __init__(self, seed: int | None = 0, out_dir: str = "data"):
        self.rng = np.random.RandomState(seed)
        self.out_dir = out_dir
        os.makedirs(out_dir, exist_ok=True)

    # -- helpers ---------------------------------------------------------
    def _rand(self, n: int) -> np.ndarray:
        return self.rng.rand(n)

    def _write(self, name: str, arr: np.ndarray) -> np.ndarray:
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        with open(os.path.join(self.out_dir, name), "wb") as f:
            f.write(arr.tobytes())
        return arr

    # -- reference: data_generator.py:20-47 -------------------------------
    def bias_generator(self, output_channel: int = 128) -> dict:
        K = output_channel
        out = {}
        out["bias"] = self._write(f"bias_{K}.bin", (self._rand(K) - 0.5))
        bnScale = out["bnScale"] = self._write(f"bnScale_{K}.bin", (self._rand(K) - 0.5))
        bnBias = out["bnBias"] = self._write(f"bnBias_{K}.bin", (self._rand(K) - 0.5))
        eMean = out["eMean"] = self._write(f"eMean_{K}.bin", (self._rand(K) - 0.5))
        eVar = out["eVar"] = self._write(f"eVar_{K}.bin", (self._rand(K) * 3 + 5))
        # folded in float32, exactly as the reference does (float32 arrays, python-float eps)
        scale_w = bnScale / np.sqrt(eVar + EPS)
        bias_w = bnBias - bnScale * eMean / np.sqrt(eVar + EPS)
        out["bnScale_winograd"] = self._write(f"bnScale_winograd_{K}.bin", scale_w)
        out["bnBias_winograd"] = self._write(f"bnBias_winograd_{K}.bin", bias_w)
        return out

    # -- reference: data_generator.py:49-53 (N is new) ---------------------
    def input_generator(self, input_channel: int = 128, feature_map_size: int = 14,
                        padding: int = 1, N: int = 1) -> np.ndarray:
        hw = feature_map_size + 2 * padding
        n = N * hw * hw * input_channel
        a = self._rand(n) - 0.5
        name = f"input_{feature_map_size}_{padding}_{input_channel}.bin"
        if N != 1:
            name = f"input_{feature_map_size}_{padding}_{input_channel}_N{N}.bin"
        return self._write(name, a).reshape(N, hw, hw, input_channel)

    # -- reference: data_generator.py:55-78 --------------------------------
    def weight_generator(self, input_channel: int = 128, output_channel: int = 128,
                         write_f2: bool = True) -> dict:
        C, K = input_channel, output_channel
        w = self._write(f"weight_NCHW_{C}_{K}.bin", self._rand(C * K * 9) - 0.5)
        g = w.reshape(K * C, 3, 3)  # index k*C + c, reference :64,70
        out = {"weight_NCHW": w.reshape(K, C, 3, 3)}
        out["weight_winograd"] = self._write(
            f"weight_winograd_{C}_{K}.bin", winograd_domain_weights(g, C, K, G_F4))
        if write_f2:
            out["weight_winograd_f2"] = self._write(
                f"weight_winograd_f2_{C}_{K}.bin", winograd_domain_weights(g, C, K, G_F2))
        return out

    # -- reference: data_generator.py:80-113 (N is new) ---------------------
    def onebyone_generator(self, input_channel: int = 256, output_channel: int = 1024,
                           feature_map_size: int = 14, N: int = 1) -> dict:
        Cin, Cout, fm = input_channel, output_channel, feature_map_size
        out = {}
        name = f"input_one_{fm}_{Cout}.bin" if N == 1 else f"input_one_{fm}_{Cout}_N{N}.bin"
        out["input"] = self._write(name, (self._rand(N * fm * fm * Cout) - 0.5) * 40)
        out["weight"] = self._write(f"weight_one_{Cout}.bin", (self._rand(Cin * Cout) - 0.5) * 40)
        bnScale = out["bnScale"] = self._write(f"bnScale_one_{Cout}.bin", (self._rand(Cout) - 0.5) * 40)
        bnBias = out["bnBias"] = self._write(f"bnBias_one_{Cout}.bin", (self._rand(Cout) - 0.5) * 40)
        eMean = out["eMean"] = self._write(f"eMean_one_{Cout}.bin", (self._rand(Cout) - 0.5) * 40)
        eVar = out["eVar"] = self._write(f"eVar_one_{Cout}.bin", self._rand(Cout) * 20 + 5)
        scale_f = bnScale / np.sqrt(eVar + EPS)
        bias_f = bnBias - bnScale * eMean / np.sqrt(eVar + EPS)
        out["bnScale_myKernel"] = self._write(f"bnScale_myKernel_one_{Cout}.bin", scale_f)
        out["bnBias_myKernel"] = self._write(f"bnBias_myKernel_one_{Cout}.bin", bias_f)
        return out


def winograd_domain_weights(g: np.ndarray, C: int, K: int, G: np.ndarray) -> np.ndarray:
    """U[(x*E+y)][c][k] = (G g_{k,c} G^T)[x][y], computed in float64 per (k, c)
    pair with two ``np.dot`` calls like the reference (data_generator.py:68-77),
    so that the float32 cast lands on the same bits."""
    E = G.shape[0]
    out = np.empty((E * E, C, K), dtype=np.float64)
    Gt = G.transpose()
    for k in range(K):
        for c in range(C):
            b = np.dot(G, g[k * C + c])
            b = np.dot(b, Gt)
            out[:, c, k] = b.reshape(E * E)
    return out.astype(np.float32)


def fold_bn(bnScale, bnBias, eMean, eVar, eps: float = EPS):
    """Folded BN in float32, as data_generator.py:41-46 / :106-112."""
    bnScale = np.asarray(bnScale, np.float32)
    bnBias = np.asarray(bnBias, np.float32)
    eMean = np.asarray(eMean, np.float32)
    eVar = np.asarray(eVar, np.float32)
    s = bnScale / np.sqrt(eVar + eps)
    b = bnBias - bnScale * eMean / np.sqrt(eVar + eps)
    return s.astype(np.float32), b.astype(np.float32)


def generate_reference_set(seed: int = 0, out_dir: str = "data", with_256: bool = True) -> None:
    """The reference's __main__ call order (data_generator.py:116-127); the
    C=K=256 set is drawn afterwards from the same stream so the 128 and 1x1
    files stay byte-identical to the reference for equal seed."""
    g = Generator(seed, out_dir)
    g.bias_generator(128)
    g.input_generator(128)
    g.weight_generator(128, 128)
    g.onebyone_generator()
    if with_256:
        g.bias_generator(256)
        g.input_generator(256)
        g.weight_generator(256, 256)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="data")
    ap.add_argument("--no-256", action="store_true")
    a = ap.parse_args()
    generate_reference_set(a.seed, a.out, not a.no_256)
    print("data written to", a.out)
