#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE generator (build container only).

Run once, in the container that has /root/reference mounted:

    python tests/golden/make_golden.py

It imports the reference's ``data_generator`` module *as is* (nothing is copied),
seeds numpy's global legacy RNG (the reference draws from it,
data_generator.py:13), calls the reference functions in the reference's
``__main__`` order (data_generator.py:116-127) followed by the C=K=256 set, and
records

  reference_files.json   name -> {bytes, sha256} for seeds 0 and 1
  outputs_seed0.npz      expected outputs of the six ./Test layers computed
                         from those reference-written files with the fp64
                         direct convolution / fp64 GEMM in oracle/oracle.py

The fixtures are data (hashes and float arrays); no reference source text is
stored.  Nothing on the GPU box reads /root/reference.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def load(d, name, n):
    a = np.fromfile(os.path.join(d, "data", name), dtype="<f4")
    assert a.size >= n, (name, a.size, n)
    return a[:n]


def run_reference_generator(seed, workdir):
    os.makedirs(os.path.join(workdir, "data"), exist_ok=True)
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        os.environ.setdefault("MPLBACKEND", "Agg")
        sys.path.insert(0, REF)
        import data_generator as dg  # the reference module, imported in place
        np.random.seed(seed)
        dg.bias_generator(output_channel=128)
        dg.input_generator(input_channel=128)
        dg.weight_generator(128, 128)
        dg.onebyone_generator()
        dg.bias_generator(output_channel=256)
        dg.input_generator(input_channel=256)
        dg.weight_generator(256, 256)
    finally:
        os.chdir(cwd)
        sys.path.remove(REF)


def main():
    from oracle import oracle as O

    files = {}
    outputs = {}
    for seed in (0, 1):
        with tempfile.TemporaryDirectory() as d:
            run_reference_generator(seed, d)
            ent = {}
            for name in sorted(os.listdir(os.path.join(d, "data"))):
                p = os.path.join(d, "data", name)
                ent[name] = {"bytes": os.path.getsize(p), "sha256": sha(p)}
            files[f"seed{seed}"] = ent
            if seed != 0:
                continue
            # ---- expected outputs from the reference-written files (seed 0) ----
            for C in (128, 256):
                inp = load(d, f"input_14_1_{C}.bin", 16 * 16 * C).reshape(1, 16, 16, C)
                w = load(d, f"weight_NCHW_{C}_{C}.bin", 9 * C * C).reshape(C, C, 3, 3)
                s = load(d, f"bnScale_winograd_{C}.bin", C)
                b = load(d, f"bnBias_winograd_{C}.bin", C)
                y = O.conv3x3_bn_relu_direct(inp, w, s, b)[0, 1:15, 1:15, :]
                outputs[f"kernel_{C}"] = y.astype(np.float32)
                # the reference's own pre-transformed weights must equal G g G^T
                U = load(d, f"weight_winograd_{C}_{C}.bin", 36 * C * C).reshape(36, C, C)
                Uo = np.einsum("xr,kcrs,ys->xyck", O.G_F4, w.astype(np.float64), O.G_F4).reshape(36, C, C)
                assert np.abs(U - Uo).max() < 1e-6
                # and the stage-by-stage restatement of the reference kernels must agree
                yr = O.winograd_f4_reference(inp, U, s, b)[0, 1:15, 1:15, :]
                err = np.abs(yr - y).max()
                print(f"C={C}: F4 restatement vs fp64 direct: max abs err {err:.3e}")
                assert err < 1e-4
            for name, (Cin, Kout, relu) in O.ONE_BY_ONE_LAYERS.items():
                A = load(d, "input_one_14_1024.bin", 196 * Cin).reshape(196, Cin)
                B = load(d, "weight_one_1024.bin", Cin * Kout).reshape(Cin, Kout)
                s = load(d, "bnScale_myKernel_one_1024.bin", Kout)
                b = load(d, "bnBias_myKernel_one_1024.bin", Kout)
                outputs[name] = O.conv1x1_bn(A, B, b, s, relu).astype(np.float32)
    with open(os.path.join(HERE, "reference_files.json"), "w") as f:
        json.dump(files, f, indent=1, sort_keys=True)
    np.savez(os.path.join(HERE, "outputs_seed0.npz"), **outputs)
    for k, v in outputs.items():
        print(k, v.shape, float(np.abs(v).max()))


if __name__ == "__main__":
    main()
