"""The repo's own data generator must write the reference generator's bytes.

tests/golden/reference_files.json holds size + SHA-256 of every file the REFERENCE
data_generator.py wrote (imported in the build container by tests/golden/make_golden.py)
for seeds 0 and 1, in the reference's __main__ call order followed by the C=K=256 set."""
import hashlib
import os

import numpy as np
import pytest


def _sha(p):
    with open(p, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


@pytest.mark.parametrize("seed", [0, 1])
def test_files_match_reference_hashes(seed, tmp_path, gen_mod, golden_files):
    out = tmp_path / "data"
    gen_mod.generate_reference_set(seed=seed, out_dir=str(out), with_256=True)
    want = golden_files[f"seed{seed}"]
    for name, meta in want.items():
        p = out / name
        assert p.exists(), name
        assert os.path.getsize(p) == meta["bytes"], name
        assert _sha(p) == meta["sha256"], f"{name} differs from the reference generator's bytes"
    # every reference file is covered; extra files are only this repo's additions
    extra = set(os.listdir(out)) - set(want)
    assert all(n.startswith("weight_winograd_f2_") for n in extra), extra


def test_sizes_match_driver_reads(data_dir):
    """Sizes the host drivers read (Kernel128_winograd.cu:216-252, Kernel128_one.cu:58-64)."""
    sz = lambda n: os.path.getsize(os.path.join(data_dir, "data", n)) // 4
    for C in (128, 256):
        assert sz(f"input_14_1_{C}.bin") == 16 * 16 * C
        assert sz(f"weight_winograd_{C}_{C}.bin") == 36 * C * C
        assert sz(f"weight_NCHW_{C}_{C}.bin") == 9 * C * C
        assert sz(f"bnBias_winograd_{C}.bin") == C
    assert sz("input_one_14_1024.bin") == 196 * 1024
    assert sz("weight_one_1024.bin") == 256 * 1024


def test_batched_inputs_extend_the_single_image(tmp_path, gen_mod):
    g1 = gen_mod.Generator(seed=3, out_dir=str(tmp_path / "a"))
    g2 = gen_mod.Generator(seed=3, out_dir=str(tmp_path / "b"))
    one = g1.input_generator(128)
    four = g2.input_generator(128, N=4)
    assert four.shape == (4, 16, 16, 128)
    np.testing.assert_array_equal(one[0], four[0])  # same stream prefix
    assert np.abs(four).max() <= 0.5 and four.std() > 0.25


def test_f2_weights_are_G_g_Gt(tmp_path, gen_mod, O):
    g = gen_mod.Generator(seed=5, out_dir=str(tmp_path))
    w = g.weight_generator(16, 64)
    U = O.f2_filter_transform(w["weight_NCHW"]).astype(np.float32)
    np.testing.assert_allclose(w["weight_winograd_f2"], U, rtol=0, atol=1e-7)
    U4 = np.einsum("xr,kcrs,ys->xyck", O.G_F4, w["weight_NCHW"].astype(np.float64), O.G_F4)
    np.testing.assert_allclose(w["weight_winograd"], U4.reshape(36, 16, 64), rtol=0, atol=1e-7)


def test_fold_bn_matches_files(data_dir, gen_mod):
    from conftest import load_bin
    s, b = gen_mod.fold_bn(load_bin(data_dir, "bnScale_128.bin"), load_bin(data_dir, "bnBias_128.bin"),
                           load_bin(data_dir, "eMean_128.bin"), load_bin(data_dir, "eVar_128.bin"))
    np.testing.assert_array_equal(s, load_bin(data_dir, "bnScale_winograd_128.bin"))
    np.testing.assert_array_equal(b, load_bin(data_dir, "bnBias_winograd_128.bin"))
