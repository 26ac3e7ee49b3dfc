"""CPU checks of the 3-D line oracle (oracle/line3d_oracle.cpp): its rand() is glibc's, and basic geometry of what Frame::isLineGood returns."""
import ctypes

import numpy as np

import oracle_lib as O
from planarslam_amd import synth


def test_glibc_rand_emulation_matches_libc():
    libc = ctypes.CDLL("libc.so.6")
    for seed in (0, 1, 2, 12345, 2 ** 31 + 5, 2 ** 32 - 1):
        libc.srand(ctypes.c_uint(seed))
        want = [libc.rand() for _ in range(40)]
        assert list(O.glibc_rand(seed, 40)) == want


def test_is_line_good_on_a_planar_scene():
    g = synth.gray_image(11); d = synth.depth_image(31)
    kl = O.extract_line_segment(g, tie_order=0)[0]
    r = O.is_line_good(kl, d, seed=7)
    good = r["good"] > 0
    assert good.sum() >= 20
    A, B = r["lines3d"][good, :3], r["lines3d"][good, 3:]
    dirs = (A - B) / np.linalg.norm(A - B, axis=1, keepdims=True)
    np.testing.assert_allclose(dirs, r["direction"][good], atol=1e-12)
    assert (np.linalg.norm(A - B, axis=1) > 0.02).all() and (r["depth_line"][good] >= 0).all() and (r["depth_line"][~good] == -1).all()
    assert (r["n_inliers"][good] <= r["n_samples"][good]).all() and (r["n_samples"] <= 51).all()
    d2 = d.copy(); d2[:] = 0
    assert O.is_line_good(kl, d2, seed=7)["good"].sum() == 0


import os  # noqa: E402

import pytest  # noqa: E402

import line3d_cases as cases  # noqa: E402


@pytest.mark.parametrize("name", list(cases.CASES))
def test_line3d_oracle_equals_real_reference_fixture(golden_dir, name):
    """tests/golden/line3d_ref.npz = the REAL src/Frame.cc:189-267 + src/LineExtractor.cpp code (oracle/_ref/ref_line3d): every output, every bit."""
    g = np.load(os.path.join(golden_dir, "line3d_ref.npz"))
    kl, d, seed = cases.build(name)
    r = O.is_line_good(kl, d, seed)
    for k in ("depth_line", "lines3d", "good", "direction"):
        np.testing.assert_array_equal(r[k], g[f"{name}/{k}"], err_msg=k)
    good = r["good"] > 0
    np.testing.assert_array_equal(r["n_inliers"][good], g[f"{name}/n_inliers"][good])


@pytest.mark.skipif(not os.path.exists(O.ref_line3d_path()), reason="oracle/_ref/ref_line3d not built (reference tree absent)")
def test_line3d_oracle_equals_real_reference_live():
    kl, d, _ = cases.build("rough_depth")
    for seed in (1, 2, 3):
        ref, got = O.run_ref_line3d(kl, d, seed), O.is_line_good(kl, d, seed)
        for k in ("depth_line", "lines3d", "good", "direction"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{k} seed {seed}")
