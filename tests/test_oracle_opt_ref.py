"""Pins the pose / BA oracles (oracle/pose_oracle.cpp, ba_oracle.cpp) to the REAL reference optimiser.

oracle/_ref/ref_opt is the reference's own src/Optimizer.cc (PoseOptimization, TranslationOptimization, LocalBundleAdjustment), src/Converter.cc,
Thirdparty/g2o (core, types, solvers), g2oAddition/*.h and include/EdgeLine.h, compiled where they lie against a mini-Eigen stand-in
(oracle/shim/minieigen) and data-holder Frame / KeyFrame / MapPoint classes (oracle/shim/opt_standins.hpp).  Its outputs on the seeded
problems of tests/opt_cases.py are committed as tests/golden/opt_ref.npz (tools/gen_golden_opt.py).  Here:
  * the oracle restatements are compared with those fixtures (always), and with ref_opt run live (when the binary is present);
  * edge level: computeError / chi2 / linearizeOplus of all 12 pose-only edge classes, incl. the real base_unary_edge.hpp numeric path: bit-exact.
Tolerances: 1e-5 on the SE3 pose (BASELINE.json north_star; observed <= 5e-8), flags and inlier counts identical.  What stays unpinned is
Eigen itself (the stand-in restates Quaternion / AngleAxis / LDLT kernels from the published algorithms)."""
import os

import numpy as np
import pytest

import opt_cases as cases
import oracle_lib as ol
from planarslam_amd.synth import TUM3

POSE_TOL = 1e-5
HAVE_REF = os.path.exists(ol.ref_opt_path())


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "opt_ref.npz"))


def _unpack(bits, shape):
    return np.unpackbits(bits)[:int(np.prod(shape))].reshape(shape)


def _check_pose(want, got, b):
    assert np.abs(got["Tcw"] - want["Tcw"]).max() <= POSE_TOL
    assert np.array_equal(got["n_inliers"], want["n_inliers"])
    for k in ("pt_outlier", "ln_outlier", "pl_outlier"):
        assert np.array_equal(got[k], want[k]), k


def golden_pose(golden, name, mode, b):
    w = dict(Tcw=golden[f"pose/{name}/{mode}/Tcw"], n_inliers=golden[f"pose/{name}/{mode}/n_inliers"])
    for k in ("pt_outlier", "ln_outlier", "pl_outlier"):
        w[k] = _unpack(golden[f"pose/{name}/{mode}/{k}"], {"pt_outlier": b["pt_valid"].shape, "ln_outlier": b["ln_valid"].shape, "pl_outlier": b["pl_valid"].shape}[k])
    return w


@pytest.mark.parametrize("name", [n for n in cases.POSE_CASES if n != "c4_b256"])
def test_pose_oracle_equals_reference_fixture(golden, name):
    build, modes = cases.POSE_CASES[name]
    b = build()
    for mode in modes:
        _check_pose(golden_pose(golden, name, mode, b), ol.pose_optimize(b, TUM3, mode), b)


def test_pose_oracle_equals_reference_fixture_config4_batch256(golden):
    build, modes = cases.POSE_CASES["c4_b256"]
    b = build()
    for mode in modes:
        _check_pose(golden_pose(golden, "c4_b256", mode, b), ol.pose_optimize(b, TUM3, mode), b)


def golden_ba(golden, name, pr):
    return dict(kf_Tcw=golden[f"ba/{name}/kf_Tcw"], lm=golden[f"ba/{name}/lm"], e_outlier=_unpack(golden[f"ba/{name}/e_outlier"], pr["e_kf"].shape))


def check_ba(want, got, pr):
    assert np.abs(got["kf_Tcw"] - want["kf_Tcw"]).max() <= POSE_TOL
    pl = pr["lm_type"] == 1
    # map points / line end points come back through float32 (MapPoint::SetWorldPos, Converter::toCvMat): 1e-4 m on badly observed depths
    assert np.abs(got["lm"][~pl, :3] - want["lm"][~pl, :3]).max() <= 1e-4
    if pl.any():
        assert np.abs(got["lm"][pl] - want["lm"][pl]).max() <= POSE_TOL
    assert np.array_equal(got["e_outlier"], want["e_outlier"])


@pytest.mark.parametrize("name", list(cases.BA_CASES))
def test_ba_oracle_equals_reference_fixture(golden, name):
    build, cur = cases.BA_CASES[name]
    pr = build()
    check_ba(golden_ba(golden, name, pr), ol.local_ba(pr, TUM3), pr)


def test_edges_bit_exact_vs_reference_fixture(golden):
    n, seed = cases.EDGE_CASES
    got = ol.pose_edges_eval(ol._edge_cases(n, seed), TUM3)
    want = golden["edges/rec"]
    assert got.shape == want.shape
    assert np.array_equal(got, want)          # errors, chi2 and Jacobians (analytic and numeric) of all 12 classes, every bit


# ---- live runs of the real reference (skipped where oracle/_ref/ref_opt was not built / did not travel) ----
@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/ref_opt not built")
def test_fixtures_are_what_the_reference_produces_now(golden):
    b = cases.POSE_CASES["c4_b8"][0]()
    for mode in (0, 1):
        r = ol.run_ref_pose(b, TUM3, mode)
        assert np.array_equal(r["Tcw"], golden[f"pose/c4_b8/{mode}/Tcw"])
    pr = cases.BA_CASES["small"][0]()
    r = ol.run_ref_local_ba(pr, TUM3, cases.BA_CASES["small"][1])
    assert np.array_equal(r["kf_Tcw"], golden["ba/small/kf_Tcw"]) and np.array_equal(r["lm"], golden["ba/small/lm"])
    e, H = ol.run_ref_edges(ol._edge_cases(*cases.EDGE_CASES), TUM3)
    assert np.array_equal(e, golden["edges/rec"])


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/ref_opt not built")
@pytest.mark.parametrize("seed", [501, 502, 503])
def test_pose_oracle_vs_live_reference_fresh_seeds(seed):
    from planarslam_amd.synth import pose_batch
    rng = np.random.default_rng(seed)
    b = pose_batch(B=5, n_points=int(rng.integers(50, 1200)), n_lines=int(rng.integers(0, 80)), n_planes=int(rng.integers(0, 8)), seed=seed,
                   outlier_frac=float(rng.uniform(0, 0.3)))
    for mode in (0, 1):
        _check_pose(ol.run_ref_pose(b, TUM3, mode), ol.pose_optimize(b, TUM3, mode), b)


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/ref_opt not built")
def test_ba_reference_rejects_graphs_it_cannot_express():
    # a line edge hanging on its observer (not on the current keyframe) is not something LocalBundleAdjustment can build (Optimizer.cc:2170-2194)
    import subprocess
    from planarslam_amd.synth import ba_local_only, ba_problem
    pr = ba_local_only(ba_problem(seed=5, n_points=50, n_lines=10, n_planes=0))
    with pytest.raises(subprocess.CalledProcessError):
        ol.run_ref_local_ba(pr, TUM3, 9)
