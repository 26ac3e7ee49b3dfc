"""GPU parity of the Frame-side glue of Tracking::Track (planarslam_amd/csrc/frame.hip) through the C ABI.

  planar_stereo_from_rgbd   vs the reference's own Frame::ComputeStereoFromRGBD + UnprojectStereo (tests/golden/frame_ref.npz, built from
                            src/Frame.cc:603-634 as oracle/_ref/ref_frame) and the oracle: bit-exact
  planar_pose_assemble      vs a numpy gather that reads the same Frame fields Optimizer::PoseOptimization reads (src/Optimizer.cc:593-981),
                            then the assembled problem through planar_pose_opt equals the oracle's PoseOptimization on it
  planar_discard_outliers   vs numpy"""
import os

import numpy as np
import pytest

import frame_cases as cases
import oracle_lib as ol
from planarslam_amd.synth import TUM3, pose_batch

pytestmark = pytest.mark.gpu


def test_stereo_from_rgbd_equals_reference_fixture(golden_dir):
    from planarslam_amd.frame import stereo_from_rgbd
    g = np.load(os.path.join(golden_dir, "frame_ref.npz"))
    keys, n, depth, Tcw = cases.stereo_case()
    got = stereo_from_rgbd(keys, n, depth, Tcw, TUM3)
    for b in range(len(n)):
        for k in ("u_right", "depth", "xw"):
            np.testing.assert_array_equal(got[k][b, :n[b]], g[f"stereo/{k}"][b, :n[b]], err_msg=f"frame {b} {k}")
        o = ol.stereo_from_rgbd(keys[b, :n[b]], depth[b], Tcw[b], TUM3)
        np.testing.assert_array_equal(got["valid"][b, :n[b]], o["valid"])
        assert not got["valid"][b, n[b]:].any() and (got["u_right"][b, n[b]:] == -1).all()


def _matches_from_pose_batch(pb, seed=5):
    """Turn a synth.pose_batch into the Frame-side view: keypoints + match indices into shuffled map arrays (so the gather is exercised)."""
    from planarslam_amd._lib import KP_DTYPE
    rng = np.random.default_rng(seed)
    B, MP = pb["pt_valid"].shape
    ML, MM = pb["ln_valid"].shape[1], pb["pl_valid"].shape[1]
    sig = (1.0 / (1.2 ** np.arange(8)).astype(np.float32) ** 2).astype(np.float32)
    keys = np.zeros((B, MP), KP_DTYPE)
    keys["x"] = pb["pt_obs"][..., 0]; keys["y"] = pb["pt_obs"][..., 1]
    oct_ = np.argmin(np.abs(pb["pt_inv_sigma2"][..., None] - sig[None, None]), -1)
    keys["octave"] = oct_
    SM = MP + 50
    perm = np.stack([rng.permutation(SM) for _ in range(B)])
    mp_xw = np.zeros((B, SM, 3), np.float32); mp_valid = np.ones((B, SM), np.uint8)
    pt_match = np.full((B, MP), -1, np.int32)
    for b in range(B):
        for i in range(pb["n_points"][b]):
            if pb["pt_valid"][b, i]:
                pt_match[b, i] = perm[b, i]; mp_xw[b, perm[b, i]] = pb["pt_xw"][b, i]
        mp_valid[b, perm[b, MP:]] = 0
    SL = ML + 7
    permL = np.stack([rng.permutation(SL) for _ in range(B)])
    ml = np.zeros((B, SL, 6)); ln_match = np.full((B, ML), -1, np.int32)
    for b in range(B):
        for i in range(pb["n_lines"][b]):
            if pb["ln_valid"][b, i]:
                ln_match[b, i] = permL[b, i]; ml[b, permL[b, i]] = pb["ln_xw"][b, i]
    SP = 3 * MM + 2
    mpl = np.zeros((B, SP, 4), np.float32); pl_match = np.full((3, B, MM), -1, np.int32)
    for b in range(B):
        for i in range(pb["n_planes"][b]):
            for k in range(3):
                if pb["pl_valid"][b, i, k]:
                    pl_match[k, b, i] = 3 * i + k; mpl[b, 3 * i + k] = pb["pl_world"][b, i, k]
    m = dict(n=pb["n_points"], keys_un=keys, u_right=pb["pt_obs"][..., 2].copy(), pt_match=pt_match, mp_xw=mp_xw, mp_valid=mp_valid, inv_level_sigma2=sig,
             n_lines=pb["n_lines"], line_eq=pb["ln_obs"], ln_match=ln_match, ml_xw6=ml, n_planes=pb["n_planes"], pl_coef=pb["pl_meas"], pl_match=pl_match,
             mpl_coef=mpl, Tcw=pb["Tcw"])
    return m, sig[oct_]


@pytest.mark.parametrize("mode", [0, 1])
def test_pose_assemble_then_optimise(mode):
    from planarslam_amd import Optimizer
    from planarslam_amd.frame import pose_assemble
    pb = pose_batch(B=5, n_points=700, n_lines=30, n_planes=5, seed=61, max_points=800, max_lines=40, max_planes=6)
    pb["n_points"][:] = [700, 650, 3, 700, 420]; pb["n_lines"][:] = [30, 0, 2, 29, 11]; pb["n_planes"][:] = [5, 2, 0, 5, 1]
    m, is2 = _matches_from_pose_batch(pb)
    got = pose_assemble(m, 800, 40, 6)
    for b in range(5):
        npt, nl, npl = pb["n_points"][b], pb["n_lines"][b], pb["n_planes"][b]
        assert (got["n_points"][b], got["n_lines"][b], got["n_planes"][b]) == (npt, nl, npl)
        v = pb["pt_valid"][b, :npt] > 0
        np.testing.assert_array_equal(got["pt_valid"][b, :npt], pb["pt_valid"][b, :npt])
        np.testing.assert_array_equal(got["pt_xw"][b, :npt][v], pb["pt_xw"][b, :npt][v])
        np.testing.assert_array_equal(got["pt_obs"][b, :npt], pb["pt_obs"][b, :npt])
        np.testing.assert_array_equal(got["pt_inv_sigma2"][b, :npt], is2[b, :npt])
        assert not got["pt_valid"][b, npt:].any()
        lv = pb["ln_valid"][b, :nl] > 0
        np.testing.assert_array_equal(got["ln_valid"][b, :nl], pb["ln_valid"][b, :nl])
        np.testing.assert_array_equal(got["ln_obs"][b, :nl], pb["ln_obs"][b, :nl])
        np.testing.assert_array_equal(got["ln_xw"][b, :nl][lv], pb["ln_xw"][b, :nl][lv])
        np.testing.assert_array_equal(got["pl_valid"][b, :npl], pb["pl_valid"][b, :npl])
        np.testing.assert_array_equal(got["pl_meas"][b, :npl], pb["pl_meas"][b, :npl])
        np.testing.assert_array_equal(got["pl_world"][b, :npl][pb["pl_valid"][b, :npl] > 0], pb["pl_world"][b, :npl][pb["pl_valid"][b, :npl] > 0])
    np.testing.assert_array_equal(got["Tcw"], pb["Tcw"])
    # the assembled problem through the optimiser: same result as the oracle on it
    opt = Optimizer(TUM3)
    res = (opt.PoseOptimization if mode == 0 else opt.TranslationOptimization)(got, 4, 10)
    want = ol.pose_optimize(got, TUM3, mode, 4, 10)
    assert np.abs(res["Tcw"] - want["Tcw"]).max() <= 1e-5 and np.array_equal(res["n_inliers"], want["n_inliers"])
    assert np.array_equal(res["pt_outlier"], want["pt_outlier"])


def test_pose_assemble_points_only_and_clipping():
    from planarslam_amd.frame import pose_assemble
    pb = pose_batch(B=2, n_points=300, n_lines=0, n_planes=0, seed=62, max_lines=1, max_planes=1)
    m, _ = _matches_from_pose_batch(pb)
    m["n_lines"] = None; m["n_planes"] = None
    got = pose_assemble(m, 256, 0, 0)           # capacity below Frame::N: the first 256 keypoints
    assert (got["n_points"] == 256).all() and (got["n_lines"] == 0).all() and (got["n_planes"] == 0).all()
    np.testing.assert_array_equal(got["pt_obs"], pb["pt_obs"][:, :256])


def test_discard_outliers():
    from planarslam_amd.frame import discard_outliers
    rng = np.random.default_rng(9)
    B, S = 4, 1100
    n = np.array([1100, 900, 0, 513], np.int32)
    match = rng.integers(-1, 50, (B, S)).astype(np.int32)
    outl = (rng.random((B, 1000)) < 0.3).astype(np.uint8)
    m, o, kept = discard_outliers(n, match, outl)
    for b in range(B):
        wm, wo = match[b].copy(), outl[b].copy()
        k = 0
        for i in range(n[b]):
            if wm[i] >= 0:
                if i < 1000 and wo[i]:
                    wm[i] = -1; wo[i] = 0
                else:
                    k += 1
        np.testing.assert_array_equal(m[b], wm); np.testing.assert_array_equal(o[b], wo)
        assert kept[b] == k


def test_undistort_keypoints_equals_the_oracle():
    """Frame::UndistortKeyPoints on the device == the oracle's literal restatement of cv::undistortPoints, bit for bit, for the TUM1/TUM2 coefficients of the
    reference's yaml files; TUM3 (k1 = 0) copies; rows past n[b] stay zero."""
    from planarslam_amd import frame as F
    keys, n = cases.undistort_case()
    for name, (K, D) in cases.DIST.items():
        cam = dict(fx=K[0], fy=K[1], cx=K[2], cy=K[3])
        got = F.undistort_keypoints(keys, n, cam, D)
        for b in range(len(n)):
            want = ol.undistort_keypoints(keys[b, :n[b]], cam, D)
            assert got[b, :n[b]].tobytes() == want.tobytes(), (name, b)
            assert not got[b, n[b]:].tobytes().strip(b"\0")
        if D[0] == 0:
            assert got[0].tobytes() == keys[0].tobytes()
