"""CPU property tests of the local-BA oracle (oracle/ba_oracle.cpp).  g2o/Eigen cannot be built here (parity unpinned,
DESIGN.md), so these check the restatement's behaviour, not the reference binary."""
import numpy as np

import oracle_lib as ol
from planarslam_amd.synth import TUM3, ba_problem


def _pose_err(T, Tg):
    return max(np.abs(T[k].reshape(4, 4)[:3, 3] - Tg[k][:3, 3]).max() for k in range(len(Tg)))


def test_ba_reduces_pose_error_and_keeps_fixed_keyframes():
    pr = ba_problem(seed=5, n_points=300, n_lines=60, n_planes=12)
    r = ol.local_ba(pr, TUM3)
    assert _pose_err(r["kf_Tcw"], pr["T_gt"]) < 0.25 * _pose_err(pr["kf_Tcw"], pr["T_gt"])
    fixed = pr["kf_fixed"] == 1
    # fixed vertices only go through the float32 <-> quaternion round trip
    assert np.abs(r["kf_Tcw"][fixed] - pr["kf_Tcw"][fixed]).max() < 1e-6
    assert 0 < r["e_outlier"].sum() < 0.2 * len(pr["e_kf"])


def test_ba_line_edge_pairs_share_their_fate():
    pr = ba_problem(seed=6, n_points=100, n_lines=80, n_planes=6)
    r = ol.local_ba(pr, TUM3)
    idx = np.nonzero(pr["e_type"] == 2)[0]
    assert len(idx) % 2 == 0
    assert np.array_equal(r["e_outlier"][idx[0::2]], r["e_outlier"][idx[1::2]])


def test_ba_noise_free_problem_stays_at_ground_truth():
    pr = ba_problem(seed=7, n_points=150, n_lines=0, n_planes=0, outlier_frac=0.0, pose_noise=(0.0, 0.0), point_noise=0.0)
    r = ol.local_ba(pr, TUM3)
    # observations are noisy, so the optimum moves a little, but far less than a perturbed start would
    assert _pose_err(r["kf_Tcw"], pr["T_gt"]) < 0.05
