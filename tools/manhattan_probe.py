"""How far the Manhattan rotation (src/Tracking.cc:250-253) is from the tracked rotation on the synthetic streams, and what TranslationOptimization keeps (developer probe)."""
import sys

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_track_gpu as tt  # noqa: E402

run = tt.run_pipeline()
for which in (0, 1):
    c = run["cap"][tt.STEPS - 2 + which]
    T = c["pbT"]
    gap = np.abs(T["Tcw_in"].reshape(-1, 4, 4)[:, :3, :3] - c["pose_in"].reshape(-1, 4, 4)[:, :3, :3]).max((1, 2))
    ang = np.degrees(np.arccos(np.clip((np.einsum("nij,nij->n", T["Tcw_in"].reshape(-1, 4, 4)[:, :3, :3], c["pose_in"].reshape(-1, 4, 4)[:, :3, :3]) - 1) / 2, -1, 1)))
    d0 = np.abs(c["Rcm_new"] - c["Rcm0"]).max(1)
    print(f"step {which}: rotation gap max {gap.max():.4f} median {np.median(gap):.4f}; angle deg max {ang.max():.2f} median {np.median(ang):.3f}; |Rcm_new - Rcm0| max {d0.max():.4f} median {np.median(d0):.5f}")
    print("   translation-opt inliers: mean %.1f min %d; pose-opt inliers mean %.1f min %d; projection matches mean %.1f" % (T["n_inliers"].mean(), T["n_inliers"].min(), c["pbP"]["n_inliers"].mean(), c["pbP"]["n_inliers"].min(), c["nm"].mean()))
    worst = np.argsort(-ang)[:5]
    print("   worst frames", worst.tolist(), "angles", np.round(ang[worst], 2).tolist(), "inliers", T["n_inliers"][worst].tolist())

# against the truth: the canvases are box rooms seen through a rotation R (synth.depth_image): the Manhattan axes in the camera frame are the rows of R
from planarslam_amd.synth import _rodrigues  # noqa: E402


def axis_err(Rest, Rtrue):
    """smallest angle (deg) between every estimated axis and the nearest true axis (sign-free), max over the three"""
    M = np.abs(Rest.T @ Rtrue)            # columns of Rest vs columns of Rtrue
    return float(np.degrees(np.arccos(np.clip(M.max(1), -1, 1))).max())


truth = []
for i in range(16):
    rng = np.random.default_rng(4321 + 3 * 4096 + i)
    truth.append(_rodrigues(rng.normal(size=3) * 0.15).T)      # camera <- room
for which in (0, 1):
    c = run["cap"][tt.STEPS - 2 + which]
    e0 = np.array([axis_err(c["Rcm0"][b].reshape(3, 3), truth[b % 16]) for b in range(tt.B)])
    e1 = np.array([axis_err(c["Rcm_new"][b].reshape(3, 3), truth[b % 16]) for b in range(tt.B)])
    print(f"step {which}: axis error vs the true room axes (deg): Rotation_cm (first frame) median {np.median(e0):.2f} max {e0.max():.2f}; this frame's MF_can median {np.median(e1):.2f} max {e1.max():.2f}")
    print("   per canvas (first 16 streams) Rotation_cm:", np.round(e0[:16], 1).tolist())
    print("   per canvas (first 16 streams) MF_can     :", np.round(e1[:16], 1).tolist())
