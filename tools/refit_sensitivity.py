"""How much Frame::MaxPointDistanceFromPlane's RANSAC + single-pass float covariance refit (oracle/planepost_oracle.cpp, i.e. PCL's SACSegmentation as
the reference calls it) moves when every coordinate of its input cloud moves by at most one float ulp.  This is the noise floor any end-to-end comparison
of mvPlaneCoefficients has: PCL's VoxelGrid sums floats in std::sort's unstable order, so a different standard library already moves the centroids that much.

    PYTHONPATH=.:tests python tools/refit_sensitivity.py        (CPU only)"""
import numpy as np

import oracle_lib as ol
from planarslam_amd.synth import depth_image

rng = np.random.default_rng(0)
worst = 0.0
for seed in range(50, 56):
    d = depth_image(seed, noise=(seed % 2 == 0), holes=(seed % 3 != 0))
    planes, labels = ol.peac_run(d)
    r = ol.plane_clouds(d, labels, planes)
    for k, p in enumerate(r["src"]):
        cloud = r["points"][r["pt_off"][k]:r["pt_off"][k + 1]]
        P = planes[p]
        c0 = np.array([P[1], P[2], P[3], -(P[1] * P[4] + P[2] * P[5] + P[3] * P[6])]).astype(np.float32)
        _, pl0, _ = ol.plane_refit(c0, cloud, 0.05)
        dm = 0.0
        for _ in range(20):
            pert = (cloud.view(np.int32) + rng.integers(-1, 2, size=cloud.shape).astype(np.int32)).view(np.float32)
            st, pl, _ = ol.plane_refit(c0, pert, 0.05)
            if st == 0:
                dm = max(dm, float(np.abs(pl - pl0).max()))
        worst = max(worst, dm)
        print(f"frame {seed} plane {p}: {len(cloud):5d} voxels, coefficient moves by up to {dm:.2e}")
print(f"worst: {worst:.2e}")
