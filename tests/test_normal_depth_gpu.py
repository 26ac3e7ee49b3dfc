"""GPU parity (bit-exact): MapPoint::UpdateNormalAndDepth through planar_update_normal_and_depth against the committed outputs of the reference's own function
(tests/golden/normal_depth_ref.npz = oracle/_ref/ref_frame normal_depth) and the oracle."""
import os

import numpy as np
import pytest

import normal_depth_cases as nc
import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "normal_depth_ref.npz"))["out"]


def test_matches_reference_fixture():
    from planarslam_amd._lib import KP_DTYPE
    from planarslam_amd.frame import update_normal_and_depth
    Tcws, sf, pts = nc.cases()
    ow = np.stack([ol.keyframe_center(T) for T in Tcws])
    P = len(pts)
    xw = np.stack([p[0] for p in pts]).reshape(P, 1, 3)              # one group per point: its reference key frame's pose
    ref_T = np.stack([Tcws[p[1]] for p in pts])
    keys = np.zeros((P, 1), KP_DTYPE); keys["octave"][:, 0] = [p[2] for p in pts]
    off = np.zeros(P + 1, np.int32); off[1:] = np.cumsum([len(p[3]) for p in pts])
    obs_ow = np.concatenate([ow[p[3]] for p in pts])
    nrm, mn, mx = update_normal_and_depth(np.ones(P, np.int32), xw, ref_T, keys, sf, obs_off=off, obs_ow=obs_ow)
    got = np.concatenate([nrm[:, 0], mn, mx], 1).astype(np.float32)
    assert np.array_equal(got.view(np.int32), GOLD.view(np.int32))


def test_single_observer_groups_and_validity():
    """The form the tracking pipeline uses: every point observed by its reference key frame only, invalid points untouched."""
    from planarslam_amd._lib import KP_DTYPE
    from planarslam_amd.frame import update_normal_and_depth
    rng = np.random.default_rng(2)
    Tcws, sf, _ = nc.cases()
    G, S = 5, 300
    xw = rng.normal(scale=3.0, size=(G, S, 3)).astype(np.float32)
    n = np.array([300, 0, 17, 299, 64], np.int32)
    valid = (rng.uniform(size=(G, S)) < 0.8).astype(np.uint8)
    keys = np.zeros((G, S), KP_DTYPE); keys["octave"] = rng.integers(0, len(sf), size=(G, S))
    nrm, mn, mx = update_normal_and_depth(n, xw, Tcws[:G], keys, sf, valid=valid)
    for g in range(G):
        c = ol.keyframe_center(Tcws[g])
        for i in range(0, S, 7):
            if i < n[g] and valid[g, i]:
                wn, wmn, wmx = ol.update_normal_and_depth(xw[g, i], c[None], c, keys["octave"][g, i], sf)
                assert np.array_equal(nrm[g, i], wn) and mn[g, i] == wmn and mx[g, i] == wmx
            else:
                assert not nrm[g, i].any() and mn[g, i] == 0 and mx[g, i] == 0
