"""CPU: oracle/guided_oracle.cpp fuse_search pinned against the REAL ORBmatcher::Fuse (src/ORBmatcher.cc:829-979 compiled where it lies into
oracle/_ref/ref_match together with the reference's own KeyFrame::GetFeaturesInArea / IsInImage / pose getters and MapPoint::PredictScale) and
against the committed fixture generated from that binary (tests/golden/fuse_points_ref.npz, tools/gen_golden_fuse.py)."""
import os

import numpy as np
import pytest

import fuse_cases as cases
import oracle_lib as O

HAVE_REF = os.path.exists(O.ref_match_path())
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fuse_points_ref.npz")
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/ref_match not built (reference tree absent)")


@needs_ref
@pytest.mark.parametrize("th", [3.0, 1.5])
def test_fuse_search_vs_real_reference(th):
    kf, mp = cases.fuse_case(seed=131)
    lsf, nlev = cases.scale()
    state, kobs = cases.kf_map_points(kf)
    idx, dist, nf = O.fuse_search(kf, mp, th, lsf, nlev)
    for b in range(kf["keys_un"].shape[0]):
        ridx, rn = O.ref_fuse(kf, mp, b, th, lsf, nlev, kf_state=state[b], kf_obs=kobs[b])
        n = int(mp["n"][b])
        np.testing.assert_array_equal(idx[b, :n], ridx)
        assert rn == nf[b] == (ridx >= 0).sum()
        assert rn > 0.3 * n                                   # the case exercises the match branch
        assert (idx[b, :n] < 0).sum() > 0.2 * n               # ... and the gates


@needs_ref
def test_fuse_search_shared_point_list_vs_real_reference():
    """LocalMapping::SearchInNeighbors fuses ONE list of map points into every neighbour key frame"""
    kf, mp = cases.fuse_case(seed=141, B=3, n_points=2000)
    lsf, nlev = cases.scale()
    one = {k: (v[:1] if isinstance(v, np.ndarray) and v.ndim >= 1 and k != "n" else v) for k, v in mp.items()}
    one["n"] = mp["n"][:1]
    idx, dist, nf = O.fuse_search(kf, one, 3.0, lsf, nlev, shared=True)
    for b in range(3):
        ridx, rn = O.ref_fuse(kf, one, b, 3.0, lsf, nlev, shared=True)
        np.testing.assert_array_equal(idx[b, :int(one["n"][0])], ridx)
        assert rn == nf[b]
    assert nf[0] > nf[1:].max()                               # the list was made from key frame 0's keypoints


def test_fuse_search_vs_reference_fixture():
    g = np.load(GOLD)
    kf, mp = cases.fuse_case(seed=131)
    lsf, nlev = cases.scale()
    for th in (3.0, 1.5):
        idx, dist, nf = O.fuse_search(kf, mp, th, lsf, nlev)
        np.testing.assert_array_equal(idx, g[f"fuse_idx_th{th}"])
        np.testing.assert_array_equal(nf, g[f"n_fused_th{th}"])


@needs_ref
@pytest.mark.parametrize("th", [3.0, 6.0])
def test_lsd_fuse_search_vs_real_reference(th):
    kf, lines, ml = cases.fuse_lines_case()
    lsf, nlev = cases.scale()
    rng = np.random.default_rng(5)
    state = rng.choice([0, 1, 2], lines["keylines"].shape, p=[0.5, 0.4, 0.1]).astype(np.uint8); kobs = rng.integers(1, 9, state.shape).astype(np.int32)
    idx, dist, nf = O.lsd_fuse_search(kf, lines, ml, th, lsf, nlev)
    # the real function indexes mvScaleFactors with an unclamped predicted level: lines whose level leaves the pyramid are kept out of its input
    # (the product and the oracle skip them; tests/test_fuse_gpu.py covers them against the oracle)
    for b in range(kf["B"]):
        n = int(ml["n"][b])
        T = kf["Tcw"][b].reshape(4, 4).astype(np.float64)
        Ow = -T[:3, :3].T @ T[:3, 3]
        mid = 0.5 * (ml["xw6"][b, :n, :3] + ml["xw6"][b, :n, 3:])
        q = np.log(ml["max_dist"][b, :n] / np.linalg.norm(mid - Ow, axis=1)) / lsf        # the predicted level is ceil(q)
        unsafe = (q <= -1 + 1e-3) | (q > nlev - 1 - 1e-3)
        assert 0 < unsafe.sum() < 0.5 * n
        safe = dict(ml); safe["usable"] = ml["usable"].copy(); safe["usable"][b, :n][unsafe] = 0
        o2, d2, n2 = O.lsd_fuse_search(kf, lines, safe, th, lsf, nlev)
        ridx, rn = O.ref_lsd_fuse(kf, lines, safe, b, th, lsf, nlev, kf_state=state[b], kf_obs=kobs[b])
        np.testing.assert_array_equal(o2[b, :n], ridx)
        assert rn == n2[b] == (ridx >= 0).sum()
        np.testing.assert_array_equal(o2[b, :n], idx[b, :n])                 # dropping them changes nothing: the oracle skipped them anyway
        assert rn > 0.2 * n and (ridx < 0).sum() > 0.2 * n


def test_lsd_fuse_search_vs_reference_fixture():
    g = np.load(GOLD)
    kf, lines, ml = cases.fuse_lines_case()
    lsf, nlev = cases.scale()
    for th in (3.0, 6.0):
        idx, dist, nf = O.lsd_fuse_search(kf, lines, ml, th, lsf, nlev)
        np.testing.assert_array_equal(idx, g[f"lsd_fuse_idx_th{th}"])
        np.testing.assert_array_equal(nf, g[f"lsd_n_fused_th{th}"])
