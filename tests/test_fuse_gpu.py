"""GPU: planar_fuse_search (guided.hip fuse_kernel) = ORBmatcher::Fuse(pKF, vpMapPoints, th), the search half (reference src/ORBmatcher.cc:829-951),
against the oracle (pinned to the real function in tests/test_oracle_fuse.py) and against the fixture the real function produced."""
import os

import numpy as np
import pytest

import fuse_cases as cases
import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fuse_points_ref.npz")


@pytest.fixture(scope="module")
def ctx():
    from planarslam_amd import Context
    return Context(0)


@pytest.mark.parametrize("th", [3.0, 1.5])
def test_fuse_search_matches_oracle_and_reference_fixture(ctx, th):
    from planarslam_amd.guided import ORBmatcher
    kf, mp = cases.fuse_case(seed=131)
    lsf, nlev = cases.scale()
    idx, dist, nf = ORBmatcher(ctx=ctx).Fuse(kf, mp, th)
    oidx, odist, onf = O.fuse_search(kf, mp, th, lsf, nlev)
    np.testing.assert_array_equal(idx, oidx); np.testing.assert_array_equal(dist, odist); np.testing.assert_array_equal(nf, onf)
    g = np.load(GOLD)
    np.testing.assert_array_equal(idx, g[f"fuse_idx_th{th}"]); np.testing.assert_array_equal(nf, g[f"n_fused_th{th}"])
    assert nf.min() > 1000


def test_fuse_search_shared_list_ragged_and_empty(ctx):
    from planarslam_amd.guided import ORBmatcher
    kf, mp = cases.fuse_case(seed=151, B=4, N=900, n_points=1500, hit=0.8)
    lsf, nlev = cases.scale()
    one = {k: (v[:1] if isinstance(v, np.ndarray) and k != "n" else v) for k, v in mp.items()}
    one["n"] = mp["n"][:1]
    idx, dist, nf = ORBmatcher(ctx=ctx).Fuse(kf, one, 3.0, shared=True)
    oidx, odist, onf = O.fuse_search(kf, one, 3.0, lsf, nlev, shared=True)
    np.testing.assert_array_equal(idx, oidx); np.testing.assert_array_equal(dist, odist); np.testing.assert_array_equal(nf, onf)
    # a key frame without keypoints, a list without usable points, an empty list
    kf2 = dict(kf); kf2["n"] = kf["n"].copy(); kf2["n"][1] = 0
    mp2 = dict(mp); mp2["usable"] = mp["usable"].copy(); mp2["usable"][2] = 0; mp2["n"] = mp["n"].copy(); mp2["n"][3] = 0
    idx, dist, nf = ORBmatcher(ctx=ctx).Fuse(kf2, mp2, 3.0)
    oidx, odist, onf = O.fuse_search(kf2, mp2, 3.0, lsf, nlev)
    for b in range(4):
        n = int(mp2["n"][b])
        np.testing.assert_array_equal(idx[b, :n], oidx[b, :n]); np.testing.assert_array_equal(dist[b, :n], odist[b, :n])
    np.testing.assert_array_equal(nf, onf)
    assert nf[0] > 0 and nf[1] == nf[2] == nf[3] == 0


def test_fuse_search_many_key_frames(ctx):
    """one workgroup per key frame: more key frames than CUs, levels / th as LocalMapping uses them"""
    from planarslam_amd.guided import ORBmatcher
    kf, mp = cases.fuse_case(seed=161, B=300, N=400, n_points=500, hit=0.7)
    lsf, nlev = cases.scale()
    idx, dist, nf = ORBmatcher(ctx=ctx).Fuse(kf, mp, 3.0)
    oidx, odist, onf = O.fuse_search(kf, mp, 3.0, lsf, nlev)
    np.testing.assert_array_equal(idx, oidx); np.testing.assert_array_equal(nf, onf)


@pytest.mark.parametrize("th", [3.0, 6.0])
def test_lsd_fuse_search_matches_oracle_and_reference_fixture(ctx, th):
    """planar_lsd_fuse_search = LSDmatcher::Fuse(pKF, vpMapLines, th), the search half (reference src/LSDmatcher.cpp:884-991)"""
    from planarslam_amd.guided import lsd_fuse
    kf, lines, ml = cases.fuse_lines_case()
    lsf, nlev = cases.scale()
    idx, dist, nf = lsd_fuse(kf, lines, ml, th, ctx=ctx)
    oidx, odist, onf = O.lsd_fuse_search(kf, lines, ml, th, lsf, nlev)
    np.testing.assert_array_equal(idx, oidx); np.testing.assert_array_equal(dist, odist); np.testing.assert_array_equal(nf, onf)
    g = np.load(GOLD)
    np.testing.assert_array_equal(idx, g[f"lsd_fuse_idx_th{th}"]); np.testing.assert_array_equal(nf, g[f"lsd_n_fused_th{th}"])
    assert nf.min() > 100


def test_lsd_fuse_search_many_key_frames_and_empty(ctx):
    from planarslam_amd.guided import lsd_fuse
    kf, lines, ml = cases.fuse_lines_case(seed=181, B=300, n_lines=80, n_ml=200)
    lines["n"][3] = 0; ml["n"][5] = 0; ml["usable"][7] = 0
    lsf, nlev = cases.scale()
    idx, dist, nf = lsd_fuse(kf, lines, ml, 3.0, ctx=ctx)
    oidx, odist, onf = O.lsd_fuse_search(kf, lines, ml, 3.0, lsf, nlev)
    for b in range(300):
        n = int(ml["n"][b])
        np.testing.assert_array_equal(idx[b, :n], oidx[b, :n]); np.testing.assert_array_equal(dist[b, :n], odist[b, :n])
    np.testing.assert_array_equal(nf, onf)
    assert nf[3] == nf[5] == nf[7] == 0 and nf.sum() > 10000
