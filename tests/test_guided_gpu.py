"""GPU parity (bit-exact): the HIP guided matchers through the C ABI against oracle/guided_oracle.cpp.
Rows a20, a21, a22, a24, a25 of SURVEY.md §8."""
import numpy as np
import pytest

import oracle_lib as O
from planarslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from planarslam_amd._lib import Context
    return Context(0)


@pytest.mark.parametrize("motion", [(0, 0, 0), (0, 0, 0.3), (0, 0, -0.3)])
@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_projection_frame(ctx, motion, check_ori):
    from planarslam_amd.guided import ORBmatcher
    fr = synth.guided_frame(B=4, N=1000, seed=61)
    cur, last = synth.guided_last_frame(fr, seed=62, motion=motion)
    for th, mono in ((15.0, False), (7.0, False), (30.0, True)):
        ref_m, ref_n = O.search_by_projection_frame(cur, last, th, mono=mono, check_orientation=check_ori)
        m, n = ORBmatcher(0.9, check_ori, ctx).SearchByProjectionFrame(cur, last, th, bMono=mono)
        np.testing.assert_array_equal(n, ref_n)
        np.testing.assert_array_equal(m, ref_m)
        assert n.min() > 100


def test_search_by_projection_frame_inout_and_ragged(ctx):
    from planarslam_amd.guided import ORBmatcher
    fr = synth.guided_frame(B=3, N=700, stride=1024, seed=63, crowd=0.6)
    cur, last = synth.guided_last_frame(fr, seed=64, dup=0.4)
    init = np.full((3, 1024), 777, np.int32)
    ref_m, ref_n = O.search_by_projection_frame(cur, last, 15.0, cur_match=init)
    m, n = ORBmatcher(0.9, True, ctx).SearchByProjectionFrame(cur, last, 15.0, cur_match=init)
    np.testing.assert_array_equal(m, ref_m); np.testing.assert_array_equal(n, ref_n)
    assert (m == 777).any() and (m == -1).any()
    # empty last frame / empty current frame
    last0 = dict(last); last0["n"] = np.zeros(3, np.int32)
    m0, n0 = ORBmatcher(0.9, True, ctx).SearchByProjectionFrame(cur, last0, 15.0)
    assert (n0 == 0).all() and (m0 == -1).all()
    cur0 = dict(cur); cur0["n"] = np.zeros(3, np.int32)
    m0, n0 = ORBmatcher(0.9, True, ctx).SearchByProjectionFrame(cur0, last, 15.0)
    assert (n0 == 0).all()


@pytest.mark.parametrize("th", [1.0, 3.0, 5.0])
def test_search_by_projection_map(ctx, th):
    from planarslam_amd.guided import ORBmatcher
    fr = synth.guided_frame(B=4, N=1000, seed=71, crowd=0.5)
    fr, pr = synth.guided_map_probes(fr, seed=72, n_probes=3000)
    for ratio in (0.8, 0.6):
        ref_m, ref_n = O.search_by_projection_map(fr, pr, th=th, nn_ratio=ratio)
        m, n = ORBmatcher(ratio, True, ctx).SearchByProjectionMap(fr, pr, th=th)
        np.testing.assert_array_equal(n, ref_n)
        np.testing.assert_array_equal(m, ref_m)
    assert n.min() > 100


def test_search_by_projection_map_dense_windows(ctx):
    """Windows holding more candidates than one chunk of the LDS list (adaptive chunking path)."""
    from planarslam_amd.guided import ORBmatcher
    fr = synth.guided_frame(B=2, N=4000, seed=73, crowd=0.9)
    fr, pr = synth.guided_map_probes(fr, seed=74, n_probes=2500)
    ref_m, ref_n = O.search_by_projection_map(fr, pr, th=12.0, nn_ratio=0.8)
    m, n = ORBmatcher(0.8, True, ctx).SearchByProjectionMap(fr, pr, th=12.0)
    np.testing.assert_array_equal(n, ref_n)
    np.testing.assert_array_equal(m, ref_m)


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow(ctx, check_ori):
    from planarslam_amd.guided import ORBmatcher
    for N, nodes, seed in ((1000, 90, 81), (1500, 12, 82), (300, 400, 83)):
        kf, f = synth.guided_bow(B=3, N=N, seed=seed, n_nodes=nodes)
        ref_m, ref_n = O.search_by_bow(kf, f, nn_ratio=0.7, check_orientation=check_ori)
        m, n = ORBmatcher(0.7, check_ori, ctx).SearchByBoW(kf, f)
        np.testing.assert_array_equal(n, ref_n)
        np.testing.assert_array_equal(m, ref_m)
    assert n.min() > 20


def test_lsd_search_by_projection(ctx):
    from planarslam_amd.guided import LSDmatcher
    for n_lines, n_ml, seed in ((40, 120, 91), (200, 500, 92), (3, 10, 93)):
        lines, ml = synth.guided_lines(B=4, n_lines=n_lines, n_ml=n_ml, seed=seed)
        for th in (1.0, 3.0):
            ref_m, ref_n = O.lsd_search_by_projection(lines, ml, synth.scale_factors(), th=th, nn_ratio=0.6)
            m, n = LSDmatcher(0.6, ctx).SearchByProjection(lines, ml, synth.scale_factors(), th=th)
            np.testing.assert_array_equal(n, ref_n)
            np.testing.assert_array_equal(m, ref_m)


@pytest.mark.parametrize("shared", [False, True])
def test_plane_search_by_coefficients(ctx, shared):
    from planarslam_amd.guided import PlaneMatcher
    fr, mp = synth.guided_planes(B=6, n_planes=10, n_map=50, n_pts=700, seed=95, shared=shared)
    ref = O.plane_search_by_coefficients(fr, mp)
    got = PlaneMatcher(ctx=ctx).SearchMapByCoefficients(fr, mp)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    assert ref[3].sum() > 10
    init = [np.full((6, 10), 5, np.int32)] * 3
    ref = O.plane_search_by_coefficients(fr, mp, init=init)
    got = PlaneMatcher(ctx=ctx).SearchMapByCoefficients(fr, mp, init=init)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)


def test_against_real_reference_fixtures(ctx):
    """tests/golden/guided_ref.npz holds outputs of the REAL reference matchers (oracle/_ref/ref_match, tools/gen_golden_guided.py)."""
    import os
    from planarslam_amd.guided import ORBmatcher, PlaneMatcher
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "guided_ref.npz"))
    seed = int(g["seed"])
    fr = synth.guided_frame(B=2, N=800, seed=seed)
    cur, last = synth.guided_last_frame(fr, seed=seed + 1, dup=0.3)
    m, n = ORBmatcher(0.9, True, ctx).SearchByProjectionFrame(cur, last, 15.0)
    np.testing.assert_array_equal(m, g["proj_frame_match"]); np.testing.assert_array_equal(n, g["proj_frame_n"])
    fr2, pr = synth.guided_map_probes(fr, seed=seed + 2, n_probes=2000)
    m, n = ORBmatcher(0.8, True, ctx).SearchByProjectionMap(fr2, pr, th=3.0)
    np.testing.assert_array_equal(m, g["proj_map_match"]); np.testing.assert_array_equal(n, g["proj_map_n"])
    kf, f = synth.guided_bow(B=2, N=800, seed=seed + 3)
    m, n = ORBmatcher(0.7, True, ctx).SearchByBoW(kf, f)
    np.testing.assert_array_equal(m, g["bow_match"]); np.testing.assert_array_equal(n, g["bow_n"])
    frp, mp = synth.guided_planes(B=4, seed=seed + 4)
    a, v, p, n = PlaneMatcher(ctx=ctx).SearchMapByCoefficients(frp, mp)
    np.testing.assert_array_equal(np.stack([a, v, p]), g["plane_avp"]); np.testing.assert_array_equal(n, g["plane_n"])
    from planarslam_amd.guided import LSDmatcher
    lines, ml = synth.guided_lines(B=3, n_lines=150, n_ml=400, seed=seed + 5)
    m, n = LSDmatcher(0.6, ctx).SearchByProjection(lines, ml, synth.scale_factors(), th=3.0)
    np.testing.assert_array_equal(m, g["lsd_proj_match"]); np.testing.assert_array_equal(n, g["lsd_proj_n"])


def test_is_in_frustum_then_search_by_projection(ctx):
    """TrackLocalMap's front end: Frame::isInFrustum for every local map point / line, then the guided search on its output."""
    from planarslam_amd.guided import Frame, LSDmatcher, ORBmatcher
    fr = synth.guided_frame(B=3, N=1000, seed=121)
    fr, mp, ml = synth.guided_local_map(fr, seed=122, n_points=4000, n_lines=500)
    F = Frame(fr, ctx=ctx)
    ref = O.is_in_frustum_points(fr, mp, F.log_scale_factor, F.n_levels)
    got = F.isInFrustumPoints(mp)
    np.testing.assert_array_equal(got["in_view"], ref["in_view"])
    iv = ref["in_view"] > 0
    assert 0.1 < iv.mean() < 0.9
    for k in ("proj_x", "proj_y", "proj_xr", "level", "view_cos"):
        np.testing.assert_array_equal(got[k][iv], ref[k][iv], err_msg=k)
    refl = O.is_in_frustum_lines(fr, ml, F.log_scale_factor)
    gotl = F.isInFrustumLines(ml)
    np.testing.assert_array_equal(gotl["in_view"], refl["in_view"])
    il = refl["in_view"] > 0
    for k in ("proj", "level", "view_cos"):
        np.testing.assert_array_equal(gotl[k][il], refl[k][il], err_msg=k)
    # the probes feed SearchByProjection unchanged (fields not written for out-of-view points are never read)
    fr2 = dict(fr); fr2["blocked"] = np.zeros(fr["keys_un"].shape, np.uint8)
    m, n = ORBmatcher(0.8, True, ctx).SearchByProjectionMap(fr2, got, th=3.0)
    rm, rn = O.search_by_projection_map(fr2, {**ref, "n": mp["n"], "desc": mp["desc"], "observed": mp["observed"]}, th=3.0, nn_ratio=0.8)
    np.testing.assert_array_equal(m, rm); np.testing.assert_array_equal(n, rn)


def test_is_in_frustum_hip_equals_reference_fixture(ctx, golden_dir):
    """Frame::isInFrustum (points and lines) on the GPU vs the fields the reference's own function bodies wrote
    (tests/golden/frame_ref.npz = src/Frame.cc:296-438 + MapPoint / MapLine::PredictScale built as oracle/_ref/ref_frame): every bit."""
    import os
    import frame_cases as cases
    from planarslam_amd.guided import Frame
    g = np.load(os.path.join(golden_dir, "frame_ref.npz"))
    fr, mp, ml = cases.frustum_case()
    lsf, nlev = cases.frustum_scale()
    F = Frame(fr, log_scale_factor=lsf, n_levels=nlev, ctx=ctx)
    got = F.isInFrustumPoints(mp)
    iv = g["frustum/points/in_view"]
    np.testing.assert_array_equal(got["in_view"], iv)
    for k in ("proj_x", "proj_y", "proj_xr", "level", "view_cos"):
        np.testing.assert_array_equal(got[k][iv > 0], g[f"frustum/points/{k}"][iv > 0], err_msg=k)
    gotl = F.isInFrustumLines(ml)
    il = g["frustum/lines/in_view"]
    np.testing.assert_array_equal(gotl["in_view"], il)
    for k in ("proj", "level", "view_cos"):
        np.testing.assert_array_equal(gotl[k][il > 0], g[f"frustum/lines/{k}"][il > 0], err_msg=k)
