"""CPU: oracle/guided_oracle.cpp (and MatchORBPoints in match_oracle.cpp) pinned against the REAL reference matchers
(src/ORBmatcher.cc, src/LSDmatcher.cpp, src/PlaneMatcher.cpp compiled where they lie into oracle/_ref/ref_match) and against the committed
fixtures generated from that binary (tests/golden/guided_*.npz, tools/gen_golden_guided.py)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from planarslam_amd import synth

HAVE_REF = os.path.exists(O.ref_match_path())
GOLD = os.path.join(os.path.dirname(__file__), "golden")
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/ref_match not built (reference tree absent)")


@needs_ref
@pytest.mark.parametrize("motion", [(0, 0, 0), (0, 0, 0.3), (0, 0, -0.3)])
def test_search_by_projection_frame_vs_real_reference(motion):
    fr = synth.guided_frame(B=2, N=800, seed=101)
    cur, last = synth.guided_last_frame(fr, seed=102, motion=motion, dup=0.3)
    for th, mono, ori in ((15.0, False, True), (7.0, True, True), (15.0, False, False)):
        m, nm = O.search_by_projection_frame(cur, last, th, mono=mono, check_orientation=ori)
        for b in range(2):
            rm, rn = O.ref_search_by_projection_frame(cur, last, b, th, mono, ori)
            n = int(cur["n"][b])
            assert rn == nm[b] and rn > 100
            np.testing.assert_array_equal(m[b, :n], rm)


@needs_ref
@pytest.mark.parametrize("th", [1.0, 3.0])
def test_search_by_projection_map_vs_real_reference(th):
    fr = synth.guided_frame(B=2, N=800, seed=103, crowd=0.5)
    fr, pr = synth.guided_map_probes(fr, seed=104, n_probes=2000)
    for ratio in (0.8, 0.6):
        m, nm = O.search_by_projection_map(fr, pr, th=th, nn_ratio=ratio)
        for b in range(2):
            rm, rn = O.ref_search_by_projection_map(fr, pr, b, th, ratio)
            assert rn == nm[b] and rn > 100
            np.testing.assert_array_equal(m[b, :int(fr["n"][b])], rm)


@needs_ref
def test_search_by_bow_vs_real_reference():
    for N, nodes, seed in ((800, 90, 105), (600, 10, 106)):
        kf, f = synth.guided_bow(B=2, N=N, seed=seed, n_nodes=nodes)
        for ori in (True, False):
            m, nm = O.search_by_bow(kf, f, nn_ratio=0.7, check_orientation=ori)
            for b in range(2):
                rm, rn = O.ref_search_by_bow(kf, f, b, 0.7, ori)
                assert rn == nm[b] and rn > 20
                np.testing.assert_array_equal(m[b, :int(f["n"][b])], rm)


@needs_ref
def test_match_orb_points_vs_real_reference():
    rng = np.random.default_rng(107)
    last = rng.integers(0, 256, (600, 32), dtype=np.uint8)
    cur = synth._flip_bits(rng, last[rng.permutation(600)[:500]], 40)
    has = (rng.random(600) < 0.8).astype(np.uint8); outl = (rng.random(600) < 0.2).astype(np.uint8)
    init = np.full(500, -1, np.int32)
    m, npair = O.match_orb_points(cur, last, has, outl, init)
    rm, rn = O.ref_match_orb_points(cur, last, has, outl)
    assert rn == npair > 100
    np.testing.assert_array_equal(m, rm)


@needs_ref
@pytest.mark.parametrize("shared", [False, True])
def test_plane_matcher_vs_real_reference(shared):
    fr, mp = synth.guided_planes(B=4, n_planes=10, n_map=40, n_pts=500, seed=108, shared=shared)
    a, v, p, n = O.plane_search_by_coefficients(fr, mp)
    for b in range(4):
        ra, rv, rp, rn = O.ref_plane_search(fr, mp, b)
        k = int(fr["n"][b])
        assert rn == n[b]
        np.testing.assert_array_equal(a[b, :k], ra); np.testing.assert_array_equal(v[b, :k], rv); np.testing.assert_array_equal(p[b, :k], rp)
    assert n.sum() > 8


@needs_ref
def test_lsd_matchers_vs_real_reference():
    for n_lines, n_ml, seed in ((40, 120, 109), (150, 400, 110)):
        lines, ml = synth.guided_lines(B=3, n_lines=n_lines, n_ml=n_ml, seed=seed)
        for th in (1.0, 3.0):
            m, nm = O.lsd_search_by_projection(lines, ml, synth.scale_factors(), th=th, nn_ratio=0.6)
            for b in range(3):
                rm, rn = O.ref_lsd_search_by_projection(lines, ml, b, synth.scale_factors(), th, 0.6)
                assert rn == nm[b]
                np.testing.assert_array_equal(m[b, :int(lines["n"][b])], rm)
        assert nm.sum() > 5
    rng = np.random.default_rng(111)
    kf = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    cur = synth._flip_bits(rng, kf[rng.permutation(40)[:35]], 60)
    has = (rng.random(40) < 0.8).astype(np.uint8)
    m, n = O.lsd_search_by_descriptor(kf, cur, has)
    rm, rn = O.ref_lsd_search_by_descriptor(kf, cur, has)
    assert n == rn > 5
    np.testing.assert_array_equal(m, rm)


def test_oracle_matches_committed_real_reference_fixtures():
    """Runs everywhere (the GPU box has no reference tree): fixtures were produced by oracle/_ref/ref_match."""
    g = np.load(os.path.join(GOLD, "guided_ref.npz"))
    fr = synth.guided_frame(B=2, N=800, seed=int(g["seed"]))
    cur, last = synth.guided_last_frame(fr, seed=int(g["seed"]) + 1, dup=0.3)
    m, nm = O.search_by_projection_frame(cur, last, 15.0)
    np.testing.assert_array_equal(m, g["proj_frame_match"]); np.testing.assert_array_equal(nm, g["proj_frame_n"])
    fr2, pr = synth.guided_map_probes(fr, seed=int(g["seed"]) + 2, n_probes=2000)
    m, nm = O.search_by_projection_map(fr2, pr, th=3.0, nn_ratio=0.8)
    np.testing.assert_array_equal(m, g["proj_map_match"]); np.testing.assert_array_equal(nm, g["proj_map_n"])
    kf, f = synth.guided_bow(B=2, N=800, seed=int(g["seed"]) + 3)
    m, nm = O.search_by_bow(kf, f, nn_ratio=0.7)
    np.testing.assert_array_equal(m, g["bow_match"]); np.testing.assert_array_equal(nm, g["bow_n"])
    frp, mp = synth.guided_planes(B=4, seed=int(g["seed"]) + 4)
    a, v, p, n = O.plane_search_by_coefficients(frp, mp)
    np.testing.assert_array_equal(np.stack([a, v, p]), g["plane_avp"]); np.testing.assert_array_equal(n, g["plane_n"])
    lines, ml = synth.guided_lines(B=3, n_lines=150, n_ml=400, seed=int(g["seed"]) + 5)
    m, nm = O.lsd_search_by_projection(lines, ml, synth.scale_factors(), th=3.0, nn_ratio=0.6)
    np.testing.assert_array_equal(m, g["lsd_proj_match"]); np.testing.assert_array_equal(nm, g["lsd_proj_n"])
