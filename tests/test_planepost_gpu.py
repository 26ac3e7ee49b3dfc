"""GPU parity: plane post-processing (planepost.hip: Frame::ComputePlanes head + Frame::MaxPointDistanceFromPlane, Map::FlagMatchedPlanePoints, the cloud merge of
MapPlane::UpdateCoefficientsAndPoints) against oracle/planepost_oracle.cpp (PCL restated: PARITY UNPINNED, PCL is not in this image).

Tolerances: every integer decision (voxel membership and order, kept planes, RANSAC iterations / samples / inlier counts, sampler draws) must be IDENTICAL.
Voxel centroids: BIT-EXACT - PCL sums a voxel's points as floats in the order std::sort (unstable introsort, key = voxel index only) leaves them, and the kernel
reproduces that order (isort.h) and that float chain.  Refit coefficients: 1e-6 against the oracle's own chain (float chains in the same order on identical clouds;
sqrt and division are correctly rounded on both sides, the eigen solver's sin / cos / atan2 may differ in the last bit)."""
import numpy as np
import pytest

import oracle_lib as ol
import planepost_cases as pc
from planarslam_amd.synth import depth_image

pytestmark = pytest.mark.gpu
INT_KEYS = ("iterations", "best_count", "s0", "s1", "s2", "n_inliers", "n_inliers_refined", "draws")


def _same_ints(a, b, ctx):
    for k in INT_KEYS:
        assert a[k] == b[k], (ctx, k, a, b)


@pytest.mark.parametrize("dist_th", [0.05, 0.02])
def test_refit_matches_oracle_on_identical_clouds(dist_th):
    from planarslam_amd import PlaneClouds
    cases = [c for c in pc.refit_cases()]
    pcz = PlaneClouds(640, 480)
    state, planes, info = pcz.refit([c["plane"] for c in cases], [c["pts"] for c in cases], dist_th=dist_th)
    for i, c in enumerate(cases):
        st, pl, inf = ol.plane_refit(c["plane"], c["pts"], dist_th)
        assert state[i] == st, (c["name"], state[i], st)
        if st == 1:
            continue
        _same_ints(info[i], inf, c["name"])
        assert np.array_equal(info[i]["model"].view(np.int32), inf["model"].view(np.int32)) or (np.isnan(inf["model"]).all() and np.isnan(info[i]["model"]).all()), c["name"]
        if st == 0:
            assert np.abs(planes[i] - pl).max() < 1e-6, (c["name"], planes[i], pl)


def test_refit_many_iterations_exact():
    """Clouds whose random 3-point models are poor: RANSAC runs tens of iterations; every decision along the way must agree."""
    from planarslam_amd import PlaneClouds
    cl = [pc.plane_cloud(300 + i, n=900 + 37 * i, th=0.012, spread=0.97, extent=2.0 + 0.5 * i) for i in range(12)]
    state, planes, info = PlaneClouds(640, 480).refit([c[0] for c in cl], [c[1] for c in cl], dist_th=0.012)
    its = []
    for i, (plane, pts) in enumerate(cl):
        st, pl, inf = ol.plane_refit(plane, pts, 0.012)
        assert state[i] == st == 0
        _same_ints(info[i], inf, i)
        assert np.abs(planes[i] - pl).max() < 1e-6
        its.append(inf["iterations"])
    assert max(its) >= 8 and sum(its) >= 50, its


def _gpu_frames(depths, dist_th=0.05, max_points=4096, debug=True, retry_per_plane=True):
    from planarslam_amd import PlaneClouds, PlaneDetection
    B = len(depths)
    det = PlaneDetection(640, 480, max_batch=B)
    res = det.run(depths)
    planes = np.zeros((B, det.max_planes, 8)); labels = np.zeros((B, 480, 640), np.int32); n = np.zeros(B, np.int32)
    for b, (p, l) in enumerate(res):
        planes[b, :len(p)] = p; labels[b] = l; n[b] = len(p)
    return res, PlaneClouds(640, 480, max_batch=B, max_points=max_points).compute(depths, labels, planes, n, dist_th=dist_th, debug=debug, retry_per_plane=retry_per_plane)


@pytest.mark.parametrize("dist_th", [0.05, 0.03])
def test_plane_clouds_match_oracle(dist_th):
    depths = np.stack([depth_image(50 + i, noise=(i % 2 == 0), holes=(i % 3 != 0)) for i in range(6)])
    res, got = _gpu_frames(depths, dist_th)
    kept, chain_gap = 0, 0.0
    for b in range(len(depths)):
        planes, labels = res[b]
        want = ol.plane_clouds(depths[b], labels, planes, dis_th=dist_th)
        g = got[b]
        assert np.array_equal(g["nvox"], want["nvox"]), b
        # centroids of EVERY detector plane would need the dropped ones too; the kept ones are compared below, the gate decision of all here
        same_gate = (g["state"] == 1) == (want["state"] == 1)
        assert same_gate.all(), (b, g["state"], want["state"])
        assert np.array_equal(g["state"], want["state"]), (b, g["state"], want["state"])
        assert g["n"] == want["n"] and np.array_equal(g["src"], want["src"]) and np.array_equal(g["pt_off"], want["pt_off"])
        assert np.array_equal(g["points"], want["points"]), (b, np.abs(g["points"] - want["points"]).max())
        for k, p in enumerate(g["src"]):
            cloud = g["points"][g["pt_off"][k]:g["pt_off"][k + 1]]
            P = planes[p]
            c0 = np.array([P[1], P[2], P[3], -(P[1] * P[4] + P[2] * P[5] + P[3] * P[6])]).astype(np.float32)
            st, pl, inf = ol.plane_refit(c0, cloud, dist_th)
            assert st == 0
            _same_ints(g["info"][p], inf, (b, p))
            _same_ints(g["info"][p], want["info"][p], (b, p))
            # the refitted coefficient against the oracle's own plane chain (its own cloud: the same floats now)
            assert np.abs(g["coef"][k] - pl).max() < 1e-6, (b, p, g["coef"][k], pl)
            chain_gap = max(chain_gap, float(np.abs(g["coef"][k] - want["coef"][k]).max()))
            # sanity of the oracle itself: PCL's float sums stay within their summation error of the exact (double) centroid
            ys, xs = np.nonzero(labels == p)
            z = depths[b][ys, xs].astype(np.float64) * np.float64(np.float32(1.0 / 5000.0))
            pts = np.stack([(xs - np.float64(np.float32(320.1))) * z / np.float64(np.float32(535.4)), (ys - np.float64(np.float32(247.6))) * z / np.float64(np.float32(539.2)), z], 1).astype(np.float32)
            _, exact, _ = ol.voxel_grid(pts, want_exact=True)
            assert np.abs(cloud - exact).max() < 2e-5, (b, p, np.abs(cloud - exact).max())
            kept += 1
    assert kept >= 10
    assert chain_gap <= 1e-6, f"largest coefficient gap between the device's and the oracle's plane chain: {chain_gap:.2e}"


def test_plane_clouds_batch_reuse_and_empty():
    """More frames than one dispatch round keeps resident, a second call on the same handle (workspace reuse), frames without planes."""
    from planarslam_amd import PlaneClouds, PlaneDetection
    src = np.stack([depth_image(900 + i) for i in range(3)])
    B = 40
    depths = src[np.arange(B) % 3].copy()
    depths[7] = 0                                                    # no plane at all
    det = PlaneDetection(640, 480, max_batch=B)
    res = det.run(depths)
    planes = np.zeros((B, det.max_planes, 8)); labels = np.zeros((B, 480, 640), np.int32); n = np.zeros(B, np.int32)
    for b, (p, l) in enumerate(res):
        planes[b, :len(p)] = p; labels[b] = l; n[b] = len(p)
    pcz = PlaneClouds(640, 480, max_batch=B)
    a = pcz.compute(depths, labels, planes, n)
    b2 = pcz.compute(depths, labels, planes, n)
    assert a[7]["n"] == 0 and len(a[7]["points"]) == 0
    for i in range(B):
        for k in ("coef", "src", "pt_off", "points"):
            assert np.array_equal(a[i][k], b2[i][k]) and np.array_equal(a[i][k], a[i % 3 if i != 7 else 7][k]), (i, k)
    assert a[0]["n"] >= 2


def test_plane_clouds_capacity_is_reported():
    from planarslam_amd import PlanarError
    depths = depth_image(50)[None]
    with pytest.raises(PlanarError):
        _gpu_frames(depths, max_points=64, debug=False, retry_per_plane=False)
    # with the plane-by-plane pass a plane that alone overflows the table is dropped and listed (64 voxels hold 0.6 m^2: nearly every plane here); the others are
    # what the oracle delivers for them
    res, got = _gpu_frames(depths, max_points=64, debug=False)
    planes, labels = res[0]
    want = ol.plane_clouds(depths[0], labels, planes)
    g = got[0]
    assert len(g["dropped"]) >= 1 and all(int(want["nvox"][i]) > 64 for i in g["dropped"])
    keep = [k for k in range(want["n"]) if int(want["src"][k]) not in g["dropped"]]
    assert g["n"] == len(keep) and np.array_equal(g["src"], want["src"][keep])
    for j, k in enumerate(keep):
        assert np.array_equal(g["points"][g["pt_off"][j]:g["pt_off"][j + 1]], want["points"][want["pt_off"][k]:want["pt_off"][k + 1]])


def test_flag_matched_plane_points_matches_oracle():
    from planarslam_amd import flag_matched_plane_points
    B = 3
    Tcw = np.stack([pc.pose(10 + b).astype(np.float32).reshape(16) for b in range(B)])
    rng = np.random.default_rng(3)
    coef = np.zeros((B, 8, 4), np.float32); matched = np.zeros((B, 8), np.uint8); n = np.array([5, 0, 8], np.int32)
    for b in range(B):
        for i in range(8):
            v = rng.normal(size=3); v /= np.linalg.norm(v)
            coef[b, i] = [*v, rng.uniform(-2, 2)]
            matched[b, i] = rng.uniform() < 0.6
    xw = pc.world_points(11, n=5000)
    flags, nm = flag_matched_plane_points(Tcw, coef, matched, n, xw)
    for b in range(B):
        f, m = ol.flag_matched_plane_points(Tcw[b], coef[b, :n[b]], matched[b, :n[b]], xw)
        assert np.array_equal(flags[b], f) and nm[b] == m, b
    assert flags[0].sum() > 100 and flags[1].sum() == 0
    # per-frame point arrays
    xwb = np.stack([pc.world_points(20 + b, n=700) for b in range(B)])
    flags, nm = flag_matched_plane_points(Tcw, coef, matched, n, xwb)
    for b in range(B):
        f, m = ol.flag_matched_plane_points(Tcw[b], coef[b, :n[b]], matched[b, :n[b]], xwb[b])
        assert np.array_equal(flags[b], f) and nm[b] == m, b


def test_merge_plane_points_matches_oracle():
    from planarslam_amd import PlaneClouds
    pcz = PlaneClouds(640, 480)
    for seed in (6, 8):
        T = np.linalg.inv(pc.pose(seed))
        _, f = pc.plane_cloud(seed + 1, n=900)
        _, m = pc.plane_cloud(seed + 2, n=600)
        got = pcz.merge(T, f, m)
        want = ol.merge_plane_points(T, f, m)
        assert len(got) == len(want) and np.array_equal(got, want), np.abs(got - want).max()
    assert len(pcz.merge(np.eye(4), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))) == 0


def test_plane_clouds_8192_voxels_same_result():
    """The largest table (128 KB of LDS for the keys, 64-entry tile tables) gives what the default gives."""
    depths = np.stack([depth_image(60 + i) for i in range(2)])
    _, a = _gpu_frames(depths, max_points=4096, debug=False)
    _, b = _gpu_frames(depths, max_points=8192, debug=False)
    for x, y in zip(a, b):
        for k in ("coef", "src", "pt_off", "points"):
            assert np.array_equal(x[k], y[k]), k


def test_plane_clouds_other_size_and_row_pitch():
    """320 x 240 (5 column strips, 4 row tiles) against the oracle, and a padded depth buffer (pitch > width) through the ABI directly."""
    import ctypes as C
    from planarslam_amd import PlaneClouds, PlaneDetection
    from planarslam_amd._lib import check, lib
    W, H = 320, 240
    cam = (267.7, 269.6, 160.05, 123.8)
    d = depth_image(77, W, H)
    planes, labels = PlaneDetection(W, H).run(d, K=cam)
    op, olab = ol.peac_run(d, *cam)
    assert np.array_equal(labels, olab) and len(planes) >= 1
    pl = np.zeros((1, 128, 8)); pl[0, :len(planes)] = planes
    npl = np.array([len(planes)], np.int32)
    pcz = PlaneClouds(W, H)
    got = pcz.compute(d[None], labels[None], pl, npl, K=cam, debug=True)[0]
    want = ol.plane_clouds(d, labels, planes, cam=cam)
    assert got["n"] == want["n"] and np.array_equal(got["state"], want["state"]) and np.array_equal(got["pt_off"], want["pt_off"]) and np.array_equal(got["nvox"], want["nvox"])
    assert np.array_equal(got["points"], want["points"])
    # the same frame with 24 bytes of padding per depth row
    pitch = W + 12
    dp = np.zeros((H, pitch), np.uint16); dp[:, :W] = d; dp[:, W:] = 12345
    PS, MP = pcz.pl_stride, pcz.max_points
    n = np.zeros(1, np.int32); coef = np.zeros((1, PS, 4), np.float32); src = np.zeros((1, PS), np.int32); off = np.zeros((1, PS + 1), np.int32); pts = np.zeros((1, MP, 3), np.float32)
    lab = np.ascontiguousarray(labels[None], np.int32)
    check(lib().planar_plane_clouds_compute(pcz.h, dp.ctypes.data, 1, pitch, pitch * H, cam[0], cam[1], cam[2], cam[3], np.float32(1.0 / 5000.0), lab.ctypes.data, pl.ctypes.data,
                                            npl.ctypes.data, 0.05, np.float32(0.1), n.ctypes.data, coef.ctypes.data, src.ctypes.data, off.ctypes.data, pts.ctypes.data, None, None, None))
    k = got["n"]
    assert n[0] == k and np.array_equal(off[0, :k + 1], got["pt_off"]) and np.array_equal(coef[0, :k], got["coef"]) and np.array_equal(pts[0, :off[0, k]], got["points"])


def test_plane_clouds_of_a_1280x720_frame():
    """Round 6: frames of more than 2^19 pixels.  A sort word carries 20 bits of pixel and 12 of voxel there (max_points <= 4096), and a plane of more than ~243 000 pixels
    does not fit the sort's LDS stop bitmaps (isort::wg_partition_long).  PlaneDetection's labels and planes, the voxel centroids (bit-exact: PCL's float sums in std::sort's
    order) and the refit against the oracle; 8192 voxels are refused for such a frame."""
    from planarslam_amd import PlaneClouds, PlaneDetection
    from planarslam_amd._lib import PlanarError
    W, H = 1280, 720
    cam = (1070.8, 1078.4, 640.2, 495.2)
    with pytest.raises(PlanarError):
        PlaneClouds(W, H, max_points=8192)
    pcz = PlaneClouds(W, H, max_batch=2, max_points=4096)
    ds = np.stack([depth_image(1587 + 11 * i, W, H, noise=(i == 0), holes=True) for i in range(2)])
    res = PlaneDetection(W, H, max_batch=2).run(ds, K=cam)
    pl = np.zeros((2, pcz.pl_stride, 8)); npl = np.zeros(2, np.int32)
    for b in range(2):
        op, olab = ol.peac_run(ds[b], *cam)
        assert np.array_equal(res[b][1], olab) and np.array_equal(res[b][0], op)
        pl[b, :len(op)] = op; npl[b] = len(op)
    assert max(int((res[0][1] == q).sum()) for q in range(npl[0])) > 300000          # a plane longer than the stop bitmaps hold
    got = pcz.compute(ds, np.stack([r[1] for r in res]), pl, npl, K=cam, debug=True)
    for b in range(2):
        want = ol.plane_clouds(ds[b], res[b][1], res[b][0], cam=cam)
        g = got[b]
        assert g["n"] == want["n"] and np.array_equal(g["state"], want["state"]) and np.array_equal(g["pt_off"], want["pt_off"]) and np.array_equal(g["nvox"], want["nvox"])
        assert np.array_equal(g["points"], want["points"]), "voxel centroids differ from the oracle (PCL's summation order)"
        assert np.abs(g["coef"] - want["coef"]).max(initial=0) <= 1e-6


def far_room_depth():
    """A room seen from far: floor, ceiling, back wall and two side walls at 5-10 m, noise-free.  Its planes together hold about 16 000 voxels of 0.1 m - twice what
    the frame's voxel table holds (max_points <= 8192) - but none of them more than the table.  pcl::VoxelGrid has no cap (reference src/Frame.cc:674-679)."""
    import torch
    from planarslam_amd import synth_se3
    from planarslam_amd.synth import TUM3, gray_image
    X, Y, Z = np.eye(3)
    inf = np.inf
    F = [(-Y, 1.45, X, Z), (Y, 2.25, X, Z), (-Z, 9.95, X, Y), (Z, 1.0, X, Y), (X, 3.95, Z, Y), (-X, 3.95, Z, Y)]
    scene = dict(n=np.array([f[0] for f in F], np.float64), d=np.array([f[1] for f in F], np.float64), a=np.array([f[2] for f in F], np.float64),
                 b=np.array([f[3] for f in F], np.float64), lo=np.full((6, 2), -inf), hi=np.full((6, 2), inf), toff=np.zeros((6, 2)), gain=np.ones(6))
    tex = torch.from_numpy(gray_image(77, 736, 576)[None])
    T = np.eye(4)[None]
    _, d = synth_se3.render(torch, [scene], tex, T, TUM3, depth_noise=False, holes=False, pixel_noise=0)
    return d[0].numpy().view(np.uint16)


def test_frame_with_more_voxels_than_the_table_goes_plane_by_plane():
    """a frame whose planes together exceed the voxel table used to lose ALL its planes in the adapter.  Now the one-pass call reports the
    overflow (status 3) and the frame goes through once more plane by plane (planar_plane_clouds_set_plane_window; include/planar_adapters.hpp and
    PlaneClouds.compute do this): every plane's cloud and refit are the oracle's, in plane order - bit-exact centroids, coefficients to 1e-6."""
    from planarslam_amd import PlaneClouds, PlaneDetection
    from planarslam_amd._lib import PlanarError
    d = far_room_depth()
    planes, labels = PlaneDetection(640, 480, max_batch=1).run(d)
    oplanes, olabels = ol.peac_run(d)
    assert np.array_equal(labels, olabels) and np.array_equal(planes, oplanes) and len(planes) >= 4
    want = ol.plane_clouds(d, olabels, oplanes)
    per_plane = np.diff(want["pt_off"])
    assert want["pt_off"][-1] > 8192 and per_plane.max() <= 8192, (int(want["pt_off"][-1]), per_plane.tolist())      # the premise: too many together, none alone
    pcz = PlaneClouds(640, 480, max_points=8192)
    pl = np.zeros((1, pcz.pl_stride, 8)); pl[0, :len(planes)] = planes
    npl = np.array([len(planes)], np.int32)
    with pytest.raises(PlanarError) as e:                                      # the one-pass call says so ...
        pcz.compute(d[None], labels[None], pl, npl, retry_per_plane=False)
    assert e.value.code == -4 and "(code 3" in str(e.value)
    got = pcz.compute(d[None], labels[None], pl, npl)[0]                       # ... and the plane-by-plane pass delivers what pcl::VoxelGrid delivers
    assert got["dropped"] == [] and got["n"] == want["n"] and np.array_equal(got["src"], want["src"]) and np.array_equal(got["pt_off"], want["pt_off"])
    assert np.array_equal(got["points"], want["points"])
    assert np.abs(got["coef"] - want["coef"]).max() <= 1e-6
    # the window is reset: an ordinary frame afterwards goes through in one pass
    d2 = depth_image(4321)
    p2, l2 = PlaneDetection(640, 480, max_batch=1).run(d2)
    pl2 = np.zeros((1, pcz.pl_stride, 8)); pl2[0, :len(p2)] = p2
    g2 = pcz.compute(d2[None], l2[None], pl2, np.array([len(p2)], np.int32))[0]
    w2 = ol.plane_clouds(d2, l2, p2)
    assert g2["dropped"] == [] and g2["n"] == w2["n"] and np.array_equal(g2["points"], w2["points"])
    # a batch with ONE overflowing frame: only that frame is redone plane by plane (per-frame codes: planar_plane_clouds_last_status), both results carry the same keys
    pcz2 = PlaneClouds(640, 480, max_batch=2, max_points=pcz.max_points)
    plb = np.zeros((2, pcz2.pl_stride, 8)); plb[0] = pl[0]; plb[1] = pl2[0]
    both = pcz2.compute(np.stack([d, d2]), np.stack([labels, l2]), plb, np.array([len(planes), len(p2)], np.int32), debug=True)
    assert set(both[0]) == set(both[1]) and both[0]["n"] == want["n"] and np.array_equal(both[0]["points"], want["points"])
    assert both[1]["n"] == w2["n"] and np.array_equal(both[1]["points"], w2["points"]) and both[1]["dropped"] == []
