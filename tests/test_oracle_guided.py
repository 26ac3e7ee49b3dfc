"""CPU: the guided-matcher oracle (oracle/guided_oracle.cpp) against independent pure-numpy restatements of the
reference loops on small cases, plus behavioural properties the reference guarantees."""
import numpy as np

import oracle_lib as O
from planarslam_amd import synth


def _ham(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def _features_in_area(fr, b, x, y, r, minL, maxL):
    """Frame::GetFeaturesInArea (reference src/Frame.cc:440-489), float32 arithmetic, python loops."""
    f32 = np.float32
    k = fr["keys_un"][b]; n = int(fr["n"][b])
    winv = f32(64) / f32(f32(fr["max_x"]) - f32(fr["min_x"])); hinv = f32(48) / f32(f32(fr["max_y"]) - f32(fr["min_y"]))
    cells = {}
    for i in range(n):
        px = int(np.round(f32(f32(k["x"][i] - f32(fr["min_x"])) * winv))); py = int(np.round(f32(f32(k["y"][i] - f32(fr["min_y"])) * hinv)))
        # np.round is half-to-even; the reference's round() is half-away: equal except on exact .5 (not generated)
        if 0 <= px < 64 and 0 <= py < 48:
            cells.setdefault((px, py), []).append(i)
    x, y, r = f32(x), f32(y), f32(r)
    x0 = max(0, int(np.floor(f32(f32(f32(x - f32(fr["min_x"])) - r) * winv))))
    x1 = min(63, int(np.ceil(f32(f32(f32(x - f32(fr["min_x"])) + r) * winv))))
    y0 = max(0, int(np.floor(f32(f32(f32(y - f32(fr["min_y"])) - r) * hinv))))
    y1 = min(47, int(np.ceil(f32(f32(f32(y - f32(fr["min_y"])) + r) * hinv))))
    if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
        return []
    chk = minL > 0 or maxL >= 0
    out = []
    for ix in range(x0, x1 + 1):
        for iy in range(y0, y1 + 1):
            for i in cells.get((ix, iy), []):
                if chk and (k["octave"][i] < minL or (maxL >= 0 and k["octave"][i] > maxL)):
                    continue
                if abs(f32(k["x"][i] - x)) < r and abs(f32(k["y"][i] - y)) < r:
                    out.append(i)
    return out


def _py_search_map(fr, pr, b, th, ratio):
    f32 = np.float32
    n = int(fr["n"][b]); blocked = fr["blocked"][b].copy().astype(bool)
    match = np.full(fr["keys_un"].shape[1], -1, np.int32); nm = 0
    for i in range(int(pr["n"][b])):
        if not pr["in_view"][b, i]:
            continue
        lvl = int(pr["level"][b, i])
        r = f32(2.5) if float(pr["view_cos"][b, i]) > 0.998 else f32(4.0)
        if th != 1.0:
            r = f32(r * f32(th))
        rr = f32(r * fr["scale_factors"][lvl])
        best = (256, -1, -1); second = (256, -1)
        for idx in _features_in_area(fr, b, pr["proj_x"][b, i], pr["proj_y"][b, i], rr, lvl - 1, lvl):
            if blocked[idx]:
                continue
            if fr["u_right"][b, idx] > 0 and abs(f32(pr["proj_xr"][b, i] - fr["u_right"][b, idx])) > rr:
                continue
            d = _ham(pr["desc"][b, i], fr["desc"][b, idx])
            if d < best[0]:
                second = (best[0], best[2]); best = (d, idx, int(fr["keys_un"]["octave"][b, idx]))
            elif d < second[0]:
                second = (d, int(fr["keys_un"]["octave"][b, idx]))
        if best[0] <= 100:
            if best[2] == second[1] and f32(best[0]) > f32(f32(ratio) * f32(second[0])):
                continue
            match[best[1]] = i; blocked[best[1]] = bool(pr["observed"][b, i]); nm += 1
    return match, nm


def test_search_by_projection_map_matches_python_restatement():
    fr = synth.guided_frame(B=2, N=300, seed=21)
    fr, pr = synth.guided_map_probes(fr, seed=22, n_probes=400)
    for th in (1.0, 3.0):
        m, nm = O.search_by_projection_map(fr, pr, th=th, nn_ratio=0.8)
        for b in range(2):
            pm, pn = _py_search_map(fr, pr, b, th, 0.8)
            assert pn == nm[b] and pn > 20
            np.testing.assert_array_equal(pm, m[b])


def test_search_by_projection_frame_properties():
    fr = synth.guided_frame(B=3, N=600, seed=31)
    for motion in ((0, 0, 0), (0, 0, 0.3), (0, 0, -0.3)):   # neither / backward / forward octave windows
        cur, last = synth.guided_last_frame(fr, seed=32, motion=motion)
        m, nm = O.search_by_projection_frame(cur, last, th=15.0)
        m_no, nm_no = O.search_by_projection_frame(cur, last, th=15.0, check_orientation=False)
        for b in range(3):
            assert nm[b] > 100 and nm_no[b] >= nm[b]
            a = m[b][m[b] >= 0]
            # every assigned index is a usable last-frame point and within the Hamming threshold
            assert last["usable"][b, a].all()
            for i2 in np.nonzero(m[b] >= 0)[0][:50]:
                assert _ham(cur["desc"][b, i2], last["mp_desc"][b, m[b, i2]]) <= 100
            # the rotation filter only removes
            assert ((m[b] == m_no[b]) | (m[b] == -1)).all()
    # initially blocked keypoints are never assigned
    assert (m[cur["blocked"] > 0] == -1).all()


def test_search_by_bow_properties():
    kf, f = synth.guided_bow(B=2, N=500, seed=41)
    m, nm = O.search_by_bow(kf, f, nn_ratio=0.7)
    m2, nm2 = O.search_by_bow(kf, f, nn_ratio=0.7, check_orientation=False)
    for b in range(2):
        assert nm[b] > 50 and nm2[b] >= nm[b]
        idx = np.nonzero(m[b] >= 0)[0]
        assert (f["node"][b, idx] == kf["node"][b, m[b, idx]]).all()       # same vocabulary node
        assert kf["usable"][b, m[b, idx]].all()
        for i in idx[:60]:
            assert _ham(f["desc"][b, i], kf["desc"][b, m[b, i]]) <= 50
        assert len(np.unique(m2[b][m2[b] >= 0])) <= (m2[b] >= 0).sum()
        assert (m[b][f["n"][b]:] == -1).all()


def test_lsd_search_by_projection_and_plane_matcher_small():
    lines, ml = synth.guided_lines(B=3, seed=51)
    m, nm = O.lsd_search_by_projection(lines, ml, synth.scale_factors(), th=1.0, nn_ratio=0.6)
    m3, nm3 = O.lsd_search_by_projection(lines, ml, synth.scale_factors(), th=3.0, nn_ratio=0.6)
    assert nm3.sum() >= nm.sum() and nm3.sum() > 5
    for b in range(3):
        assert (m3[b][lines["blocked"][b] > 0] == -1).all()
        assert ((m3[b] >= 0).sum()) <= nm3[b]     # re-assignment of an unobserved line may overwrite
    fr, mp = synth.guided_planes(B=4, seed=52)
    a, v, p, n = O.plane_search_by_coefficients(fr, mp)
    assert n.sum() >= 8
    for b in range(4):
        for i in range(int(fr["n"][b])):
            if a[b, i] >= 0:
                pw = fr["Tcw"][b].reshape(4, 4).T @ fr["coef"][b, i]
                cosang = float(pw[:3] @ mp["coef"][b, a[b, i], :3])
                assert abs(cosang) > 0.86 and mp["valid"][b, a[b, i]]
            if v[b, i] >= 0:
                pw = fr["Tcw"][b].reshape(4, 4).T @ fr["coef"][b, i]
                assert abs(float(pw[:3] @ mp["coef"][b, v[b, i], :3])) < 0.08716 + 1e-6
    fr2, mp2 = synth.guided_planes(B=3, seed=53, shared=True)
    a2, _, _, n2 = O.plane_search_by_coefficients(fr2, mp2)
    assert n2.sum() >= 3


def test_is_in_frustum_points_and_lines_properties():
    fr = synth.guided_frame(B=2, N=300, seed=61)
    fr, mp, ml = synth.guided_local_map(fr, seed=62)
    lsf = float(np.float32(np.log(np.float32(1.2))))
    o = O.is_in_frustum_points(fr, mp, lsf, 8)
    iv = o["in_view"] > 0
    assert 0.1 < iv[mp["valid"] > 0].mean() < 0.9 and not iv[mp["valid"] == 0].any()
    assert (o["proj_x"][iv] >= 0).all() and (o["proj_x"][iv] <= 640).all() and (o["proj_y"][iv] >= 0).all() and (o["proj_y"][iv] <= 480).all()
    assert (o["view_cos"][iv] >= 0.5).all() and (o["level"][iv] >= 0).all() and (o["level"][iv] <= 7).all()
    # independent float64 projection agrees to float32 accuracy
    for b in range(2):
        T = fr["Tcw"][b].reshape(4, 4).astype(np.float64)
        j = np.nonzero(iv[b])[0][:200]
        Xc = (T[:3, :3] @ mp["xw"][b, j].T.astype(np.float64)).T + T[:3, 3]
        np.testing.assert_allclose(o["proj_x"][b, j], fr["fx"] * Xc[:, 0] / Xc[:, 2] + fr["cx"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(o["proj_xr"][b, j], o["proj_x"][b, j] - fr["bf"] / Xc[:, 2], rtol=0, atol=2e-3)
    ol = O.is_in_frustum_lines(fr, ml, lsf)
    il = ol["in_view"] > 0
    assert 0.05 < il[ml["valid"] > 0].mean() < 0.9
    assert (ol["proj"][il] >= 0).all() and (ol["proj"][il][:, [0, 2]] <= 640).all()


def test_undistort_keypoints_inverts_the_distortion_model():
    """The oracle of Frame::UndistortKeyPoints (cv::undistortPoints' five fixed-point iterations; unpinned - OpenCV is not in this image): pushing its
    output through the forward Brown-Conrady model comes back to the key point - to the few hundredths of a pixel five iterations reach inside the image, and
    coarser in the far corners where the iteration has not converged (what the reference then uses as mnMinX .. mnMaxY)."""
    import frame_cases as fc
    keys, n = fc.undistort_case()
    for name, (K, D) in fc.DIST.items():
        cam = dict(fx=K[0], fy=K[1], cx=K[2], cy=K[3])
        un = O.undistort_keypoints(keys[0], cam, D)
        for f in ("size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(un[f], keys[0][f])
        if D[0] == 0:
            assert np.array_equal(un, keys[0])
            continue
        fx, fy, cx, cy = [np.float64(np.float32(v)) for v in K]
        k1, k2, p1, p2, k3 = [np.float64(np.float32(v)) for v in D]
        x = (un["x"].astype(np.float64) - cx) / fx; y = (un["y"].astype(np.float64) - cy) / fy
        r2 = x * x + y * y
        rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        err = np.hypot(xd * fx + cx - keys[0]["x"], yd * fy + cy - keys[0]["y"])
        r_px = np.hypot(keys[0]["x"] - cx, keys[0]["y"] - cy)
        assert err[r_px < 250].max() < 0.05, (name, err[r_px < 250].max())
        assert err.max() < 6.0, (name, err.max())
        assert np.abs(un["x"] - keys[0]["x"]).max() > 5                      # and it does move the corners by many pixels
