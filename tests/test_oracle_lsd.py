"""CPU: the line-extraction oracle (oracle/lsd_oracle.cpp).  No OpenCV exists here, so these are behavioural checks
(geometry of detected segments on synthetic edges, descriptor invariances), not a pin against the library."""
import numpy as np

import oracle_lib as O
from planarslam_amd import synth


def _rect_image():
    img = np.full((480, 640), 40, np.uint8)
    img[100:300, 150:450] = 200           # one bright rectangle -> 4 long edges
    rng = np.random.default_rng(3)
    return np.clip(img.astype(np.int32) + rng.integers(-2, 3, img.shape), 0, 255).astype(np.uint8)


def test_lsd_finds_rectangle_edges():
    out = O.lsd_detect(_rect_image(), tie_order=0)
    xy = out["xy"]
    length = np.hypot(xy[:, 0] - xy[:, 2], xy[:, 1] - xy[:, 3])
    long = xy[length > 150]
    assert len(long) == 4
    horiz = long[np.abs(long[:, 1] - long[:, 3]) < 2]
    vert = long[np.abs(long[:, 0] - long[:, 2]) < 2]
    assert len(horiz) == 2 and len(vert) == 2
    assert sorted(np.round(horiz[:, 1]).astype(int).tolist()) in ([100, 300], [99, 299], [100, 299], [99, 300])
    assert sorted(np.round(vert[:, 0]).astype(int).tolist()) in ([150, 450], [149, 449], [150, 449], [149, 450])
    assert (out["wpn"][:, 2] > 0).all()     # every reported segment passed the NFA test


def test_tie_order_only_changes_a_few_segments():
    img = synth.gray_image(1234)
    a = O.lsd_detect(img, tie_order=0)["xy"]; b = O.lsd_detect(img, tie_order=1)["xy"]
    sa = set(map(tuple, a.round(2).tolist())); sb = set(map(tuple, b.round(2).tolist()))
    assert len(a) > 100 and abs(len(a) - len(b)) <= 5
    assert len(sa & sb) >= 0.9 * len(sa)


def test_extract_line_segment_contract():
    img = synth.gray_image(77)
    kl, desc, eq, df, nd = O.extract_line_segment(img, tie_order=1)
    assert nd > 40 and len(kl) == 40
    assert (np.diff(kl["response"]) <= 0).all() and (kl["class_id"] == np.arange(40)).all() and (kl["octave"] == 0).all()
    np.testing.assert_allclose(kl["response"], kl["line_length"] / 640, rtol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(eq, axis=1), 1, atol=1e-12)
    # the line equation vanishes on both endpoints
    for i in range(40):
        assert abs(eq[i] @ [kl["start_x"][i], kl["start_y"][i], 1]) < 1e-9 and abs(eq[i] @ [kl["end_x"][i], kl["end_y"][i], 1]) < 1e-9
    np.testing.assert_allclose(np.linalg.norm(df, axis=1), 1, atol=1e-5)     # LBD float vector is unit length
    assert desc.any(axis=1).all()
    # few lines: no sort, class ids in detection order
    kl2, desc2, eq2, _, nd2 = O.extract_line_segment(_rect_image(), tie_order=1)
    assert nd2 == len(kl2) <= 40 and (kl2["class_id"] == np.arange(len(kl2))).all()
    # blank image: nothing
    kl3, *_ , nd3 = O.extract_line_segment(np.full((480, 640), 90, np.uint8))
    assert nd3 == 0 and len(kl3) == 0


def test_wrapper_vs_real_reference_LSDextractor():
    """LineSegment::ExtractLineSegment itself is reference code: src/LSDextractor.cpp compiled where it lies (oracle/_ref/ref_lsd) with
    LSDDetector / BinaryDescriptor forwarding to the restatement.  Pins the call sequence, the std::sort by response (ties included:
    the grid image below has hundreds of equal-length edges), the cut to 40, class ids and the line equations."""
    import os
    import pytest
    if not os.path.exists(O.ref_lsd_path()):
        pytest.skip("oracle/_ref/ref_lsd not built (reference tree absent)")
    rng = np.random.default_rng(23)
    grid = np.full((480, 640), 60, np.uint8)
    for gy in range(0, 480, 24):
        for gx in range(0, 640, 24):
            if rng.random() < 0.9:
                w, h = rng.integers(12, 20, 2)
                grid[gy + 2:gy + 2 + h, gx + 2:gx + 2 + w] = rng.integers(120, 255)
    rect = np.full((480, 640), 40, np.uint8); rect[100:300, 150:450] = 200
    for img in (synth.gray_image(1234), synth.gray_image(77), grid, rect, np.full((480, 640), 90, np.uint8)):
        for tie in (0, 1):
            kl, desc, eq, _, _ = O.extract_line_segment(img, tie_order=tie)
            rk, rd, re = O.run_ref_lsd(img, tie_order=tie)
            assert len(rk) == len(kl)
            assert rk.tobytes() == kl.tobytes()
            np.testing.assert_array_equal(rd, desc)
            np.testing.assert_array_equal(re, eq)
