"""Oracle vs OpenCV itself - ONLY when a maintainer has produced tests/golden/opencv_golden.npz with tools/dump_opencv_golden.py on a machine
that has OpenCV 3.4.x + contrib (neither this container nor the GPU box has it).  Absent file = skipped, and the oracle headers keep
saying "parity unpinned below OpenCV" for resize / blur / FAST / LSD / LBD."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from planarslam_amd import synth

PATH = os.path.join(os.path.dirname(__file__), "golden", "opencv_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/opencv_golden.npz not generated (needs OpenCV 3.4.x + contrib, tools/dump_opencv_golden.py)")


@pytest.fixture(scope="module")
def gold():
    return np.load(PATH)


def test_resize_blur_fast(gold):
    L = O.lib()
    for s in gold["seeds"]:
        img = synth.gray_image(int(s))
        lv = img
        for l in range(1, 8):
            want = gold[f"resize/{s}/{l}"]
            out = np.zeros_like(want)
            L.orc_resize_linear_u8(lv.ctypes.data, lv.shape[1], lv.shape[0], lv.shape[1], out.ctypes.data, want.shape[1], want.shape[0], want.shape[1])
            np.testing.assert_array_equal(out, want, err_msg=f"resize level {l}")
            lv = out
        out = np.zeros_like(img)
        L.orc_gaussian7_s2_u8(img.ctypes.data, img.shape[1], img.shape[0], img.shape[1], out.ctypes.data, img.shape[1])
        np.testing.assert_array_equal(out, gold[f"blur/{s}"])
        cell = np.ascontiguousarray(img[100:160, 200:260])
        for th in (20, 7):
            buf = np.zeros((4096, 3), np.int32)
            n = L.orc_fast9_16(cell.ctypes.data, 60, 60, 60, th, 1, buf.ctypes.data, 4096)
            np.testing.assert_array_equal(buf[:n], gold[f"fast/{s}/{th}"])


def test_lsd_and_lbd(gold):
    for s in gold["seeds"]:
        img = synth.gray_image(int(s))
        if f"lsd/{s}/xy" in gold:
            got = O.lsd_detect(img, tie_order=0)
            np.testing.assert_array_equal(got["xy"], gold[f"lsd/{s}/xy"])
        if f"lbd/{s}/desc" in gold:
            kl, desc, _, _, _ = O.extract_line_segment(img, tie_order=0)
            np.testing.assert_array_equal(desc, gold[f"lbd/{s}/desc"])
            want = gold[f"lbd/{s}/keylines"]
            np.testing.assert_array_equal(kl["pt_x"], want[:, 3].astype(np.float32))
            np.testing.assert_array_equal(kl["response"], want[:, 5].astype(np.float32))


def test_undistort_points(gold):
    import frame_cases as fc
    if "undistort/TUM1" not in gold.files:
        pytest.skip("opencv_golden.npz predates the undistortPoints vectors: re-run tools/dump_opencv_golden.py")
    keys, _ = fc.undistort_case()
    for name, (K, D) in fc.DIST.items():
        un = O.undistort_keypoints(keys[0], dict(fx=K[0], fy=K[1], cx=K[2], cy=K[3]), D)
        if D[0] == 0:
            continue                      # the reference does not call the library then (Frame.cc:546-549)
        assert np.array_equal(np.stack([un["x"], un["y"]], 1), gold[f"undistort/{name}"]), name
