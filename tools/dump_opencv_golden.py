#!/usr/bin/env python3
"""Emit golden vectors of the OpenCV primitives the oracle RESTATES, for a maintainer who has the library versions the reference links
(OpenCV 3.4.1 + opencv_contrib line_descriptor) - this container and the GPU box have neither.

    python tools/dump_opencv_golden.py            # needs cv2 (3.4.x with contrib); writes tests/golden/opencv_golden.npz

Inputs are regenerated from seeds by planarslam_amd.synth (no image files), so only OUTPUTS are stored.  tests/test_oracle_opencv_golden.py picks the
file up when it exists and compares oracle/cvprim.cpp (resize, GaussianBlur, FAST), oracle/lsd_oracle.cpp (LSD segments, keylines, LBD) against it;
that is what retires the "parity unpinned below OpenCV" caveat of ORB levels / blur / FAST and of LSD / LBD (a11, a12).
Nothing here is imported by the product or by the GPU tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from planarslam_amd import synth  # noqa: E402


def main():
    try:
        import cv2
    except ImportError:
        raise SystemExit("cv2 is not importable here; run this where OpenCV 3.4.x (+ contrib line_descriptor) is installed")
    out = {"cv_version": np.array(cv2.__version__)}
    seeds = (1234, 5, 9)
    out["seeds"] = np.array(seeds)
    for s in seeds:
        img = synth.gray_image(s)
        # cv::resize(level l-1 -> level l, INTER_LINEAR) as ORBextractor::ComputePyramid does (src/ORBextractor.cc:1107-1132): scale 1/1.2 per level
        lv, pyr = img, []
        for l in range(1, 8):
            sc = 1.0 / (1.2 ** l)
            sz = (int(round(img.shape[1] * sc)), int(round(img.shape[0] * sc)))
            lv = cv2.resize(lv, sz, interpolation=cv2.INTER_LINEAR)
            pyr.append(lv)
            out[f"resize/{s}/{l}"] = lv
        # GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) (src/ORBextractor.cc:1077)
        out[f"blur/{s}"] = cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
        # cv::FAST(cell, th, nonmaxSuppression = true) at both thresholds of the reference (iniThFAST 20, minThFAST 7)
        for th in (20, 7):
            fd = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
            kps = fd.detect(img[100:160, 200:260], None)
            out[f"fast/{s}/{th}"] = np.array([[int(k.pt[0]), int(k.pt[1]), int(k.response)] for k in kps], np.int32).reshape(-1, 3)
        # cv::createLineSegmentDetector(LSD_REFINE_ADV)->detect (what line_descriptor::LSDDetector runs per octave)
        try:
            lsd = cv2.createLineSegmentDetector(cv2.LSD_REFINE_ADV)
            lines, width, prec, nfa = lsd.detect(img)
            out[f"lsd/{s}/xy"] = np.asarray(lines, np.float32).reshape(-1, 4)
            out[f"lsd/{s}/wpn"] = np.stack([np.ravel(width), np.ravel(prec), np.ravel(nfa)], 1).astype(np.float64)
        except Exception as e:   # removed from some 3.4.x / 4.x builds for licence reasons
            print("LineSegmentDetector unavailable:", e)
        # line_descriptor::LSDDetector::detect(img, keylines, 1.2 -> scale 1, 1 octave) + BinaryDescriptor::compute (src/LSDextractor.cpp:12-29)
        try:
            ld = cv2.line_descriptor
            det = ld.LSDDetector_createLSDDetector()
            kls = det.detect(img, 1, 1)
            kls = sorted(kls, key=lambda k: -k.response)[:40]
            for i, k in enumerate(kls):
                k.class_id = i
            bd = ld.BinaryDescriptor_createBinaryDescriptor()
            kls, desc = bd.compute(img, kls)
            out[f"lbd/{s}/keylines"] = np.array([[k.angle, k.class_id, k.octave, k.pt[0], k.pt[1], k.response, k.size, k.startPointX, k.startPointY, k.endPointX,
                                                  k.endPointY, k.sPointInOctaveX, k.sPointInOctaveY, k.ePointInOctaveX, k.ePointInOctaveY, k.lineLength, k.numOfPixels]
                                                 for k in kls], np.float64)
            out[f"lbd/{s}/desc"] = np.asarray(desc, np.uint8)
        except Exception as e:
            print("line_descriptor python bindings unavailable:", e)
    # Frame::UndistortKeyPoints: cv::undistortPoints(mat, mat, K, D, R = empty, P = K) on tests/frame_cases.undistort_case()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frame_cases as fc
    keys, _ = fc.undistort_case()
    pts = np.stack([keys[0]["x"], keys[0]["y"]], 1).reshape(-1, 1, 2).astype(np.float32)
    for name, (K, D) in fc.DIST.items():
        Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32); Dm = np.array(D, np.float32).reshape(5, 1)
        out[f"undistort/{name}"] = cv2.undistortPoints(pts, Km, Dm, None, None, Km).reshape(-1, 2).astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "opencv_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays from OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()
