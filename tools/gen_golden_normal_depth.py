#!/usr/bin/env python3
"""tests/golden/normal_depth_ref.npz: normal / min / max distance of tests/normal_depth_cases.py after the REFERENCE's own MapPoint::UpdateNormalAndDepth
(oracle/_ref/ref_frame normal_depth = src/MapPoint.cc:347-388 + src/KeyFrame.cc:79-93, 107-111).  PYTHONPATH=.:tests python tools/gen_golden_normal_depth.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import normal_depth_cases as nc  # noqa: E402
import oracle_lib as ol  # noqa: E402

Tcws, sf, pts = nc.cases()
out = ol.run_ref_normal_depth(Tcws, sf, pts)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "normal_depth_ref.npz"), out=out)
print("wrote", out.shape)
