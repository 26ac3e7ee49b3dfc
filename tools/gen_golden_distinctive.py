#!/usr/bin/env python3
"""tests/golden/distinctive_ref.npz: mDescriptor of every case of tests/distinctive_cases.py after the REFERENCE's own MapPoint::ComputeDistinctiveDescriptors
(oracle/_ref/ref_frame distinctive = src/MapPoint.cc:259-324 + ORBmatcher::DescriptorDistance compiled against oracle/shim).  Run in the container that has
/root/reference:  PYTHONPATH=.:tests python tools/gen_golden_distinctive.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import distinctive_cases as dc  # noqa: E402
import oracle_lib as ol  # noqa: E402

cs = dc.cases()
rng = np.random.default_rng(9)
bad = [(rng.uniform(size=len(d)) < (0.15 if i % 4 == 0 else 0.0)).astype(np.uint8) for i, d in enumerate(cs)]
out = ol.run_ref_distinctive(list(zip(cs, bad)))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "distinctive_ref.npz"), chosen=out, bad=np.concatenate(bad + [np.zeros(0, np.uint8)]),
                    n=np.array([len(d) for d in cs], np.int32))
# MapLine::ComputeDistinctiveDescriptors (src/MapLine.cpp:241-312, cv::norm NORM_HAMMING on LBD rows): same cases under another seed
csl = dc.cases(seed=17, n_points=40)
badl = [(rng.uniform(size=len(d)) < (0.2 if i % 3 == 0 else 0.0)).astype(np.uint8) for i, d in enumerate(csl)]
outl = ol.run_ref_distinctive(list(zip(csl, badl)), lines=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "distinctive_lines_ref.npz"), chosen=outl, bad=np.concatenate(badl + [np.zeros(0, np.uint8)]),
                    n=np.array([len(d) for d in csl], np.int32))
print("wrote", len(cs), "point cases,", len(csl), "line cases")
