"""tests/golden/line3d_ref.npz: what the REAL reference's Frame::isLineGood (oracle/_ref/ref_line3d: src/Frame.cc:189-267 + src/LineExtractor.cpp helpers extracted by line
range, srand(seed + i) before line i) returns on the seeded cases of tests/line3d_cases.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import line3d_cases as cases  # noqa: E402
import oracle_lib as O  # noqa: E402

out = {}
for name in cases.CASES:
    kl, d, seed = cases.build(name)
    r = O.run_ref_line3d(kl, d, seed)
    for k, v in r.items():
        out[f"{name}/{k}"] = v
    print(name, len(kl), "lines,", int(r["good"].sum()), "good, inliers", r["n_inliers"][r["good"] > 0][:8])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "line3d_ref.npz"), **out)
