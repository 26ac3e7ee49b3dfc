#!/usr/bin/env python3
"""Generate tests/golden/orb_*.npz: inputs + outputs of the REAL reference ORBextractor
(oracle/_ref/ref_orb = /root/reference/src/ORBextractor.cc compiled against oracle/shim).

Runs only in the authoring container (needs /root/reference for the natural images and for
building oracle/_ref).  The committed fixtures travel to the GPU box, where /root/reference
does not exist.
"""
import os, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from planarslam_amd.synth import gray_image

out = os.path.join(ROOT, "tests", "golden")
cases = {}
for name in ["R1", "OFF1"]:
    a = np.asarray(Image.open(f"/root/reference/Examples/{name}.png").convert("L"))
    cases[name] = (a[40:520, 60:700].copy(), dict(nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7))
cases["synth1234"] = (gray_image(1234), dict(nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7))
rng = np.random.default_rng(5)
cases["noise"] = (rng.integers(0, 256, (240, 320)).astype(np.uint8), dict(nfeatures=500, scale=1.2, nlevels=4, ini=20, mn=7))
cases["odd517"] = (gray_image(10, 517, 389), dict(nfeatures=700, scale=1.3, nlevels=5, ini=25, mn=9))
for name, (img, p) in cases.items():
    kps, desc, pyr = ol.run_ref_orb(img, **p)
    np.savez_compressed(os.path.join(out, f"orb_{name}.npz"), image=img, kps=kps, desc=desc,
                        params=np.array([p["nfeatures"], p["scale"], p["nlevels"], p["ini"], p["mn"]], np.float64),
                        level_shapes=np.array([q.shape for q in pyr], np.int32),
                        level_sums=np.array([int(q.astype(np.int64).sum()) for q in pyr], np.int64),
                        last_level=pyr[-1])
    print(name, img.shape, len(kps))
