#!/usr/bin/env python3
"""Generate tests/golden/peac_*.npz: depth inputs + outputs of the REAL reference plane extractor
(oracle/_ref/ref_peac = /root/reference/src/PlaneExtractor.cpp + include/peac/*.hpp compiled against oracle/shim).
Runs only in the authoring container; the committed fixtures travel to the GPU box."""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from planarslam_amd.synth import depth_image

out = os.path.join(ROOT, "tests", "golden")
cases = {"room4321": depth_image(4321), "room7_clean": depth_image(7, noise=False, holes=False), "room9_noholes": depth_image(9, holes=False)}
rng = np.random.default_rng(3)
clutter = depth_image(11).astype(np.int32)
for _ in range(25):   # small fronto-parallel boxes: many depth discontinuities, many small segments
    x, y, w, h = rng.integers(0, 600), rng.integers(0, 440), rng.integers(20, 120), rng.integers(20, 100)
    clutter[y:y + h, x:x + w] = np.maximum(clutter[y:y + h, x:x + w] - rng.integers(500, 4000), 2500)
cases["clutter11"] = clutter.astype(np.uint16)
for name, d in cases.items():
    planes, labels = ol.run_ref_peac(d)
    arr = np.array([[p["N"], *p["normal"], *p["center"], p["mse"]] for p in planes], np.float64).reshape(-1, 8)
    np.savez_compressed(os.path.join(out, f"peac_{name}.npz"), depth=d, labels=labels.astype(np.int8), planes=arr)
    print(name, len(planes), float((labels >= 0).mean()))
