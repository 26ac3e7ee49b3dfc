"""Seeded cases shared by tools/gen_golden_line3d.py (the REAL reference's Frame::isLineGood -> tests/golden/line3d_ref.npz), the oracle test and the GPU test."""
import numpy as np

import oracle_lib as O
from planarslam_amd import synth

# name -> (gray seed, depth seed, noise, holes, rand seed, extra)
CASES = {
    "scene_a": (11, 111, True, True, 7, None),
    "scene_b_noise_free": (12, 112, False, True, 0, None),
    "scene_c_big_seed": (13, 113, True, False, 4000000000, None),
    "rough_depth": (14, 114, True, True, 99, "rough"),          # strong depth noise: RANSAC rejects samples, lines fail the 0.4 support ratio
    "half_without_depth": (15, 115, True, True, 5, "half"),      # fewer than 10 samples on many lines
}


def build(name):
    gs, ds, noise, holes, seed, extra = CASES[name]
    kl = O.extract_line_segment(synth.gray_image(gs), tie_order=0)[0]
    d = synth.depth_image(ds, noise=noise, holes=holes)
    if extra == "rough":
        rng = np.random.default_rng(ds)
        d = np.clip(d.astype(np.int64) + rng.normal(0, 400, d.shape).astype(np.int64) * (rng.random(d.shape) < 0.3), 0, 65535).astype(np.uint16)
    if extra == "half":
        d = d.copy(); d[:, 300:] = 0
    return kl, d, seed
