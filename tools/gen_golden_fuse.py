"""Generates tests/golden/fuse_points_ref.npz from the REAL ORBmatcher::Fuse (oracle/_ref/ref_match, mode fuse: src/ORBmatcher.cc:829-979 with
KeyFrame::GetFeaturesInArea / IsInImage and MapPoint::PredictScale compiled in from the reference tree).  Run in the container that has /root/reference:
    make -C oracle _ref/ref_match && PYTHONPATH=.:tests python tools/gen_golden_fuse.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuse_cases as cases
import oracle_lib as O

kf, mp = cases.fuse_case(seed=131)
lsf, nlev = cases.scale()
state, kobs = cases.kf_map_points(kf)
out = {}
B, S = mp["usable"].shape
for th in (3.0, 1.5):
    idx = np.full((B, S), -1, np.int32); nf = np.zeros(B, np.int32)
    for b in range(B):
        r, n = O.ref_fuse(kf, mp, b, th, lsf, nlev, kf_state=state[b], kf_obs=kobs[b])
        idx[b, :len(r)] = r; nf[b] = n
    out[f"fuse_idx_th{th}"] = idx; out[f"n_fused_th{th}"] = nf
    print("th", th, "fused per key frame", nf, "of", mp["n"])
# LSDmatcher::Fuse: the real function reads mvScaleFactors[level] with an unclamped level; lines whose level leaves the pyramid (skipped by the product and
# the oracle) are kept out of its input, with a margin for float rounding
kfl, lines, ml = cases.fuse_lines_case()
rng = np.random.default_rng(5)
lstate = rng.choice([0, 1, 2], lines["keylines"].shape, p=[0.5, 0.4, 0.1]).astype(np.uint8); lobs = rng.integers(1, 9, lstate.shape).astype(np.int32)
Bl, Sl = ml["usable"].shape
for th in (3.0, 6.0):
    idx = np.full((Bl, Sl), -1, np.int32); nf = np.zeros(Bl, np.int32)
    for b in range(Bl):
        n = int(ml["n"][b])
        T = kfl["Tcw"][b].reshape(4, 4).astype(np.float64)
        Ow = -T[:3, :3].T @ T[:3, 3]
        q = np.log(ml["max_dist"][b, :n] / np.linalg.norm(0.5 * (ml["xw6"][b, :n, :3] + ml["xw6"][b, :n, 3:]) - Ow, axis=1)) / lsf
        safe = dict(ml); safe["usable"] = ml["usable"].copy(); safe["usable"][b, :n][(q <= -1 + 1e-3) | (q > nlev - 1 - 1e-3)] = 0
        r, k = O.ref_lsd_fuse(kfl, lines, safe, b, th, lsf, nlev, kf_state=lstate[b], kf_obs=lobs[b])
        idx[b, :len(r)] = r; nf[b] = k
    out[f"lsd_fuse_idx_th{th}"] = idx; out[f"lsd_n_fused_th{th}"] = nf
    print("lines th", th, "fused per key frame", nf, "of", ml["n"])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fuse_points_ref.npz"), **out)
