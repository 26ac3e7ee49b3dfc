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
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fuse_points_ref.npz"), **out)
