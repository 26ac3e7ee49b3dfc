"""Generate tests/golden/frame_ref.npz from the REAL reference function bodies: oracle/_ref/ref_frame = line ranges of src/Tracking.cc
(TrackManhattanFrame, ProjectSN2Conic, ProjectSN2MF, MeanShift), src/Frame.cc (isInFrustum x2, ...), src/MapPoint.cc, src/MapLine.cpp extracted
at build time and compiled against the cv::Mat stand-in (recipe: oracle/Makefile).  Inputs are regenerated from seeds (tests/frame_cases.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import frame_cases as cases  # noqa: E402
import oracle_lib as O  # noqa: E402
from planarslam_amd import synth  # noqa: E402

out = {}
for name, kw in cases.MANHATTAN_CASES.items():
    sc = synth.manhattan_scene(**kw)
    B = len(sc["n_normals"])
    R = np.zeros((B, 3, 3), np.float32)
    member = np.zeros((B, sc["normals"].shape[1] + sc["lines"].shape[1]), np.uint8)
    for b in range(B):
        n, m = int(sc["n_normals"][b]), int(sc["n_lines"][b])
        r = O.run_ref_manhattan(sc["R_last"][b], sc["normals"][b, :n], sc["lines"][b, :m])
        R[b] = r["R"]; member[b, :n] = r["member"][:n]; member[b, sc["normals"].shape[1]:sc["normals"].shape[1] + m] = r["member"][n:]
    out[f"manhattan/{name}/R"] = R; out[f"manhattan/{name}/member"] = member
R0, MF, T = cases.manhattan_pose_case()
out["manhattan_pose/Tcw"] = O.run_ref_manhattan_pose(R0, MF, T)      # src/Tracking.cc:251-253 + :1778
frame, mp, ml = cases.frustum_case()
lsf, nlev = cases.frustum_scale()
B, S = mp["valid"].shape
pt = dict(in_view=np.zeros((B, S), np.uint8), proj_x=np.zeros((B, S), np.float32), proj_y=np.zeros((B, S), np.float32), proj_xr=np.zeros((B, S), np.float32),
          level=np.zeros((B, S), np.int32), view_cos=np.zeros((B, S), np.float32))
SL = ml["valid"].shape[1]
ln = dict(in_view=np.zeros((B, SL), np.uint8), proj=np.zeros((B, SL, 4), np.float32), level=np.zeros((B, SL), np.int32), view_cos=np.zeros((B, SL), np.float32))
for b in range(B):
    idx, rec = O.run_ref_frustum_points(frame, mp, b, lsf, nlev, 0.5)
    assert np.array_equal(rec["ret"], rec["in_view"])
    pt["in_view"][b, idx] = rec["in_view"]; pt["proj_x"][b, idx] = rec["px"]; pt["proj_y"][b, idx] = rec["py"]; pt["proj_xr"][b, idx] = rec["pxr"]
    pt["level"][b, idx] = rec["level"]; pt["view_cos"][b, idx] = rec["vc"]
    idx, rec = O.run_ref_frustum_lines(frame, ml, b, lsf, 0.5)
    ln["in_view"][b, idx] = rec["ret"]; ln["proj"][b, idx] = rec["p"]; ln["level"][b, idx] = rec["level"]; ln["view_cos"][b, idx] = rec["vc"]
for k, v in pt.items():
    out[f"frustum/points/{k}"] = v
for k, v in ln.items():
    out[f"frustum/lines/{k}"] = v
keys, n, depth, Tcw = cases.stereo_case()
B, S = keys.shape
st = dict(u_right=np.full((B, S), -1, np.float32), depth=np.full((B, S), -1, np.float32), xw=np.zeros((B, S, 3), np.float32))
for b in range(B):
    rec = O.run_ref_stereo(keys[b, :n[b]], depth[b], Tcw[b], synth.TUM3)
    st["u_right"][b, :n[b]] = rec["u_right"]; st["depth"][b, :n[b]] = rec["depth"]; st["xw"][b, :n[b]] = rec["xw"]
for k, v in st.items():
    out[f"stereo/{k}"] = v
path = os.path.join(ROOT, "tests", "golden", "frame_ref.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
