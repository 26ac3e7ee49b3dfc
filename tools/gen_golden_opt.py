"""Generate tests/golden/opt_ref.npz from the REAL reference optimiser: oracle/_ref/ref_opt = src/Optimizer.cc + src/Converter.cc + the
vendored g2o + g2oAddition/*.h + include/EdgeLine.h compiled where they lie (recipe: oracle/Makefile).  Inputs are regenerated from seeds
(tests/opt_cases.py); only the reference's outputs are stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import opt_cases as cases  # noqa: E402
from planarslam_amd import synth  # noqa: E402

out = {}
for name, (build, modes) in cases.POSE_CASES.items():
    b = build()
    for mode in modes:
        r = O.run_ref_pose(b, synth.TUM3, mode)
        for k in ("Tcw", "n_inliers"):
            out[f"pose/{name}/{mode}/{k}"] = r[k]
        for k in ("pt_outlier", "ln_outlier", "pl_outlier"):
            out[f"pose/{name}/{mode}/{k}"] = np.packbits(r[k])
for name, (build, cur) in cases.BA_CASES.items():
    pr = build()
    r = O.run_ref_local_ba(pr, synth.TUM3, cur)
    out[f"ba/{name}/kf_Tcw"] = r["kf_Tcw"]; out[f"ba/{name}/lm"] = r["lm"]; out[f"ba/{name}/e_outlier"] = np.packbits(r["e_outlier"])
n, seed = cases.EDGE_CASES
e, H = O.run_ref_edges(O._edge_cases(n, seed), synth.TUM3)
out["edges/rec"] = e; out["edges/H"] = H
path = os.path.join(ROOT, "tests", "golden", "opt_ref.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
