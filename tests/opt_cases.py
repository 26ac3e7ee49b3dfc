"""Seeded problems shared by tools/gen_golden_opt.py (which runs the REAL reference optimiser, oracle/_ref/ref_opt, on them and stores
its outputs in tests/golden/opt_ref.npz), tests/test_oracle_opt_ref.py (oracle vs those outputs) and the GPU tests (HIP vs those outputs)."""
import numpy as np

from planarslam_amd import synth


def _ragged():
    b = synth.pose_batch(B=6, n_points=300, n_lines=20, n_planes=3, seed=31, max_points=512, max_lines=40, max_planes=8)
    b["n_points"][:] = [300, 120, 7, 300, 64, 299]
    b["n_lines"][:] = [20, 0, 3, 20, 1, 19]
    b["n_planes"][:] = [3, 0, 1, 2, 3, 0]
    return b


def _few():
    b = synth.pose_batch(B=2, n_points=10, n_lines=0, n_planes=0, seed=3)
    b["pt_valid"][:] = 0
    b["pt_valid"][:, :2] = 1
    return b


def _planes_partly_missing():
    b = synth.pose_batch(B=4, n_points=400, n_lines=30, n_planes=6, seed=77)
    b["pl_valid"][:, ::2, 1] = 0      # no parallel association for every other plane
    b["pl_valid"][:, 1::3, 2] = 0     # no vertical association for every third
    b["pl_valid"][1, :, 0] = 0        # a frame without matched planes
    return b


# name -> (builder, modes)
POSE_CASES = {
    "c4_b8": (lambda: synth.pose_batch(B=8, seed=7), (0, 1)),                       # BASELINE config 4 shape
    "c4_b256": (lambda: synth.pose_batch(B=256, seed=1000), (0, 1)),                # BASELINE config 4 at its batch size
    "ragged": (_ragged, (0, 1)),
    "few": (_few, (0, 1)),
    "points_only": (lambda: synth.pose_batch(B=3, n_lines=0, n_planes=0, seed=9, max_lines=4, max_planes=2), (0, 1)),
    "planes_partly_missing": (_planes_partly_missing, (0, 1)),
    "no_outliers": (lambda: synth.pose_batch(B=4, seed=100, outlier_frac=0.0), (0,)),
    "far_start": (lambda: synth.pose_batch(B=4, seed=41, rot_pert=0.08, trans_pert=0.25), (0, 1)),
}


def _ba(cur, **kw):
    return synth.ba_local_only(synth.ba_problem(lines_on_kf=cur, **kw))


# name -> (builder, current keyframe)
BA_CASES = {
    "small": (lambda: _ba(9, seed=5, n_points=300, n_lines=60, n_planes=12), 9),
    "config5": (lambda: _ba(9, seed=99), 9),                                         # BASELINE config 5: 2400 + 500 + 100 features, 10 keyframes
    "points_only": (lambda: _ba(2, seed=11, n_kf=3, n_points=200, n_lines=0, n_planes=0, n_fixed_extra=4), 2),
    "cur_in_the_middle": (lambda: _ba(4, seed=23, n_kf=8, n_points=500, n_lines=100, n_planes=20), 4),
    "no_outliers": (lambda: _ba(5, seed=37, n_kf=6, n_points=400, n_lines=80, n_planes=10, outlier_frac=0.0), 5),
}

EDGE_CASES = (32, 3)   # n, seed
