"""Shared inputs of the ORBmatcher::Fuse tests (CPU oracle tests, GPU tests, tools/gen_golden_fuse.py)."""
import numpy as np

from planarslam_amd import synth


def scale():
    return float(np.float32(np.log(np.float32(1.2)))), 8       # KeyFrame::mfLogScaleFactor = log(mfScaleFactor) (float), mnScaleLevels


def fuse_case(seed=131, B=3, N=1000, n_points=3000, hit=0.6):
    fr = synth.guided_frame(B=B, N=N, seed=seed, crowd=0.4)
    return synth.guided_fuse_points(fr, seed=seed + 1, n_points=n_points, hit=hit)


def kf_map_points(kf, seed=7):
    """which keypoints of the key frames already carry a map point (0 none, 1 yes, 2 a bad one) and its Observations(): only the map edits of
    Fuse read them, the search half does not"""
    rng = np.random.default_rng(seed)
    shape = kf["keys_un"].shape
    return rng.choice([0, 1, 2], shape, p=[0.5, 0.4, 0.1]).astype(np.uint8), rng.integers(1, 9, shape).astype(np.int32)


def fuse_lines_case(seed=171, B=3, n_lines=60, n_ml=300):
    return synth.guided_fuse_lines(B=B, n_lines=n_lines, n_ml=n_ml, seed=seed)
