"""Measurement aid: PoseOptimization 4x10 on BASELINE config 4 (1024 frames) with the two register-allocation variants of pose_opt_kernel
(PLANAR_POSE_WAVES=1|2, read once per process).  Prints ms per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from planarslam_amd import Optimizer, synth  # noqa: E402

b = synth.pose_batch(B=1024, seed=1000)
opt = Optimizer(synth.TUM3)
for mode, name in ((0, "PoseOptimization"), (1, "TranslationOptimization")):
    f = opt.PoseOptimization if mode == 0 else opt.TranslationOptimization
    f(b, 4, 10)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); r = f(b, 4, 10); t.append(time.perf_counter() - t0)
    print(os.environ.get("PLANAR_POSE_WAVES", "default"), name, "ms per 1024 frames (host buffers in/out):", round(min(t) * 1e3, 2), "inliers", int(r["n_inliers"].sum()))
