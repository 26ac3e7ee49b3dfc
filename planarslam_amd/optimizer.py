"""Host-side mirror of Planar_SLAM::Optimizer's per-frame entry points (reference include/Optimizer.h:37,43)
over the C ABI: PoseOptimization / TranslationOptimization on a batch of frames given as the
structure-of-arrays described in include/planar_abi.h (`planar_pose_batch`)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, PoseBatch, PoseParams, check, lib

POSE_FULL, POSE_TRANSLATION = 0, 1
_IN = ("n_points", "n_lines", "n_planes", "pt_valid", "pt_xw", "pt_obs", "pt_inv_sigma2", "ln_valid", "ln_obs", "ln_xw",
       "pl_meas", "pl_valid", "pl_world")


def make_params(d) -> PoseParams:
    return PoseParams(d["fx"], d["fy"], d["cx"], d["cy"], d["bf"], d["angle_info"], d["distance_info"], d["parallel_info"],
                      d["vertical_info"], d["plane_chi"], d["vp_chi"])


class Optimizer:
    def __init__(self, params: dict, ctx: Context | None = None):
        self.L = lib()
        self.ctx = ctx or Context(0)
        self.params = make_params(params)

    def _run(self, batch: dict, mode: int, rounds: int, its: int):
        B = len(batch["n_points"])
        MP, ML, MM = batch["pt_valid"].shape[1], batch["ln_valid"].shape[1], batch["pl_valid"].shape[1]
        out = dict(Tcw=np.zeros((B, 16), np.float32), pt_outlier=np.zeros((B, MP), np.uint8), ln_outlier=np.zeros((B, ML), np.uint8),
                   pl_outlier=np.zeros((B, MM, 3), np.uint8), n_inliers=np.zeros(B, np.int32), lm_iters=np.zeros(B, np.int32))
        pb = PoseBatch()
        pb.B, pb.max_points, pb.max_lines, pb.max_planes = B, MP, ML, MM
        keep = []
        for k in _IN:
            a = np.ascontiguousarray(batch[k]); keep.append(a); setattr(pb, k, a.ctypes.data)
        tin = np.ascontiguousarray(batch["Tcw"], np.float32); keep.append(tin)
        pb.Tcw_in = tin.ctypes.data
        pb.Tcw_out, pb.pt_outlier, pb.ln_outlier = out["Tcw"].ctypes.data, out["pt_outlier"].ctypes.data, out["ln_outlier"].ctypes.data
        pb.pl_outlier, pb.n_inliers, pb.lm_iters = out["pl_outlier"].ctypes.data, out["n_inliers"].ctypes.data, out["lm_iters"].ctypes.data
        check(self.L.planar_pose_opt(self.ctx.h, C.byref(pb), C.byref(self.params), mode, rounds, its))
        return out

    def PoseOptimization(self, batch: dict, rounds: int = 4, its: int = 10):
        """Optimizer::PoseOptimization(Frame*) for every frame of `batch` (synth.pose_batch layout)."""
        return self._run(batch, POSE_FULL, rounds, its)

    def TranslationOptimization(self, batch: dict, rounds: int = 4, its: int = 10):
        return self._run(batch, POSE_TRANSLATION, rounds, its)

    def enqueue_dev(self, pb: PoseBatch, mode: int = POSE_FULL, rounds: int = 4, its: int = 10):
        """Device-pointer batch (all array fields are device addresses); enqueue only."""
        check(self.L.planar_pose_opt_dev(self.ctx.h, C.byref(pb), C.byref(self.params), mode, rounds, its))
