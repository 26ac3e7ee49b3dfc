"""Host-side mirror of the Frame-side glue of Tracking::Track over the C ABI (frame.hip), batched over frames.

    stereo_from_rgbd     Frame::ComputeStereoFromRGBD + UnprojectStereo   reference src/Frame.cc:603-634
    pose_assemble        the Frame fields Optimizer::PoseOptimization / TranslationOptimization read once the matchers ran
                         (src/Optimizer.cc:593-668, 689-745, 859-981), gathered from match indices into a pose batch
    discard_outliers     the loop after the optimiser, src/Tracking.cc:1784-1812
Arrays are numpy, [B, stride, ...]; no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import KP_DTYPE, Context, PoseBatch, TrackMatches, check, lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def undistort_keypoints(keys, n, cam, dist_coef, ctx: Context | None = None):
    """Frame::UndistortKeyPoints (reference src/Frame.cc:545-573): keys [B, stride] KP_DTYPE (mvKeys), dist_coef (k1, k2, p1, p2, k3) -> mvKeysUn [B, stride]."""
    ctx = ctx or Context(0)
    keys = _c(keys, KP_DTYPE); n = _c(n, np.int32); d = _c(dist_coef, np.float32)
    if d.shape != (5,):
        raise ValueError("dist_coef is (k1, k2, p1, p2, k3)")
    B, S = keys.shape
    out = np.zeros_like(keys)
    check(lib().planar_undistort_keypoints(ctx.h, B, keys.ctypes.data, n.ctypes.data, S, cam["fx"], cam["fy"], cam["cx"], cam["cy"], d.ctypes.data, out.ctypes.data))
    return out


def stereo_from_rgbd(keys, n, depth, Tcw, cam, depth_factor=1.0 / 5000.0, keys_un=None, ctx: Context | None = None):
    """keys [B, stride] KP_DTYPE (mvKeys; keys_un = mvKeysUn, default the same), depth [B, H, W] uint16, Tcw [B, 16].
    Returns dict(u_right, depth [B, stride] float32, xw [B, stride, 3] float32, valid [B, stride] uint8)."""
    ctx = ctx or Context(0)
    keys = _c(keys, KP_DTYPE); n = _c(n, np.int32); depth = _c(depth, np.uint16); Tcw = _c(Tcw, np.float32).reshape(len(n), 16)
    ku = keys if keys_un is None else _c(keys_un, KP_DTYPE)
    B, S = keys.shape
    H, W = depth.shape[1:]
    out = dict(u_right=np.zeros((B, S), np.float32), depth=np.zeros((B, S), np.float32), xw=np.zeros((B, S, 3), np.float32), valid=np.zeros((B, S), np.uint8))
    check(lib().planar_stereo_from_rgbd(ctx.h, B, keys.ctypes.data, ku.ctypes.data, n.ctypes.data, S, depth.ctypes.data, W, W * H, float(np.float32(depth_factor)),
                                        cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"], Tcw.ctypes.data, out["u_right"].ctypes.data,
                                        out["depth"].ctypes.data, out["xw"].ctypes.data, out["valid"].ctypes.data))
    return out


def track_matches(m: dict):
    """planar_track_matches over a dict of numpy arrays (device tensors are handled by the caller, see bench.py).  Returns (struct, keep-alive)."""
    t = TrackMatches()
    keep = {}

    def put(name, arr, dt):
        if arr is None:
            setattr(t, name, None); return None
        a = _c(arr, dt); keep[name] = a; setattr(t, name, a.ctypes.data); return a
    n = put("n", m["n"], np.int32)
    ku = put("keys_un", m["keys_un"], KP_DTYPE)
    t.B, t.stride = len(n), ku.shape[1]
    put("u_right", m["u_right"], np.float32); put("pt_match", m["pt_match"], np.int32)
    xw = put("mp_xw", m["mp_xw"], np.float32); t.mp_stride = xw.shape[1]
    put("mp_valid", m.get("mp_valid"), np.uint8)
    sig = np.asarray(m["inv_level_sigma2"], np.float32)
    t.n_levels = len(sig)
    for i, v in enumerate(sig):
        t.inv_level_sigma2[i] = float(v)
    if m.get("n_lines") is not None:
        put("n_lines", m["n_lines"], np.int32)
        le = put("line_eq", m["line_eq"], np.float64); t.ln_stride = le.shape[1]
        put("ln_match", m["ln_match"], np.int32)
        ml = put("ml_xw6", m["ml_xw6"], np.float64); t.ml_stride = ml.shape[1]
    if m.get("n_planes") is not None:
        put("n_planes", m["n_planes"], np.int32)
        pc = put("pl_coef", m["pl_coef"], np.float32); t.pl_stride = pc.shape[1]
        put("pl_match", m["pl_match"], np.int32)
        mc = put("mpl_coef", m["mpl_coef"], np.float32)
        t.mpl_shared = 1 if mc.ndim == 2 else 0
        t.mpl_stride = mc.shape[-2]
    put("Tcw", np.asarray(m["Tcw"], np.float32).reshape(t.B, 16), np.float32)
    return t, keep


def pose_assemble(m: dict, max_points: int, max_lines: int, max_planes: int, ctx: Context | None = None):
    """Returns the pose batch (dict of numpy arrays in planarslam_amd.synth.pose_batch's layout) PoseOptimization would read from the Frame."""
    ctx = ctx or Context(0)
    t, keep = track_matches(m)
    B = t.B
    ML, MM = max(max_lines, 1), max(max_planes, 1)
    out = dict(n_points=np.zeros(B, np.int32), n_lines=np.zeros(B, np.int32), n_planes=np.zeros(B, np.int32), pt_valid=np.zeros((B, max_points), np.uint8),
               pt_xw=np.zeros((B, max_points, 3), np.float32), pt_obs=np.zeros((B, max_points, 3), np.float32), pt_inv_sigma2=np.zeros((B, max_points), np.float32),
               ln_valid=np.zeros((B, ML), np.uint8), ln_obs=np.zeros((B, ML, 3)), ln_xw=np.zeros((B, ML, 6)), pl_meas=np.zeros((B, MM, 4), np.float32),
               pl_valid=np.zeros((B, MM, 3), np.uint8), pl_world=np.zeros((B, MM, 3, 4), np.float32), Tcw=np.zeros((B, 16), np.float32))
    pb = PoseBatch()
    pb.B, pb.max_points, pb.max_lines, pb.max_planes = B, max_points, max_lines, max_planes
    for k in ("n_points", "n_lines", "n_planes", "pt_valid", "pt_xw", "pt_obs", "pt_inv_sigma2", "ln_valid", "ln_obs", "ln_xw", "pl_meas", "pl_valid", "pl_world"):
        setattr(pb, k, out[k].ctypes.data)
    pb.Tcw_in = out["Tcw"].ctypes.data
    check(lib().planar_pose_assemble(ctx.h, C.byref(t), C.byref(pb)))
    if max_lines == 0:
        out["ln_valid"] = out["ln_valid"][:, :0]; out["ln_obs"] = out["ln_obs"][:, :0]; out["ln_xw"] = out["ln_xw"][:, :0]
    if max_planes == 0:
        out["pl_meas"] = out["pl_meas"][:, :0]; out["pl_valid"] = out["pl_valid"][:, :0]; out["pl_world"] = out["pl_world"][:, :0]
    return out


def discard_outliers(n, match, outlier, ctx: Context | None = None):
    """Returns (match with flagged entries set to -1, outlier flags cleared there, matches left per frame)."""
    ctx = ctx or Context(0)
    n = _c(n, np.int32); m = _c(match, np.int32).copy(); o = _c(outlier, np.uint8).copy()
    kept = np.zeros(len(n), np.int32)
    check(lib().planar_discard_outliers(ctx.h, len(n), n.ctypes.data, m.shape[1], o.shape[1], m.ctypes.data, o.ctypes.data, kept.ctypes.data))
    return m, o, kept


def update_normal_and_depth(n, xw, ref_Tcw, keys_un, scale_factors, valid=None, obs_off=None, obs_ow=None, ctx: Context | None = None):
    """MapPoint::UpdateNormalAndDepth (reference src/MapPoint.cc:347-388) for G groups of points (group = the points of one reference key frame):
    n [G], xw [G, stride, 3], ref_Tcw [G, 16], keys_un [G, stride] KP_DTYPE, optional per-point observer centres (obs_off [G*stride+1], obs_ow [m, 3]).
    Returns normal [G, stride, 3], min_dist, max_dist [G, stride] (zeros where nothing was written)."""
    ctx = ctx or Context(0)
    n = _c(n, np.int32); xw = _c(xw, np.float32); T = _c(ref_Tcw, np.float32).reshape(len(n), 16); ku = _c(keys_un, KP_DTYPE); sf = _c(scale_factors, np.float32)
    G, S = xw.shape[:2]
    v = None if valid is None else _c(valid, np.uint8)
    oo = None if obs_off is None else _c(obs_off, np.int32)
    ow = None if obs_ow is None else _c(obs_ow, np.float32)
    nrm = np.zeros((G, S, 3), np.float32); mn = np.zeros((G, S), np.float32); mx = np.zeros((G, S), np.float32)
    check(lib().planar_update_normal_and_depth(ctx.h, G, n.ctypes.data, S, xw.ctypes.data, None if v is None else v.ctypes.data, T.ctypes.data, ku.ctypes.data,
                                               None if oo is None else oo.ctypes.data, None if ow is None else ow.ctypes.data, sf.ctypes.data, len(sf), nrm.ctypes.data,
                                               mn.ctypes.data, mx.ctypes.data))
    return nrm, mn, mx
