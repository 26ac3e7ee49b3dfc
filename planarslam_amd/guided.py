"""Host-side mirrors of the reference's guided matchers over the C ABI (batched over frames).

    ORBmatcher.SearchByProjection(CurrentFrame, LastFrame, th, bMono)   reference src/ORBmatcher.cc:1396 (include/ORBmatcher.h:45)
    ORBmatcher.SearchByProjection(F, vpMapPoints, th)                   reference src/ORBmatcher.cc:46   (include/ORBmatcher.h:41)
    ORBmatcher.SearchByBoW(pKF, F, vpMapPointMatches)                   reference src/ORBmatcher.cc:160  (include/ORBmatcher.h:59)
    ORBmatcher.Fuse(pKF, vpMapPoints, th), the search half              reference src/ORBmatcher.cc:829  (include/ORBmatcher.h:81)
    LSDmatcher.SearchByProjection(F, vpMapLines, th)                    reference src/LSDmatcher.cpp:141
    LSDmatcher.Fuse(pKF, vpMapLines, th), the search half               reference src/LSDmatcher.cpp:884
    PlaneMatcher.SearchMapByCoefficients(pF, vpMapPlanes)               reference src/PlaneMatcher.cpp:10

The reference's Frame / MapPoint objects become dicts of numpy arrays (the field names of include/planar_abi.h's
planar_frame_view / planar_last_frame_view / planar_map_probes).  No CPU fallback: everything runs in libplanar_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import KEYLINE_DTYPE, KP_DTYPE, MAX_LEVELS, Context, FrameView, LastFrameView, MapProbes, check, lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def frame_view(d: dict):
    """dict -> (planar_frame_view, keepalive).  Keys: n[B], keys_un[B,S] (KP_DTYPE), u_right[B,S], desc[B,S,32],
    optional blocked[B,S], Tcw[B,16]; scalars min_x,max_x,min_y,max_y,fx,fy,cx,cy,bf,b; scale_factors[levels]."""
    keep = dict(n=_c(d["n"], np.int32), keys_un=_c(d["keys_un"], KP_DTYPE), u_right=_c(d["u_right"], np.float32), desc=_c(d["desc"], np.uint8))
    B, S = keep["keys_un"].shape
    v = FrameView()
    v.B, v.stride = B, S
    for k in ("n", "keys_un", "u_right", "desc"):
        setattr(v, k, keep[k].ctypes.data)
    if d.get("blocked") is not None:
        keep["blocked"] = _c(d["blocked"], np.uint8); v.blocked = keep["blocked"].ctypes.data
    if d.get("Tcw") is not None:
        keep["Tcw"] = _c(d["Tcw"], np.float32).reshape(B, 16); v.Tcw = keep["Tcw"].ctypes.data
    for k in ("min_x", "max_x", "min_y", "max_y", "fx", "fy", "cx", "cy", "bf", "b"):
        setattr(v, k, float(d[k]))
    # Frame::mfGridElementWidthInv / HeightInv (src/Frame.cc:122-123)
    v.grid_w_inv = float(np.float32(64) / np.float32(np.float32(d["max_x"]) - np.float32(d["min_x"])))
    v.grid_h_inv = float(np.float32(48) / np.float32(np.float32(d["max_y"]) - np.float32(d["min_y"])))
    sf = _c(d["scale_factors"], np.float32)
    assert len(sf) <= MAX_LEVELS
    for i, x in enumerate(sf):
        v.scale_factors[i] = float(x)
    return v, keep


def last_frame_view(d: dict):
    keep = dict(n=_c(d["n"], np.int32), Tcw=_c(d["Tcw"], np.float32), usable=_c(d["usable"], np.uint8), xw=_c(d["xw"], np.float32),
                octave=_c(d["octave"], np.int32), angle=_c(d["angle"], np.float32), mp_desc=_c(d["mp_desc"], np.uint8),
                mp_observed=_c(d["mp_observed"], np.uint8))
    v = LastFrameView()
    v.stride = keep["usable"].shape[1]
    for k, a in keep.items():
        setattr(v, k, a.ctypes.data)
    return v, keep


def map_probes(d: dict):
    keep = dict(n=_c(d["n"], np.int32), in_view=_c(d["in_view"], np.uint8), proj_x=_c(d["proj_x"], np.float32), proj_y=_c(d["proj_y"], np.float32),
                proj_xr=_c(d["proj_xr"], np.float32), level=_c(d["level"], np.int32), view_cos=_c(d["view_cos"], np.float32),
                desc=_c(d["desc"], np.uint8), observed=_c(d["observed"], np.uint8))
    v = MapProbes()
    v.stride = keep["in_view"].shape[1]
    for k, a in keep.items():
        setattr(v, k, a.ctypes.data)
    return v, keep


class ORBmatcher:
    """Guided half of ORBmatcher (the brute-force half lives in matcher.ORBmatcher)."""

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True, ctx: Context | None = None):
        self.mfNNratio, self.mbCheckOrientation = nnratio, checkOri
        self.ctx = ctx or Context(0)

    def SearchByProjectionFrame(self, cur: dict, last: dict, th: float, bMono: bool = False, cur_match=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono).  Returns (cur_match [B,S] = last-frame keypoint
        index per current keypoint / -1, nmatches [B])."""
        fv, k1 = frame_view(cur)
        lv, k2 = last_frame_view(last)
        m = np.full((fv.B, fv.stride), -1, np.int32) if cur_match is None else _c(cur_match, np.int32).copy()
        nm = np.zeros(fv.B, np.int32)
        check(lib().planar_search_by_projection_frame(self.ctx.h, C.byref(fv), C.byref(lv), th, int(bMono), int(self.mbCheckOrientation),
                                                      m.ctypes.data, nm.ctypes.data))
        return m, nm

    def SearchByProjectionMap(self, frame: dict, probes: dict, th: float = 1.0, match=None):
        """SearchByProjection(F, vpMapPoints, th).  Returns (match [B,S] = map-point index per keypoint / -1, nmatches)."""
        fv, k1 = frame_view(frame)
        pv, k2 = map_probes(probes)
        m = np.full((fv.B, fv.stride), -1, np.int32) if match is None else _c(match, np.int32).copy()
        nm = np.zeros(fv.B, np.int32)
        check(lib().planar_search_by_projection_map(self.ctx.h, C.byref(fv), C.byref(pv), th, self.mfNNratio, m.ctypes.data, nm.ctypes.data))
        return m, nm

    def SearchByBoW(self, kf: dict, f: dict):
        """kf: n, node, usable, angle, desc ; f: n, node, angle, desc.  Returns (match [B,Sf] = key-frame feature index / -1, nmatches)."""
        a = dict(n=_c(kf["n"], np.int32), node=_c(kf["node"], np.int32), usable=_c(kf["usable"], np.uint8), angle=_c(kf["angle"], np.float32),
                 desc=_c(kf["desc"], np.uint8))
        b = dict(n=_c(f["n"], np.int32), node=_c(f["node"], np.int32), angle=_c(f["angle"], np.float32), desc=_c(f["desc"], np.uint8))
        B, ks = a["node"].shape
        fs = b["node"].shape[1]
        m = np.full((B, fs), -1, np.int32)
        nm = np.zeros(B, np.int32)
        check(lib().planar_search_by_bow(self.ctx.h, B, a["n"].ctypes.data, ks, a["node"].ctypes.data, a["usable"].ctypes.data, a["angle"].ctypes.data,
                                         a["desc"].ctypes.data, b["n"].ctypes.data, fs, b["node"].ctypes.data, b["angle"].ctypes.data,
                                         b["desc"].ctypes.data, self.mfNNratio, int(self.mbCheckOrientation), m.ctypes.data, nm.ctypes.data))
        return m, nm


    def Fuse(self, kf: dict, mp: dict, th: float = 3.0, inv_level_sigma2=None, log_scale_factor: float | None = None, n_levels: int | None = None, shared: bool = False):
        """Fuse(pKF, vpMapPoints, th) for B key frames (`kf`: a frame dict with Tcw), the search half: which keypoint of the key frame each map point
        would be fused with.  mp: n, usable (non-NULL, not bad, not yet in the key frame), xw, normal, min_dist, max_dist (mfMinDistance /
        mfMaxDistance), desc; shared: one list [1,S] for every key frame (LocalMapping::SearchInNeighbors).
        Returns (fuse_idx [B,S] keypoint index / -1, fuse_dist [B,S] best Hamming distance (256: no candidate), n_fused [B] = the return values)."""
        fv, keep = frame_view(kf)
        sf = np.asarray(kf["scale_factors"], np.float32)
        nl = n_levels or len(sf)
        lsf = float(np.float32(np.log(np.float32(sf[1])))) if log_scale_factor is None else log_scale_factor      # KeyFrame::mfLogScaleFactor
        # mvInvLevelSigma2[i] = 1 / (mvScaleFactors[i] * mvScaleFactors[i]) (src/ORBextractor.cc:441-446)
        inv = _c(1.0 / (sf[:nl] * sf[:nl]) if inv_level_sigma2 is None else inv_level_sigma2, np.float32)
        a = dict(n=_c(mp["n"], np.int32), usable=_c(mp["usable"], np.uint8), xw=_c(mp["xw"], np.float32), normal=_c(mp["normal"], np.float32),
                 min_dist=_c(mp["min_dist"], np.float32), max_dist=_c(mp["max_dist"], np.float32), desc=_c(mp["desc"], np.uint8))
        S = a["usable"].shape[-1]
        idx = np.full((fv.B, S), -1, np.int32); dist = np.full((fv.B, S), 256, np.int32); nf = np.zeros(fv.B, np.int32)
        check(lib().planar_fuse_search(self.ctx.h, C.byref(fv), inv.ctypes.data, lsf, nl, a["n"].ctypes.data, S, int(shared), a["usable"].ctypes.data, a["xw"].ctypes.data,
                                       a["normal"].ctypes.data, a["min_dist"].ctypes.data, a["max_dist"].ctypes.data, a["desc"].ctypes.data, th,
                                       idx.ctypes.data, dist.ctypes.data, nf.ctypes.data))
        return idx, dist, nf


class LSDmatcher:
    def __init__(self, nnratio: float = 0.6, ctx: Context | None = None):
        self.mfNNratio = nnratio
        self.ctx = ctx or Context(0)

    def SearchByProjection(self, lines: dict, maplines: dict, scale_factors, th: float = 1.0, match=None):
        """lines: n, keylines (KEYLINE_DTYPE [B,S]), ldesc, optional blocked ; maplines: n, in_view, proj [B,M,4], level,
        view_cos, desc, observed.  Returns (match [B,S] = map-line index / -1, nmatches)."""
        a = dict(n=_c(lines["n"], np.int32), kl=_c(lines["keylines"], KEYLINE_DTYPE), ldesc=_c(lines["ldesc"], np.uint8))
        blocked = _c(lines["blocked"], np.uint8) if lines.get("blocked") is not None else None
        b = dict(n=_c(maplines["n"], np.int32), in_view=_c(maplines["in_view"], np.uint8), proj=_c(maplines["proj"], np.float32),
                 level=_c(maplines["level"], np.int32), view_cos=_c(maplines["view_cos"], np.float32), desc=_c(maplines["desc"], np.uint8),
                 observed=_c(maplines["observed"], np.uint8))
        sf = _c(scale_factors, np.float32)
        B, S = a["kl"].shape
        M = b["in_view"].shape[1]
        m = np.full((B, S), -1, np.int32) if match is None else _c(match, np.int32).copy()
        nm = np.zeros(B, np.int32)
        check(lib().planar_lsd_search_by_projection(self.ctx.h, B, a["n"].ctypes.data, S, a["kl"].ctypes.data, a["ldesc"].ctypes.data,
                                                    blocked.ctypes.data if blocked is not None else None, b["n"].ctypes.data, M,
                                                    b["in_view"].ctypes.data, b["proj"].ctypes.data, b["level"].ctypes.data, b["view_cos"].ctypes.data,
                                                    b["desc"].ctypes.data, b["observed"].ctypes.data, sf.ctypes.data, len(sf), th, self.mfNNratio,
                                                    m.ctypes.data, nm.ctypes.data))
        return m, nm


def pose_view(kf: dict):
    """planar_frame_view holding only what the line matchers read of a key frame: B, Tcw [B,16], intrinsics, image bounds, scale factors"""
    keep = dict(Tcw=_c(kf["Tcw"], np.float32))
    v = FrameView()
    v.B, v.stride = keep["Tcw"].reshape(-1, 16).shape[0], 1
    v.Tcw = keep["Tcw"].ctypes.data
    for k in ("min_x", "max_x", "min_y", "max_y", "fx", "fy", "cx", "cy", "bf", "b"):
        setattr(v, k, float(kf[k]))
    for i, x in enumerate(_c(kf["scale_factors"], np.float32)):
        v.scale_factors[i] = float(x)
    return v, keep


def lsd_fuse(kf: dict, lines: dict, ml: dict, th: float = 3.0, log_scale_factor: float | None = None, n_levels: int | None = None, shared: bool = False,
             ctx: Context | None = None):
    """LSDmatcher::Fuse(pKF, vpMapLines, th) for B key frames, the search half (reference src/LSDmatcher.cpp:884-991).  kf: Tcw + intrinsics + bounds +
    scale_factors; lines: n, keylines [B,S] (KEYLINE_DTYPE), ldesc; ml: n, usable, xw6 (float64), normal (float64), min_dist, max_dist, desc.
    Returns (fuse_idx [B,S] key-line index / -1, fuse_dist [B,S] (INT_MAX: no candidate), n_fused [B])."""
    ctx = ctx or Context(0)
    fv, keep = pose_view(kf)
    sf = np.asarray(kf["scale_factors"], np.float32)
    nl = n_levels or len(sf)
    lsf = float(np.float32(np.log(np.float32(sf[1])))) if log_scale_factor is None else log_scale_factor
    a = dict(nl=_c(lines["n"], np.int32), kl=_c(lines["keylines"], KEYLINE_DTYPE), ld=_c(lines["ldesc"], np.uint8), n=_c(ml["n"], np.int32), usable=_c(ml["usable"], np.uint8),
             xw6=_c(ml["xw6"], np.float64), normal=_c(ml["normal"], np.float64), min_dist=_c(ml["min_dist"], np.float32), max_dist=_c(ml["max_dist"], np.float32),
             desc=_c(ml["desc"], np.uint8))
    S = a["usable"].shape[-1]
    idx = np.full((fv.B, S), -1, np.int32); dist = np.full((fv.B, S), 2 ** 31 - 1, np.int32); nf = np.zeros(fv.B, np.int32)
    check(lib().planar_lsd_fuse_search(ctx.h, C.byref(fv), lsf, nl, a["nl"].ctypes.data, a["kl"].shape[1], a["kl"].ctypes.data, a["ld"].ctypes.data, a["n"].ctypes.data, S,
                                       int(shared), a["usable"].ctypes.data, a["xw6"].ctypes.data, a["normal"].ctypes.data, a["min_dist"].ctypes.data,
                                       a["max_dist"].ctypes.data, a["desc"].ctypes.data, th, idx.ctypes.data, dist.ctypes.data, nf.ctypes.data))
    return idx, dist, nf


class PlaneMatcher:
    """include/PlaneMatcher.h:19 defaults."""

    def __init__(self, dTh=0.1, aTh=0.86, verTh=0.08716, parTh=0.9962, ctx: Context | None = None):
        self.th = np.array([dTh, aTh, verTh, parTh], np.float32)
        self.ctx = ctx or Context(0)

    def SearchMapByCoefficients(self, frame: dict, mapplanes: dict, init=None):
        """frame: n[B], coef[B,S,4], Tcw[B,16] ; mapplanes: n, valid, coef[.,M,4], npts[.,M], pts[.,M,P,3], shared(bool).
        Returns (match, ver, par [B,S] int32 (-1 = unassigned), nmatches[B])."""
        n = _c(frame["n"], np.int32); coef = _c(frame["coef"], np.float32); T = _c(frame["Tcw"], np.float32)
        mn = _c(mapplanes["n"], np.int32); mv = _c(mapplanes["valid"], np.uint8); mc = _c(mapplanes["coef"], np.float32)
        mnp = _c(mapplanes["npts"], np.int32); mp = _c(mapplanes["pts"], np.float32)
        B, S = coef.shape[:2]
        M, P = mp.shape[-3], mp.shape[-2]
        out = [np.full((B, S), -1, np.int32) if init is None else _c(init[i], np.int32).copy() for i in range(3)]
        nm = np.zeros(B, np.int32)
        check(lib().planar_plane_search_by_coefficients(self.ctx.h, B, n.ctypes.data, S, coef.ctypes.data, T.ctypes.data, int(bool(mapplanes.get("shared"))),
                                                        mn.ctypes.data, M, mv.ctypes.data, mc.ctypes.data, mnp.ctypes.data, P, mp.ctypes.data,
                                                        self.th.ctypes.data, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, nm.ctypes.data))
        return out[0], out[1], out[2], nm


class Frame:
    """The two Frame::isInFrustum overloads (reference src/Frame.cc:312-367, :369-438), batched over frames.  They fill the MapPoint /
    MapLine tracking fields that SearchByProjection(F, vpMapPoints / vpMapLines, th) reads, returned here as the probe dicts those take."""

    def __init__(self, frame: dict, log_scale_factor: float | None = None, n_levels: int | None = None, ctx: Context | None = None):
        self.frame, self.ctx = frame, ctx or Context(0)
        sf = np.asarray(frame["scale_factors"], np.float32)
        self.n_levels = n_levels or len(sf)
        # Frame::mfLogScaleFactor = log(mfScaleFactor), float (src/Frame.cc:67)
        self.log_scale_factor = float(np.float32(np.log(np.float32(sf[1])))) if log_scale_factor is None else log_scale_factor

    def isInFrustumPoints(self, mp: dict, viewingCosLimit: float = 0.5):
        """mp: n[B], valid, xw[B,S,3], normal[B,S,3], min_dist, max_dist (+ desc, observed passed through)."""
        fv, keep = frame_view(self.frame)
        a = dict(n=_c(mp["n"], np.int32), valid=_c(mp["valid"], np.uint8), xw=_c(mp["xw"], np.float32), normal=_c(mp["normal"], np.float32),
                 min_dist=_c(mp["min_dist"], np.float32), max_dist=_c(mp["max_dist"], np.float32))
        B, S = a["valid"].shape
        out = dict(n=a["n"], in_view=np.zeros((B, S), np.uint8), proj_x=np.zeros((B, S), np.float32), proj_y=np.zeros((B, S), np.float32),
                   proj_xr=np.zeros((B, S), np.float32), level=np.zeros((B, S), np.int32), view_cos=np.zeros((B, S), np.float32))
        check(lib().planar_is_in_frustum_points(self.ctx.h, C.byref(fv), self.log_scale_factor, self.n_levels, a["n"].ctypes.data, S, a["valid"].ctypes.data,
                                                a["xw"].ctypes.data, a["normal"].ctypes.data, a["min_dist"].ctypes.data, a["max_dist"].ctypes.data,
                                                viewingCosLimit, out["in_view"].ctypes.data, out["proj_x"].ctypes.data, out["proj_y"].ctypes.data,
                                                out["proj_xr"].ctypes.data, out["level"].ctypes.data, out["view_cos"].ctypes.data))
        for k in ("desc", "observed"):
            if k in mp:
                out[k] = mp[k]
        return out

    def isInFrustumLines(self, ml: dict, viewingCosLimit: float = 0.5):
        """ml: n[B], valid, xw6[B,S,6] float64, normal[B,S,3] float64, min_dist, max_dist (+ desc, observed passed through)."""
        fv, keep = frame_view(self.frame)
        a = dict(n=_c(ml["n"], np.int32), valid=_c(ml["valid"], np.uint8), xw6=_c(ml["xw6"], np.float64), normal=_c(ml["normal"], np.float64),
                 min_dist=_c(ml["min_dist"], np.float32), max_dist=_c(ml["max_dist"], np.float32))
        B, S = a["valid"].shape
        out = dict(n=a["n"], in_view=np.zeros((B, S), np.uint8), proj=np.zeros((B, S, 4), np.float32), level=np.zeros((B, S), np.int32),
                   view_cos=np.zeros((B, S), np.float32))
        check(lib().planar_is_in_frustum_lines(self.ctx.h, C.byref(fv), self.log_scale_factor, a["n"].ctypes.data, S, a["valid"].ctypes.data,
                                               a["xw6"].ctypes.data, a["normal"].ctypes.data, a["min_dist"].ctypes.data, a["max_dist"].ctypes.data,
                                               viewingCosLimit, out["in_view"].ctypes.data, out["proj"].ctypes.data, out["level"].ctypes.data,
                                               out["view_cos"].ctypes.data))
        for k in ("desc", "observed"):
            if k in ml:
                out[k] = ml[k]
        return out
