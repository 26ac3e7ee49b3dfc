"""Host-side mirror of the reference's PlaneDetection (include/PlaneExtractor.h:36-56) over the C ABI:
readDepthImage + runPlaneDetection become one batched call."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, PlanarError, check, lib


class PlaneDetection:
    def __init__(self, width: int = 640, height: int = 480, max_batch: int = 1, ctx: Context | None = None):
        self.L = lib()
        self.ctx = ctx or Context(0)
        self.width, self.height, self.max_batch = width, height, max_batch
        h = C.c_void_p()
        check(self.L.planar_peac_create(self.ctx.h, width, height, max_batch, C.byref(h)))
        self.h = h
        self.max_planes = self.L.planar_peac_max_planes()

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_peac_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, depth: np.ndarray, K=(535.4, 539.2, 320.1, 247.6), depth_factor: float = 1.0 / 5000.0):
        """depth: (H,W) or (B,H,W) uint16 (readDepthImage refuses anything else, src/PlaneExtractor.cpp:34-38).
        Returns per frame (planes [n,8] = N, normal, center, mse; labels [H,W] int32, -1 = no plane)."""
        if depth.dtype != np.uint16:
            raise TypeError("depth image must be CV_16U (uint16)")
        single = depth.ndim == 2
        d = np.ascontiguousarray(depth[None] if single else depth)
        B, H, W = d.shape
        if (W, H) != (self.width, self.height):
            raise ValueError(f"detector was created for {self.width}x{self.height}, got {W}x{H}")
        labels = np.zeros((B, H, W), np.int32)
        planes = np.zeros((B, self.max_planes, 8), np.float64)
        n = np.zeros(B, np.int32)
        check(self.L.planar_peac_segment(self.h, d.ctypes.data, B, W, W * H, K[0], K[1], K[2], K[3], np.float32(depth_factor),
                                         labels.ctypes.data, planes.ctypes.data, n.ctypes.data))
        res = [(planes[b, :n[b]].copy(), labels[b]) for b in range(B)]
        return res[0] if single else res

    def segment_dev(self, d_depth, d_labels, d_planes, d_n, B, K=(535.4, 539.2, 320.1, 247.6), depth_factor=1.0 / 5000.0):
        check(self.L.planar_peac_segment_dev(self.h, d_depth, B, self.width, self.width * self.height, K[0], K[1], K[2], K[3],
                                             np.float32(depth_factor), d_labels, d_planes, d_n))


class SurfaceNormals:
    """The tail of Frame::ComputePlanes (reference src/Frame.cc:694-751): depth -> the SurfaceNormal list Tracking::TrackManhattanFrame reads."""

    def __init__(self, width: int = 640, height: int = 480, max_batch: int = 1, ctx: Context | None = None):
        self.L = lib()
        self.ctx = ctx or Context(0)
        self.width, self.height, self.max_batch = width, height, max_batch
        h = C.c_void_p()
        check(self.L.planar_normals_create(self.ctx.h, width, height, max_batch, C.byref(h)))
        self.h = h
        self.count = check(self.L.planar_normals_count(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_normals_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compute(self, depth: np.ndarray, K=(535.4, 539.2, 320.1, 247.6), depth_factor: float = 1.0 / 5000.0):
        """depth (B,H,W) or (H,W) uint16 -> normals [B, count, 3] float32 (NaN = none), points [B, count, 3] (SurfaceNormal::cameraPosition)."""
        if depth.dtype != np.uint16:
            raise TypeError("depth image must be CV_16U (uint16)")
        single = depth.ndim == 2
        d = np.ascontiguousarray(depth[None] if single else depth)
        B, H, W = d.shape
        if (W, H) != (self.width, self.height):
            raise ValueError(f"created for {self.width}x{self.height}, got {W}x{H}")
        nrm = np.zeros((B, self.count, 3), np.float32); pts = np.zeros((B, self.count, 3), np.float32)
        check(self.L.planar_normals_compute(self.h, d.ctypes.data, B, W, W * H, K[0], K[1], K[2], K[3], np.float32(depth_factor), nrm.ctypes.data, pts.ctypes.data))
        return (nrm[0], pts[0]) if single else (nrm, pts)

    def compute_dev(self, d_depth, d_normals, B, K=(535.4, 539.2, 320.1, 247.6), depth_factor=1.0 / 5000.0, d_points=None, out_stride=None):
        check(self.L.planar_normals_compute_dev(self.h, d_depth, B, self.width, self.width * self.height, K[0], K[1], K[2], K[3], np.float32(depth_factor),
                                                d_normals, d_points, out_stride or self.count))


REFIT_INFO = ("iterations", "best_count", "s0", "s1", "s2", "n_inliers", "n_inliers_refined", "draws")


def _refit_info(raw):
    raw = np.asarray(raw, np.int32)
    d = {k: int(raw[i]) for i, k in enumerate(REFIT_INFO)}
    d["model"] = raw[8:12].copy().view(np.float32)
    return d


class PlaneClouds:
    """The head of Frame::ComputePlanes (reference src/Frame.cc:652-692) with Frame::MaxPointDistanceFromPlane (:755-812): the detector's planes ->
    mvPlanePoints (0.1 m voxel centroids) and mvPlaneCoefficients (RANSAC refit), planes failing Plane.DistanceThreshold dropped."""

    def __init__(self, width: int = 640, height: int = 480, max_batch: int = 1, max_points: int = 4096, ctx: Context | None = None):
        self.L = lib()
        self.ctx = ctx or Context(0)
        self.width, self.height, self.max_batch, self.max_points = width, height, max_batch, max_points
        h = C.c_void_p()
        check(self.L.planar_plane_clouds_create(self.ctx.h, width, height, max_batch, max_points, C.byref(h)))
        self.h = h
        ps = C.c_int()
        check(self.L.planar_plane_clouds_stride(self.h, C.byref(ps), None))
        self.pl_stride = ps.value

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_plane_clouds_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compute(self, depth, labels, planes, n_planes, dist_th=0.05, leaf=0.1, K=(535.4, 539.2, 320.1, 247.6), depth_factor=1.0 / 5000.0, debug=False, retry_per_plane=True):
        """depth [B,H,W] u16, labels [B,H,W] i32, planes [B,max_planes,8] f64, n_planes [B] (PlaneDetection's outputs) -> per frame
        dict(n, coef [n,4], src [n], pt_off [n+1], points [m,3]) (+ state / nvox / info per detector plane with debug=True)."""
        d = np.ascontiguousarray(depth, np.uint16); lab = np.ascontiguousarray(labels, np.int32); pl = np.ascontiguousarray(planes, np.float64)
        npl = np.ascontiguousarray(n_planes, np.int32)
        B, H, W = d.shape
        PS, MP = self.pl_stride, self.max_points
        assert pl.shape == (B, PS, 8), pl.shape
        n = np.zeros(B, np.int32); coef = np.zeros((B, PS, 4), np.float32); src = np.zeros((B, PS), np.int32); off = np.zeros((B, PS + 1), np.int32)
        pts = np.zeros((B, MP, 3), np.float32)
        state = np.zeros((B, PS), np.int32); nvox = np.zeros((B, PS), np.int32); info = np.zeros((B, PS, 12), np.int32)
        rc = self.L.planar_plane_clouds_compute(self.h, d.ctypes.data, B, W, W * H, K[0], K[1], K[2], K[3], np.float32(depth_factor), lab.ctypes.data, pl.ctypes.data,
                                                npl.ctypes.data, float(dist_th), np.float32(leaf), n.ctypes.data, coef.ctypes.data, src.ctypes.data, off.ctypes.data,
                                                pts.ctypes.data, state.ctypes.data if debug else None, nvox.ctypes.data if debug else None,
                                                info.ctypes.data if debug else None)
        status = np.zeros(B, np.int32)
        try:
            check(rc)
        except PlanarError as e:                      # PLANAR_ECAPACITY: the per-frame codes say which frames, and why (planar_plane_clouds_last_status)
            if e.code != -4:
                raise
            check(self.L.planar_plane_clouds_last_status(self.h, B, status.ctypes.data))
            # code 3: the frame's planes together hold more than max_points voxels (pcl::VoxelGrid has no cap): only THOSE frames go plane by plane; anything else is an error
            if not retry_per_plane or ((status != 0) & (status != 3)).any():
                raise
        out = []
        for b in range(B):
            if status[b] == 3:
                out.append(self._per_plane(d[b:b + 1], lab[b:b + 1], pl[b:b + 1], npl[b:b + 1], dist_th, leaf, K, depth_factor, debug))
                continue
            k = int(n[b])
            r = dict(n=k, coef=coef[b, :k].copy(), src=src[b, :k].copy(), pt_off=off[b, :k + 1].copy(), points=pts[b, :off[b, k]].copy(), dropped=[])
            if debug:
                P = int(npl[b])
                r.update(state=state[b, :P].copy(), nvox=nvox[b, :P].copy(), info=[_refit_info(info[b, i]) for i in range(P)])
            out.append(r)
        return out

    def _per_plane(self, d, lab, pl, npl, dist_th, leaf, K, depth_factor, debug=False):
        """One frame whose planes together overflow the voxel table: plane by plane through the plane window, results appended in plane order (the sequence of the
        loop of Frame::ComputePlanes, src/Frame.cc:655-692).  A single plane of more than max_points voxels is dropped (listed in the result's `dropped`).  Same keys as
        the one-pass result (with debug: state / nvox / info per detector plane; a dropped plane has state -2)."""
        H, W = d.shape[1:]
        PS, MP = self.pl_stride, self.max_points
        P = int(npl[0])
        n = np.zeros(1, np.int32); coef = np.zeros((1, PS, 4), np.float32); src = np.zeros((1, PS), np.int32); off = np.zeros((1, PS + 1), np.int32)
        pts = np.zeros((1, MP, 3), np.float32)
        state = np.zeros((1, PS), np.int32); nvox = np.zeros((1, PS), np.int32); info = np.zeros((1, PS, 12), np.int32); st1 = np.zeros(1, np.int32)
        res = dict(n=0, coef=[], src=[], pt_off=[0], points=[], dropped=[], state=np.full(P, -2, np.int32), nvox=np.zeros(P, np.int32), info=[None] * P)
        try:
            for i in range(P):
                check(self.L.planar_plane_clouds_set_plane_window(self.h, i, 1))
                rc = self.L.planar_plane_clouds_compute(self.h, d.ctypes.data, 1, W, W * H, K[0], K[1], K[2], K[3], np.float32(depth_factor), lab.ctypes.data, pl.ctypes.data,
                                                        npl.ctypes.data, float(dist_th), np.float32(leaf), n.ctypes.data, coef.ctypes.data, src.ctypes.data, off.ctypes.data,
                                                        pts.ctypes.data, state.ctypes.data if debug else None, nvox.ctypes.data if debug else None,
                                                        info.ctypes.data if debug else None)
                try:
                    check(rc)
                except PlanarError as e:
                    if e.code != -4:
                        raise
                    check(self.L.planar_plane_clouds_last_status(self.h, 1, st1.ctypes.data))
                    if st1[0] != 3:
                        raise
                    res["dropped"].append(i); continue
                if debug:
                    res["state"][i] = state[0, i]; res["nvox"][i] = nvox[0, i]; res["info"][i] = _refit_info(info[0, i])
                if int(n[0]) == 1:
                    res["n"] += 1; res["coef"].append(coef[0, 0].copy()); res["src"].append(i); res["points"].append(pts[0, :off[0, 1]].copy())
                    res["pt_off"].append(res["pt_off"][-1] + int(off[0, 1]))
        finally:
            check(self.L.planar_plane_clouds_set_plane_window(self.h, 0, -1))
        k = res["n"]
        r = dict(n=k, coef=np.array(res["coef"], np.float32).reshape(k, 4), src=np.array(res["src"], np.int32), pt_off=np.array(res["pt_off"], np.int32),
                 points=np.concatenate(res["points"] + [np.zeros((0, 3), np.float32)]), dropped=res["dropped"])
        if debug:
            r.update(state=res["state"], nvox=res["nvox"], info=res["info"])
        return r

    def compute_dev(self, d_depth, d_labels, d_planes, d_n_planes, B, d_n_out, d_coef, d_src, d_pt_off, d_points, d_status, dist_th=0.05, leaf=0.1,
                    K=(535.4, 539.2, 320.1, 247.6), depth_factor=1.0 / 5000.0):
        check(self.L.planar_plane_clouds_compute_dev(self.h, d_depth, B, self.width, self.width * self.height, K[0], K[1], K[2], K[3], np.float32(depth_factor), d_labels,
                                                     d_planes, d_n_planes, float(dist_th), np.float32(leaf), d_n_out, d_coef, d_src, d_pt_off, d_points, d_status, None, None, None))

    def refit(self, planes, clouds, dist_th=0.05):
        """Frame::MaxPointDistanceFromPlane on given clouds: planes [q,4], clouds: list of [n,3] -> (state [q], planes [q,4], info list)."""
        planes = np.array(planes, np.float32).reshape(-1, 4).copy()
        q = len(planes)
        off = np.zeros(q + 1, np.int32)
        off[1:] = np.cumsum([len(c) for c in clouds])
        pts = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32).reshape(-1, 3) for c in clouds] + [np.zeros((1, 3), np.float32)]), np.float32)
        state = np.zeros(q, np.int32); info = np.zeros((q, 12), np.int32)
        check(self.L.planar_plane_refit(self.h, q, pts.ctypes.data, off.ctypes.data, float(dist_th), planes.ctypes.data, state.ctypes.data, info.ctypes.data))
        return state, planes, [_refit_info(info[i]) for i in range(q)]

    def merge(self, Twc, frame_points, map_points, leaf=0.1):
        """The cloud half of MapPlane::UpdateCoefficientsAndPoints: -> merged voxel centroids [m,3]."""
        T = np.ascontiguousarray(Twc, np.float64).reshape(16)
        f = np.ascontiguousarray(frame_points, np.float32).reshape(-1, 3); m = np.ascontiguousarray(map_points, np.float32).reshape(-1, 3)
        out = np.zeros((self.max_points, 3), np.float32); n = C.c_int32()
        check(self.L.planar_merge_plane_points(self.h, T.ctypes.data, f.ctypes.data if len(f) else None, len(f), m.ctypes.data if len(m) else None, len(m), np.float32(leaf),
                                               out.ctypes.data, self.max_points, C.byref(n)))
        return out[:n.value].copy()


def flag_matched_plane_points(Tcw, coef, matched, n_planes, xw, ctx: Context | None = None):
    """Map::FlagMatchedPlanePoints for B frames: Tcw [B,16], coef [B,S,4], matched [B,S] u8, n_planes [B], xw [n,3] (shared) or [B,n,3] -> (flags [B,n] u8, n_matches [B])."""
    L = lib(); ctx = ctx or Context(0)
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(-1, 16); B = len(Tcw)
    coef = np.ascontiguousarray(coef, np.float32); matched = np.ascontiguousarray(matched, np.uint8); npl = np.ascontiguousarray(n_planes, np.int32)
    xw = np.ascontiguousarray(xw, np.float32)
    shared = xw.ndim == 2
    npts = xw.shape[-2]
    flags = np.zeros((B, npts), np.uint8); nm = np.zeros(B, np.int32)
    check(L.planar_flag_matched_plane_points(ctx.h, B, Tcw.ctypes.data, coef.ctypes.data, matched.ctypes.data, npl.ctypes.data, coef.shape[1], xw.ctypes.data, npts, int(shared),
                                             flags.ctypes.data, nm.ctypes.data))
    return flags, nm
