"""Host-side mirror of the reference's PlaneDetection (include/PlaneExtractor.h:36-56) over the C ABI:
readDepthImage + runPlaneDetection become one batched call."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, check, lib


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
