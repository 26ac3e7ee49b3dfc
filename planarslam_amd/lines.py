"""Host-side mirror of the reference line extractor over the C ABI.

    LineSegment.ExtractLineSegment   reference src/LSDextractor.cpp:12-39 (include/LSDextractor.h:344-352)
Batched: a [B, H, W] uint8 array stands for B calls.  No CPU fallback: everything runs in libplanar_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import KEYLINE_DTYPE, Context, check, lib

SEG_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("width", "<f8"), ("p", "<f8"), ("nfa", "<f8")])


class LineSegment:
    def __init__(self, width: int = 640, height: int = 480, max_batch: int = 1, ctx: Context | None = None, tie_order: int = 0, top_only: bool = False):
        self.ctx = ctx or Context(0)
        self.W, self.H, self.max_batch = width, height, max_batch
        h = C.c_void_p()
        check(lib().planar_lsd_create(self.ctx.h, width, height, max_batch, C.byref(h)))
        self.h = h
        w_, h_ = C.c_int(), C.c_int()
        check(lib().planar_lsd_scaled_size(self.h, C.byref(w_), C.byref(h_)))
        self.scaled = (w_.value, h_.value)
        # 0: pixels of one gradient bin in libstdc++ std::sort order (the reference library), 1: raster order
        check(lib().planar_lsd_set_tie_order(self.h, tie_order))
        self.tie_order = tie_order
        # the NFA stage only for the regions that can end among the kept key lines (same key lines; planar_lsd_set_top_only)
        check(lib().planar_lsd_set_top_only(self.h, int(top_only)))          # (2: test mode, every settled frame is redone)
        self.top_only = top_only

    def ExtractLineSegment(self, img, lsdNFeatures: int = 40):
        """img [B,H,W] or [H,W] uint8.  Returns (keylines [B,40] KEYLINE_DTYPE, ldesc [B,40,32] uint8,
        keylineFunctions [B,40,3] float64, n [B]); rows >= n[b] are unspecified."""
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim == 2:
            img = img[None]
        B, H, W = img.shape
        assert (H, W) == (self.H, self.W) and B <= self.max_batch
        kl = np.zeros((B, lsdNFeatures), KEYLINE_DTYPE)
        desc = np.zeros((B, lsdNFeatures, 32), np.uint8)
        eq = np.zeros((B, lsdNFeatures, 3), np.float64)
        n = np.zeros(B, np.int32)
        check(lib().planar_lsd_extract(self.h, img.ctypes.data, B, W, W * H, lsdNFeatures, kl.ctypes.data, desc.ctypes.data, eq.ctypes.data,
                                       n.ctypes.data))
        return kl, desc, eq, n

    def read_stage(self, frame: int, stage: int):
        """Diagnostics of the last call (see planar_lsd_read_stage)."""
        w, h = self.scaled
        if stage in (0, 1):
            out = np.zeros((h, w), np.float32 if stage == 0 else np.uint32)
        elif stage == 2:
            out = np.zeros(w * h, np.int32)
        elif stage == 3:
            out = np.zeros(lib().planar_lsd_max_segments(), SEG_DTYPE)
        elif stage == 5:
            out = np.zeros(10, np.int64)
        elif stage == 6:
            out = np.zeros(4, np.int64)
        else:
            out = np.zeros(1, np.int32)
        r = check(lib().planar_lsd_read_stage(self.h, frame, stage, out.ctypes.data, out.nbytes))
        return out[:min(r, len(out))] if stage in (2, 3) else out

    def close(self):
        if getattr(self, "h", None):
            lib().planar_lsd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def is_line_good(keylines, n_lines, depth, seeds, cam=(535.4, 539.2, 320.1, 247.6), depth_factor=1.0 / 5000.0, ctx: Context | None = None):
    """Frame::isLineGood (reference src/Frame.cc:189-267) for a batch: keylines [B,S] KEYLINE_DTYPE, n_lines [B], depth [B,H,W] u16, seeds [B] u32.
    Returns dict(depth_line [B,S] f32, lines3d [B,S,6], good [B,S] u8, direction [B,S,3], n_inliers [B,S], packed_dirs [B,S,3], n_good [B])."""
    import ctypes as C

    from ._lib import KEYLINE_DTYPE
    L = lib()
    ctx = ctx or Context(0)
    kl = np.ascontiguousarray(keylines, KEYLINE_DTYPE)
    B, S = kl.shape
    d = np.ascontiguousarray(depth, np.uint16)
    H, W = d.shape[1:]
    nl = np.ascontiguousarray(n_lines, np.int32); sd = np.ascontiguousarray(seeds, np.uint32)
    out = dict(depth_line=np.zeros((B, S), np.float32), lines3d=np.zeros((B, S, 6)), good=np.zeros((B, S), np.uint8), direction=np.zeros((B, S, 3)),
               n_inliers=np.zeros((B, S), np.int32), packed_dirs=np.zeros((B, S, 3)), n_good=np.zeros(B, np.int32))
    check(L.planar_is_line_good(ctx.h, B, kl.ctypes.data, nl.ctypes.data, S, d.ctypes.data, W, H, W, W * H, np.float32(depth_factor), cam[0], cam[1], cam[2], cam[3],
                                sd.ctypes.data, *[out[k].ctypes.data for k in ("depth_line", "lines3d", "good", "direction", "n_inliers", "packed_dirs", "n_good")]))
    return out
