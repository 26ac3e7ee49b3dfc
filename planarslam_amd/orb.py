"""Host-side mirror of Planar_SLAM::ORBextractor (reference include/ORBextractor.h:45-112) over the
C ABI.  Same constructor arguments and getters; operator() becomes __call__ and works on a batch."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import KP_DTYPE, Context, OrbParams, check, lib


class ORBextractor:
    HARRIS_SCORE = 0
    FAST_SCORE = 1

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int,
                 width: int = 640, height: int = 480, max_batch: int = 1, ctx: Context | None = None):
        self.L = lib()
        self.ctx = ctx or Context(0)
        self.width, self.height, self.max_batch = width, height, max_batch
        self.nlevels = nlevels
        self.scaleFactor = scaleFactor
        p = OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        h = C.c_void_p()
        check(self.L.planar_orb_create(self.ctx.h, C.byref(p), width, height, max_batch, C.byref(h)))
        self.h = h
        self.kp_cap = check(self.L.planar_orb_max_keypoints(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_orb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- reference getters (include/ORBextractor.h:63-83) ---
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def _factors(self):
        out = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        check(self.L.planar_orb_get_scale_factors(self.h, *[o.ctypes.data for o in out]))
        return out

    def GetScaleFactors(self):
        return self._factors()[0]

    def GetInverseScaleFactors(self):
        return self._factors()[1]

    def GetScaleSigmaSquares(self):
        return self._factors()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._factors()[3]

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        check(self.L.planar_orb_features_per_level(self.h, out.ctypes.data))
        return out

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        check(self.L.planar_orb_level_size(self.h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    # --- operator() (src/ORBextractor.cc:1043) ---
    def __call__(self, image: np.ndarray, mask=None):
        """image: (H,W) or (B,H,W) uint8.  Returns (keypoints, descriptors) for a single image or a
        list of such pairs for a batch.  `mask` is ignored, as in the reference."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8:
            raise TypeError("image must be CV_8UC1 (uint8)")      # reference asserts type()==CV_8UC1 (:1050)
        single = image.ndim == 2
        batch = np.ascontiguousarray(image[None] if single else image)
        B, H, W = batch.shape
        if (W, H) != (self.width, self.height):
            raise ValueError(f"extractor was created for {self.width}x{self.height}, got {W}x{H}")
        kps = np.zeros((B, self.kp_cap), KP_DTYPE)
        desc = np.zeros((B, self.kp_cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        check(self.L.planar_orb_extract(self.h, batch.ctypes.data, B, W, W * H, kps.ctypes.data, desc.ctypes.data,
                                        n.ctypes.data))
        res = [(kps[b, :n[b]].copy(), desc[b, :n[b]].copy()) for b in range(B)]
        return res[0] if single else res

    def extract_dev(self, d_gray, d_kps, d_desc, d_n, B, pitch=None, frame_stride=None):
        """Enqueue on the context stream; all arguments are device pointers (ints)."""
        pitch = pitch or self.width
        frame_stride = frame_stride or pitch * self.height
        check(self.L.planar_orb_extract_dev(self.h, d_gray, B, pitch, frame_stride, d_kps, d_desc, d_n))

    # --- stage read-back (tests, mvImagePyramid) ---
    def read_level(self, frame, level, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        f = self.L.planar_orb_read_blurred if blurred else self.L.planar_orb_read_level
        check(f(self.h, frame, level, out.ctypes.data))
        return out

    def read_candidates(self, frame, level, cap=200000):
        out = np.zeros((cap, 3), np.int32)
        n = check(self.L.planar_orb_read_candidates(self.h, frame, level, out.ctypes.data, cap))
        return out[:n].copy()

    # --- per-launch HIP-event timing (bench.py roofline leg) ---
    def set_profiling(self, enable: bool):
        check(self.L.planar_orb_set_profiling(self.h, int(enable)))

    def get_profile(self):
        """Returns ({kernel name: (total_ms, launches)}, calls) since profiling was enabled; resets."""
        n = check(self.L.planar_orb_profile_num_launches(self.h))
        ms = np.zeros(n, np.float64)
        calls = C.c_int64()
        check(self.L.planar_orb_get_profile(self.h, ms.ctypes.data, C.byref(calls)))
        out = {}
        for i in range(n):
            name = self.L.planar_orb_profile_launch_name(self.h, i).decode()
            t, k = out.get(name, (0.0, 0))
            out[name] = (t + float(ms[i]), k + calls.value)
        return out, calls.value
