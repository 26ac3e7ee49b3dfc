"""Host-side mirrors of the reference matchers over the C ABI (batched over frame pairs).

    ORBmatcher.MatchORBPoints        reference src/ORBmatcher.cc:1332   (include/ORBmatcher.h:53)
    ORBmatcher.DescriptorDistance    reference src/ORBmatcher.cc:1712   (via hamming_knn)
    LSDmatcher.SearchByDescriptor    reference src/LSDmatcher.cpp:242   (include/LSDmatcher.h)
Descriptor arrays are [B, stride, 32] uint8 with per-pair row counts n[B]."""
from __future__ import annotations

import numpy as np

from ._lib import Context, check, lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def hamming_knn(q, nq, t, nt, k=1, ctx: Context | None = None):
    """cv::BFMatcher(NORM_HAMMING).match (k=1) / knnMatch(k=2).  Returns idx, dist [B, q_stride, k]."""
    ctx = ctx or Context(0)
    q, t, nq, nt = _c(q, np.uint8), _c(t, np.uint8), _c(nq, np.int32), _c(nt, np.int32)
    B, qs, ts = q.shape[0], q.shape[1], t.shape[1]
    idx = np.zeros((B, qs, k), np.int32)
    dist = np.zeros((B, qs, k), np.int32)
    check(lib().planar_hamming_knn(ctx.h, q.ctypes.data, nq.ctypes.data, qs, t.ctypes.data, nt.ctypes.data, ts, B, k,
                                   idx.ctypes.data, dist.ctypes.data))
    return idx, dist


def distinctive_descriptors(observations, ctx: Context | None = None):
    """MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:259-324) for a batch of map points: observations = list of [n_p, 32] uint8 arrays (the
    descriptors of the point's non-bad observing key frames, in observation-map order) -> (best [P] index per point, -1 = none; median [P])."""
    ctx = ctx or Context(0)
    P = len(observations)
    off = np.zeros(P + 1, np.int32)
    off[1:] = np.cumsum([len(o) for o in observations])
    desc = np.ascontiguousarray(np.concatenate([np.asarray(o, np.uint8).reshape(-1, 32) for o in observations] + [np.zeros((1, 32), np.uint8)]), np.uint8)
    best = np.zeros(P, np.int32); med = np.zeros(P, np.int32)
    check(lib().planar_distinctive_descriptors(ctx.h, P, desc.ctypes.data, off.ctypes.data, best.ctypes.data, med.ctypes.data))
    return best, med


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True, ctx: Context | None = None):
        self.mfNNratio, self.mbCheckOrientation = nnratio, checkOri
        self.ctx = ctx or Context(0)

    def MatchORBPoints(self, cur_desc, n_cur, last_desc, n_last, last_has_mp, last_outlier, cur_match=None):
        """Returns (cur_match [B, cur_stride] int32, npair [B]).  cur_match[q] = last-frame keypoint index whose
        MapPoint the reference copies into CurrentFrame.mvpMapPoints[q]; untouched entries keep their input value
        (-1 if no array is given)."""
        cur_desc, last_desc = _c(cur_desc, np.uint8), _c(last_desc, np.uint8)
        n_cur, n_last = _c(n_cur, np.int32), _c(n_last, np.int32)
        has, outl = _c(last_has_mp, np.uint8), _c(last_outlier, np.uint8)
        B, cs, ls = cur_desc.shape[0], cur_desc.shape[1], last_desc.shape[1]
        m = np.full((B, cs), -1, np.int32) if cur_match is None else _c(cur_match, np.int32).copy()
        npair = np.zeros(B, np.int32)
        check(lib().planar_match_orb_points(self.ctx.h, cur_desc.ctypes.data, n_cur.ctypes.data, cs, last_desc.ctypes.data,
                                            n_last.ctypes.data, ls, has.ctypes.data, outl.ctypes.data, B, m.ctypes.data, npair.ctypes.data))
        return m, npair


class LSDmatcher:
    def __init__(self, ctx: Context | None = None):
        self.ctx = ctx or Context(0)

    def SearchByDescriptor(self, kf_desc, n_kf, cur_desc, n_cur, kf_has_ml):
        """Returns (cur_match [B, cur_stride] int32 (kf line index or -1), nmatches [B])."""
        kf_desc, cur_desc = _c(kf_desc, np.uint8), _c(cur_desc, np.uint8)
        n_kf, n_cur, has = _c(n_kf, np.int32), _c(n_cur, np.int32), _c(kf_has_ml, np.uint8)
        B, ks, cs = kf_desc.shape[0], kf_desc.shape[1], cur_desc.shape[1]
        m = np.zeros((B, cs), np.int32)
        nm = np.zeros(B, np.int32)
        check(lib().planar_lsd_search_by_descriptor(self.ctx.h, kf_desc.ctypes.data, n_kf.ctypes.data, ks, cur_desc.ctypes.data,
                                                    n_cur.ctypes.data, cs, has.ctypes.data, B, m.ctypes.data, nm.ctypes.data))
        return m, nm
