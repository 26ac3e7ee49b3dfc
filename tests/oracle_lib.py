"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE (the CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        # PLANAR_ORACLE_LIB: bench.py's cpu_baseline leg points this at oracle/liboracle_fast.so (-O3 -march=native, built on the box it is timed on);
        # the tests always check against the -O2 build
        path = os.environ.get("PLANAR_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_orb_create.restype = C.c_void_p
        L.orc_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_orb_destroy.argtypes = [C.c_void_p]
        L.orc_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        for name in ("orc_orb_features_per_level", "orc_orb_umax", "orc_orb_num_candidates"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        L.orc_orb_scale.argtypes = [C.c_void_p, C.c_int]
        L.orc_orb_scale.restype = C.c_float
        L.orc_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_orb_get_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_orb_get_blurred.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_orb_get_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_orb_get_level_kps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gaussian7_s2_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_cv_round_f.argtypes = [C.c_float]
        L.orc_cv_round_d.argtypes = [C.c_double]
        _LIB = L
    return _LIB


class OrbOracle:
    """CPU oracle of ORBextractor (reference src/ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        self.L = lib()
        self.nlevels = nlevels
        self.h = self.L.orc_orb_create(nfeatures, scale, nlevels, ini, mn)
        self.cap = max(4 * nfeatures, 64)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_orb_destroy(self.h)
            self.h = None

    def extract(self, gray: np.ndarray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        H, W = gray.shape
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = self.L.orc_orb_extract(self.h, gray.ctypes.data, W, H, W, kps.ctypes.data, desc.ctypes.data, self.cap)
        assert n >= 0
        return kps[:n].copy(), desc[:n].copy()

    def features_per_level(self):
        return [self.L.orc_orb_features_per_level(self.h, l) for l in range(self.nlevels)]

    def level(self, l):
        w, h = C.c_int(), C.c_int()
        assert self.L.orc_orb_level_size(self.h, l, C.byref(w), C.byref(h)) == 0
        out = np.zeros((h.value, w.value), np.uint8)
        self.L.orc_orb_get_level(self.h, l, out.ctypes.data)
        return out

    def blurred(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orc_orb_level_size(self.h, l, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        if self.L.orc_orb_get_blurred(self.h, l, out.ctypes.data) != 0:
            return None
        return out

    def candidates(self, l):
        n = self.L.orc_orb_num_candidates(self.h, l)
        out = np.zeros((n, 3), np.int32)
        self.L.orc_orb_get_candidates(self.h, l, out.ctypes.data)
        return out

    def level_kps(self, l):
        out = np.zeros(self.cap, KP_DTYPE)
        n = self.L.orc_orb_get_level_kps(self.h, l, out.ctypes.data, self.cap)
        return out[:n].copy()


def ref_orb_path():
    return os.path.join(ORACLE_DIR, "_ref", "ref_orb")


def run_ref_orb(gray: np.ndarray, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
    """Run the REAL reference ORBextractor (oracle/_ref/ref_orb). Returns (kps, desc, pyramid)."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    H, W = gray.shape
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.raw"), os.path.join(d, "out.bin")
        gray.tofile(fin)
        subprocess.check_call([ref_orb_path(), fin, str(W), str(H), str(nfeatures), repr(float(scale)), str(nlevels),
                               str(ini), str(mn), fout])
        buf = open(fout, "rb").read()
    n = int(np.frombuffer(buf, "<i4", 1, 0)[0])
    off = 4
    kps = np.frombuffer(buf, KP_DTYPE, n, off).copy(); off += 28 * n
    desc = np.frombuffer(buf, np.uint8, 32 * n, off).reshape(n, 32).copy(); off += 32 * n
    nl = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
    pyr = []
    for _ in range(nl):
        w, h = np.frombuffer(buf, "<i4", 2, off); off += 8
        pyr.append(np.frombuffer(buf, np.uint8, int(w) * int(h), off).reshape(int(h), int(w)).copy()); off += int(w) * int(h)
    return kps, desc, pyr


# ---- pose optimisation oracle ----
class PoseParamsC(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("angle_info", C.c_double), ("distance_info", C.c_double), ("parallel_info", C.c_double),
                ("vertical_info", C.c_double), ("plane_chi", C.c_double), ("vp_chi", C.c_double)]


def pose_params(d):
    return PoseParamsC(d["fx"], d["fy"], d["cx"], d["cy"], d["bf"], d["angle_info"], d["distance_info"], d["parallel_info"],
                       d["vertical_info"], d["plane_chi"], d["vp_chi"])


def pose_optimize(batch, params, mode=0, rounds=4, its=10):
    """CPU oracle of Optimizer::PoseOptimization (mode 0) / TranslationOptimization (mode 1) on a synth.pose_batch()."""
    L = lib()
    B = len(batch["n_points"])
    MP, ML, MM = batch["pt_valid"].shape[1], batch["ln_valid"].shape[1], batch["pl_valid"].shape[1]
    res = dict(Tcw=np.zeros((B, 16), np.float32), pt_outlier=np.zeros((B, MP), np.uint8), ln_outlier=np.zeros((B, ML), np.uint8),
               pl_outlier=np.zeros((B, MM, 3), np.uint8), n_inliers=np.zeros(B, np.int32), lm_iters=np.zeros(B, np.int32),
               final_chi2=np.zeros(B, np.float64))
    prm = pose_params(params)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.orc_pose_optimize_batch.restype = C.c_int
    L.orc_pose_optimize_batch(
        C.c_int(B), C.c_int(MP), C.c_int(ML), C.c_int(MM), p(batch["n_points"]), p(batch["n_lines"]), p(batch["n_planes"]),
        p(batch["pt_valid"]), p(batch["pt_xw"]), p(batch["pt_obs"]), p(batch["pt_inv_sigma2"]), p(batch["ln_valid"]),
        p(batch["ln_obs"]), p(batch["ln_xw"]), p(batch["pl_meas"]), p(batch["pl_valid"]), p(batch["pl_world"]), p(batch["Tcw"]),
        C.byref(prm), C.c_int(mode), C.c_int(rounds), C.c_int(its), p(res["Tcw"]), p(res["pt_outlier"]), p(res["ln_outlier"]),
        p(res["pl_outlier"]), p(res["n_inliers"]), p(res["lm_iters"]), p(res["final_chi2"]))
    return res


# ---- matcher oracle ----
def bf_knn(q, t, k):
    L = lib()
    q, t = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((len(q), k), np.int32); dist = np.zeros((len(q), k), np.int32)
    L.orc_bf_knn(C.c_void_p(q.ctypes.data), len(q), C.c_void_p(t.ctypes.data), len(t), k, C.c_void_p(idx.ctypes.data), C.c_void_p(dist.ctypes.data))
    return idx, dist


def match_orb_points(cur, last, has_mp, outl, cur_match):
    L = lib()
    cur, last = np.ascontiguousarray(cur, np.uint8), np.ascontiguousarray(last, np.uint8)
    has_mp, outl = np.ascontiguousarray(has_mp, np.uint8), np.ascontiguousarray(outl, np.uint8)
    m = np.ascontiguousarray(cur_match, np.int32).copy()
    n = L.orc_match_orb_points(C.c_void_p(cur.ctypes.data), len(cur), C.c_void_p(last.ctypes.data), len(last), C.c_void_p(has_mp.ctypes.data),
                               C.c_void_p(outl.ctypes.data), C.c_void_p(m.ctypes.data))
    return m, n


def lsd_search_by_descriptor(kf, cur, has_ml):
    L = lib()
    kf, cur, has_ml = np.ascontiguousarray(kf, np.uint8), np.ascontiguousarray(cur, np.uint8), np.ascontiguousarray(has_ml, np.uint8)
    m = np.zeros(len(cur), np.int32)
    n = L.orc_lsd_search_by_descriptor(C.c_void_p(kf.ctypes.data), len(kf), C.c_void_p(cur.ctypes.data), len(cur), C.c_void_p(has_ml.ctypes.data),
                                       C.c_void_p(m.ctypes.data))
    return m, n


# ---- PEAC reference runner ----
def ref_peac_path():
    return os.path.join(ORACLE_DIR, "_ref", "ref_peac")


def run_ref_peac(depth: np.ndarray, fx=535.4, fy=539.2, cx=320.1, cy=247.6, factor=1.0 / 5000.0):
    """Run the REAL reference PlaneDetection (oracle/_ref/ref_peac). Returns (planes, labels[H,W])."""
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.raw"), os.path.join(d, "out.bin")
        depth.tofile(fin)
        subprocess.check_call([ref_peac_path(), fin, str(W), str(H), repr(fx), repr(fy), repr(cx), repr(cy), repr(float(np.float32(factor))), fout],
                              stdout=subprocess.DEVNULL)
        buf = open(fout, "rb").read()
    n = int(np.frombuffer(buf, "<i4", 1, 0)[0]); off = 4
    planes = []
    for _ in range(n):
        N = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
        v = np.frombuffer(buf, "<f8", 7, off); off += 56
        nv = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
        planes.append(dict(N=N, normal=v[:3].copy(), center=v[3:6].copy(), mse=float(v[6]), nverts=nv))
    labels = np.frombuffer(buf, "<i4", H * W, off).reshape(H, W).copy()
    return planes, labels


def peac_run(depth: np.ndarray, fx=535.4, fy=539.2, cx=320.1, cy=247.6, factor=1.0 / 5000.0, max_planes=64, want_blocks=False):
    """CPU oracle of PlaneDetection::readDepthImage + runPlaneDetection.  Returns (planes [n,8], labels[H,W]) (+ blocks)."""
    L = lib()
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    labels = np.zeros((H, W), np.int32)
    planes = np.zeros((max_planes, 8), np.float64)
    blocks = np.zeros(((H // 10) * (W // 10), 6), np.float64)
    L.orc_peac_run.restype = C.c_int
    n = L.orc_peac_run(C.c_void_p(depth.ctypes.data), W, H, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(factor),
                       C.c_void_p(labels.ctypes.data), C.c_void_p(planes.ctypes.data), max_planes, C.c_void_p(blocks.ctypes.data))
    if want_blocks:
        return planes[:min(n, max_planes)].copy(), labels, blocks
    return planes[:min(n, max_planes)].copy(), labels


def local_ba(prob, params, its1=5, its2=10):
    """CPU oracle of the numerical core of Optimizer::LocalBundleAdjustment on a synth.ba_problem()."""
    L = lib()
    K, NL, NE = len(prob["kf_fixed"]), len(prob["lm_type"]), len(prob["e_kf"])
    res = dict(kf_Tcw=np.zeros((K, 16), np.float32), lm=np.zeros((NL, 4), np.float64), e_outlier=np.zeros(NE, np.uint8))
    chi = C.c_double()
    prm = pose_params(params)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.orc_local_ba.restype = C.c_int
    res["lm_iters"] = L.orc_local_ba(C.c_int(K), p(prob["kf_Tcw"]), p(prob["kf_fixed"]), C.c_int(NL), p(prob["lm_type"]), p(prob["lm_init"]), C.c_int(NE),
                                     p(prob["e_kf"]), p(prob["e_lm"]), p(prob["e_type"]), p(prob["e_meas"]), p(prob["e_inv_sigma2"]), C.byref(prm),
                                     C.c_int(its1), C.c_int(its2), p(res["kf_Tcw"]), p(res["lm"]), p(res["e_outlier"]), C.byref(chi))
    res["chi2"] = chi.value
    return res


def ba_reduced_system(prob, params, lam=1e-3):
    """Oracle: reduced camera system (S, b, robust chi2) of the first LM linearisation of a (shard of a) BA problem."""
    L = lib()
    K, NL, NE = len(prob["kf_fixed"]), len(prob["lm_type"]), len(prob["e_kf"])
    npose = int((np.asarray(prob["kf_fixed"]) == 0).sum())
    S = np.zeros((6 * npose, 6 * npose)); b = np.zeros(6 * npose); chi = C.c_double()
    prm = pose_params(params)
    a = {k: np.ascontiguousarray(prob[k]) for k in ("kf_Tcw", "kf_fixed", "lm_type", "lm_init", "e_kf", "e_lm", "e_type", "e_meas", "e_inv_sigma2")}
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    L.orc_ba_reduced_system.restype = C.c_int
    L.orc_ba_reduced_system(C.c_int(K), p(a["kf_Tcw"]), p(a["kf_fixed"]), C.c_int(NL), p(a["lm_type"]), p(a["lm_init"]), C.c_int(NE), p(a["e_kf"]), p(a["e_lm"]),
                            p(a["e_type"]), p(a["e_meas"]), p(a["e_inv_sigma2"]), C.byref(prm), C.c_double(lam), p(S), p(b), C.byref(chi))
    return S, b, chi.value


# ---- guided matchers (oracle/guided_oracle.cpp); inputs are the dicts planarslam_amd.guided takes ----
def _g():
    from planarslam_amd import guided
    return guided


def search_by_projection_frame(cur, last, th, mono=False, check_orientation=True, cur_match=None):
    G = _g(); L = lib()
    fv, k1 = G.frame_view(cur); lv, k2 = G.last_frame_view(last)
    m = np.full((fv.B, fv.stride), -1, np.int32) if cur_match is None else np.ascontiguousarray(cur_match, np.int32).copy()
    nm = np.zeros(fv.B, np.int32)
    L.orc_search_by_projection_frame(C.byref(fv), C.byref(lv), C.c_float(th), int(mono), int(check_orientation), C.c_void_p(m.ctypes.data),
                                     C.c_void_p(nm.ctypes.data))
    return m, nm


def search_by_projection_map(frame, probes, th=1.0, nn_ratio=0.6, match=None):
    G = _g(); L = lib()
    fv, k1 = G.frame_view(frame); pv, k2 = G.map_probes(probes)
    m = np.full((fv.B, fv.stride), -1, np.int32) if match is None else np.ascontiguousarray(match, np.int32).copy()
    nm = np.zeros(fv.B, np.int32)
    L.orc_search_by_projection_map(C.byref(fv), C.byref(pv), C.c_float(th), C.c_float(nn_ratio), C.c_void_p(m.ctypes.data), C.c_void_p(nm.ctypes.data))
    return m, nm


def search_by_bow(kf, f, nn_ratio=0.7, check_orientation=True):
    L = lib()
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    p = lambda a: C.c_void_p(a.ctypes.data)
    a = [c(kf["n"], np.int32), c(kf["node"], np.int32), c(kf["usable"], np.uint8), c(kf["angle"], np.float32), c(kf["desc"], np.uint8)]
    b = [c(f["n"], np.int32), c(f["node"], np.int32), c(f["angle"], np.float32), c(f["desc"], np.uint8)]
    B, ks = a[1].shape; fs = b[1].shape[1]
    m = np.full((B, fs), -1, np.int32); nm = np.zeros(B, np.int32)
    L.orc_search_by_bow(B, p(a[0]), ks, p(a[1]), p(a[2]), p(a[3]), p(a[4]), p(b[0]), fs, p(b[1]), p(b[2]), p(b[3]), C.c_float(nn_ratio),
                        int(check_orientation), p(m), p(nm))
    return m, nm


def lsd_search_by_projection(lines, maplines, scale_factors, th=1.0, nn_ratio=0.6, match=None):
    from planarslam_amd._lib import KEYLINE_DTYPE
    L = lib()
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    p = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
    n, kl, ld = c(lines["n"], np.int32), c(lines["keylines"], KEYLINE_DTYPE), c(lines["ldesc"], np.uint8)
    bl = c(lines["blocked"], np.uint8) if lines.get("blocked") is not None else None
    mn, iv, pr = c(maplines["n"], np.int32), c(maplines["in_view"], np.uint8), c(maplines["proj"], np.float32)
    lv, vc, md, ob = c(maplines["level"], np.int32), c(maplines["view_cos"], np.float32), c(maplines["desc"], np.uint8), c(maplines["observed"], np.uint8)
    sf = c(scale_factors, np.float32)
    B, S = kl.shape; M = iv.shape[1]
    m = np.full((B, S), -1, np.int32) if match is None else c(match, np.int32).copy()
    nm = np.zeros(B, np.int32)
    L.orc_lsd_search_by_projection(B, p(n), S, p(kl), p(ld), p(bl), p(mn), M, p(iv), p(pr), p(lv), p(vc), p(md), p(ob), p(sf), len(sf), C.c_float(th),
                                   C.c_float(nn_ratio), p(m), p(nm))
    return m, nm


def plane_search_by_coefficients(frame, mapplanes, th=(0.1, 0.86, 0.08716, 0.9962), init=None):
    L = lib()
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    p = lambda a: C.c_void_p(a.ctypes.data)
    n, coef, T = c(frame["n"], np.int32), c(frame["coef"], np.float32), c(frame["Tcw"], np.float32)
    mn, mv, mc = c(mapplanes["n"], np.int32), c(mapplanes["valid"], np.uint8), c(mapplanes["coef"], np.float32)
    mnp, mp = c(mapplanes["npts"], np.int32), c(mapplanes["pts"], np.float32)
    B, S = coef.shape[:2]; M, P = mp.shape[-3], mp.shape[-2]
    out = [np.full((B, S), -1, np.int32) if init is None else c(init[i], np.int32).copy() for i in range(3)]
    nm = np.zeros(B, np.int32)
    tha = np.asarray(th, np.float32)
    L.orc_plane_search_by_coefficients(B, p(n), S, p(coef), p(T), int(bool(mapplanes.get("shared"))), p(mn), M, p(mv), p(mc), p(mnp), P, p(mp), p(tha),
                                       p(out[0]), p(out[1]), p(out[2]), p(nm))
    return out[0], out[1], out[2], nm


# ---- line extraction (oracle/lsd_oracle.cpp) ----
def lsd_detect(img, tie_order=1, want_stages=False):
    """Raw LSD segments of cv::LineSegmentDetector(LSD_REFINE_ADV).  Returns dict(xy [n,4] f32, wpn [n,3] f64, + stages)."""
    L = lib()
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    w, h = int(np.rint(W * 0.8)), int(np.rint(H * 0.8))
    cap = 8192
    xy = np.zeros((cap, 4), np.float32); wpn = np.zeros((cap, 3), np.float64)
    scaled = np.zeros((h, w), np.uint8); ang = np.zeros((h, w), np.float64); order = np.full(w * h, -1, np.int32)
    L.orc_lsd_detect.restype = C.c_int
    n = L.orc_lsd_detect(C.c_void_p(img.ctypes.data), W, H, W, int(tie_order), C.c_void_p(xy.ctypes.data), C.c_void_p(wpn.ctypes.data), cap,
                         C.c_void_p(scaled.ctypes.data) if want_stages else None, C.c_void_p(ang.ctypes.data) if want_stages else None,
                         C.c_void_p(order.ctypes.data) if want_stages else None)
    out = dict(xy=xy[:n].copy(), wpn=wpn[:n].copy())
    if want_stages:
        out.update(scaled=scaled, angles=ang, order=order)
    return out


def extract_line_segment(img, tie_order=1, max_lines=40):
    """LineSegment::ExtractLineSegment.  Returns (keylines [n], desc [n,32], eq [n,3], desc_float [n,72], n_detected)."""
    from planarslam_amd._lib import KEYLINE_DTYPE
    L = lib()
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    kl = np.zeros(max_lines, KEYLINE_DTYPE); desc = np.zeros((max_lines, 32), np.uint8); eq = np.zeros((max_lines, 3)); df = np.zeros((max_lines, 72), np.float32)
    nd = C.c_int()
    L.orc_extract_line_segment.restype = C.c_int
    n = L.orc_extract_line_segment(C.c_void_p(img.ctypes.data), W, H, W, int(tie_order), max_lines, C.c_void_p(kl.ctypes.data), C.c_void_p(desc.ctypes.data),
                                   C.c_void_p(eq.ctypes.data), C.c_void_p(df.ctypes.data), C.byref(nd))
    return kl[:n].copy(), desc[:n].copy(), eq[:n].copy(), df[:n].copy(), nd.value


def std_sort_desc(keys):
    L = lib()
    k = np.ascontiguousarray(keys, np.float32).copy(); perm = np.zeros(len(k), np.int32)
    L.orc_std_sort_desc(C.c_void_p(k.ctypes.data), C.c_void_p(perm.ctypes.data), len(k))
    return k, perm


# ---- the REAL reference matchers (oracle/_ref/ref_match: src/ORBmatcher.cc, src/PlaneMatcher.cpp) ----
BINARY_OVERRIDE = {}   # tests/test_adapters_gpu.py points "ref_match" / "ref_opt" at the harness built over include/planar_adapters.hpp


def ref_match_path():
    return BINARY_OVERRIDE.get("ref_match") or os.path.join(ORACLE_DIR, "_ref", "ref_match")


def _run_ref_match(mode, blocks, n_out):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            for a in blocks:
                a = np.ascontiguousarray(a)
                f.write(np.int64(a.nbytes).tobytes()); f.write(a.tobytes())
        subprocess.check_call([ref_match_path(), mode, fin, fout])
        buf = open(fout, "rb").read()
    outs, off = [], 0
    for _ in range(n_out):
        n = int(np.frombuffer(buf, "<i8", 1, off)[0]); off += 8
        outs.append(np.frombuffer(buf, "<i4", n // 4, off).copy()); off += n
    return outs


def _frame_blocks(fr, b):
    n = int(fr["n"][b])
    f32 = np.float32
    gw = f32(64) / f32(f32(fr["max_x"]) - f32(fr["min_x"])); gh = f32(48) / f32(f32(fr["max_y"]) - f32(fr["min_y"]))
    intr = np.array([fr["min_x"], fr["max_x"], fr["min_y"], fr["max_y"], gw, gh, fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["bf"], fr["b"]], np.float32)
    blocked = fr["blocked"][b, :n] if fr.get("blocked") is not None else np.zeros(n, np.uint8)
    return [fr["keys_un"][b, :n], fr["u_right"][b, :n].astype(np.float32), fr["desc"][b, :n], blocked.astype(np.uint8), intr,
            np.asarray(fr["scale_factors"], np.float32)]


def ref_search_by_projection_frame(cur, last, b, th, mono=False, check_orientation=True):
    """One frame pair through the real ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono). Returns (match[n], nmatches)."""
    nl = int(last["n"][b])
    blocks = [np.array([th, float(mono), float(check_orientation)], np.float32)] + _frame_blocks(cur, b) + [
        np.asarray(cur["Tcw"][b], np.float32), np.asarray(last["Tcw"][b], np.float32), last["usable"][b, :nl].astype(np.uint8), last["xw"][b, :nl].astype(np.float32),
        last["octave"][b, :nl].astype(np.int32), last["angle"][b, :nl].astype(np.float32), last["mp_desc"][b, :nl], last["mp_observed"][b, :nl].astype(np.uint8)]
    m, nm = _run_ref_match("proj_frame", blocks, 2)
    return m, int(nm[0])


def ref_search_by_projection_map(frame, probes, b, th, nn_ratio):
    npb = int(probes["n"][b])
    blocks = [np.array([th, nn_ratio], np.float32)] + _frame_blocks(frame, b) + [
        probes["in_view"][b, :npb].astype(np.uint8), probes["proj_x"][b, :npb].astype(np.float32), probes["proj_y"][b, :npb].astype(np.float32),
        probes["proj_xr"][b, :npb].astype(np.float32), probes["level"][b, :npb].astype(np.int32), probes["view_cos"][b, :npb].astype(np.float32),
        probes["desc"][b, :npb], probes["observed"][b, :npb].astype(np.uint8)]
    m, nm = _run_ref_match("proj_map", blocks, 2)
    return m, int(nm[0])


def ref_search_by_bow(kf, f, b, nn_ratio, check_orientation=True):
    nk, nf = int(kf["n"][b]), int(f["n"][b])
    blocks = [np.array([nn_ratio, float(check_orientation)], np.float32), kf["node"][b, :nk].astype(np.int32), kf["usable"][b, :nk].astype(np.uint8),
              kf["angle"][b, :nk].astype(np.float32), kf["desc"][b, :nk], f["node"][b, :nf].astype(np.int32), f["angle"][b, :nf].astype(np.float32), f["desc"][b, :nf]]
    m, nm = _run_ref_match("bow", blocks, 2)
    return m, int(nm[0])


def ref_match_orb_points(cur_desc, last_desc, last_has_mp, last_outlier):
    blocks = [np.zeros(1, np.float32), np.ascontiguousarray(cur_desc, np.uint8), np.ascontiguousarray(last_desc, np.uint8),
              np.asarray(last_has_mp, np.uint8), np.asarray(last_outlier, np.uint8)]
    m, nm = _run_ref_match("match_orb", blocks, 2)
    return m, int(nm[0])


def ref_plane_search(frame, mapplanes, b, th=(0.1, 0.86, 0.08716, 0.9962)):
    m = 0 if mapplanes.get("shared") else b
    npl, nmp = int(frame["n"][b]), int(mapplanes["n"][m])
    blocks = [np.asarray(th, np.float32), frame["coef"][b, :npl].astype(np.float32), np.asarray(frame["Tcw"][b], np.float32),
              mapplanes["valid"][m, :nmp].astype(np.uint8), mapplanes["coef"][m, :nmp].astype(np.float32), mapplanes["npts"][m, :nmp].astype(np.int32),
              mapplanes["pts"][m, :nmp].astype(np.float32)]
    a, v, p, nm = _run_ref_match("plane", blocks, 4)
    return a, v, p, int(nm[0])


def ref_lsd_search_by_projection(lines, maplines, b, scale_factors, th, nn_ratio):
    from planarslam_amd._lib import KEYLINE_DTYPE
    nl, nm = int(lines["n"][b]), int(maplines["n"][b])
    blocked = lines["blocked"][b, :nl] if lines.get("blocked") is not None else np.zeros(nl, np.uint8)
    blocks = [np.array([th, nn_ratio], np.float32), np.ascontiguousarray(lines["keylines"][b, :nl], KEYLINE_DTYPE), lines["ldesc"][b, :nl],
              blocked.astype(np.uint8), maplines["in_view"][b, :nm].astype(np.uint8), maplines["proj"][b, :nm].astype(np.float32),
              maplines["level"][b, :nm].astype(np.int32), maplines["view_cos"][b, :nm].astype(np.float32), maplines["desc"][b, :nm],
              maplines["observed"][b, :nm].astype(np.uint8), np.asarray(scale_factors, np.float32)]
    m, n = _run_ref_match("lsd_proj", blocks, 2)
    return m, int(n[0])


def ref_lsd_search_by_descriptor(kf_desc, cur_desc, kf_has_ml):
    blocks = [np.zeros(1, np.float32), np.ascontiguousarray(kf_desc, np.uint8), np.ascontiguousarray(cur_desc, np.uint8), np.asarray(kf_has_ml, np.uint8)]
    m, n = _run_ref_match("lsd_desc", blocks, 2)
    return m, int(n[0])


def is_in_frustum_points(frame, mp, log_scale_factor, n_levels, limit=0.5):
    G = _g(); L = lib()
    fv, keep = G.frame_view(frame)
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    p = lambda a: C.c_void_p(a.ctypes.data)
    a = [c(mp["n"], np.int32), c(mp["valid"], np.uint8), c(mp["xw"], np.float32), c(mp["normal"], np.float32), c(mp["min_dist"], np.float32), c(mp["max_dist"], np.float32)]
    B, S = a[1].shape
    out = dict(in_view=np.zeros((B, S), np.uint8), proj_x=np.zeros((B, S), np.float32), proj_y=np.zeros((B, S), np.float32), proj_xr=np.zeros((B, S), np.float32),
               level=np.zeros((B, S), np.int32), view_cos=np.zeros((B, S), np.float32))
    L.orc_is_in_frustum_points(C.byref(fv), C.c_float(log_scale_factor), int(n_levels), p(a[0]), S, p(a[1]), p(a[2]), p(a[3]), p(a[4]), p(a[5]), C.c_float(limit),
                               p(out["in_view"]), p(out["proj_x"]), p(out["proj_y"]), p(out["proj_xr"]), p(out["level"]), p(out["view_cos"]))
    return out


def _inv_sigma2(frame, n_levels):
    sf = np.asarray(frame["scale_factors"], np.float32)[:n_levels]
    return np.ascontiguousarray(np.float32(1.0) / (sf * sf), np.float32)


def fuse_search(kf, mp, th, log_scale_factor, n_levels, shared=False, inv_level_sigma2=None):
    """oracle/guided_oracle.cpp fuse_search = ORBmatcher::Fuse(pKF, vpMapPoints, th), search half.  Returns (fuse_idx, fuse_dist, n_fused)."""
    G = _g(); L = lib()
    fv, keep = G.frame_view(kf)
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    p = lambda a: C.c_void_p(a.ctypes.data)
    inv = _inv_sigma2(kf, n_levels) if inv_level_sigma2 is None else c(inv_level_sigma2, np.float32)
    a = [c(mp["n"], np.int32), c(mp["usable"], np.uint8), c(mp["xw"], np.float32), c(mp["normal"], np.float32), c(mp["min_dist"], np.float32), c(mp["max_dist"], np.float32),
         c(mp["desc"], np.uint8)]
    S = a[1].shape[-1]
    idx = np.full((fv.B, S), -1, np.int32); dist = np.full((fv.B, S), 256, np.int32); nf = np.zeros(fv.B, np.int32)
    L.orc_fuse_search(C.byref(fv), p(inv), C.c_float(log_scale_factor), int(n_levels), p(a[0]), S, int(shared), p(a[1]), p(a[2]), p(a[3]), p(a[4]), p(a[5]), p(a[6]),
                      C.c_float(th), p(idx), p(dist), p(nf))
    return idx, dist, nf


def ref_fuse(kf, mp, b, th, log_scale_factor, n_levels, shared=False, kf_state=None, kf_obs=None):
    """One key frame through the REAL ORBmatcher::Fuse (oracle/_ref/ref_match, mode fuse).  Returns (fuse_idx [n], nFused)."""
    pb = 0 if shared else b
    npb = int(mp["n"][pb]); nk = int(kf["n"][b])
    state = np.zeros(nk, np.uint8) if kf_state is None else np.asarray(kf_state, np.uint8)[:nk]
    kobs = np.zeros(nk, np.int32) if kf_obs is None else np.asarray(kf_obs, np.int32)[:nk]
    blocks = [np.array([th, log_scale_factor, float(n_levels)], np.float32)] + _frame_blocks(kf, b) + [
        _inv_sigma2(kf, n_levels), np.asarray(kf["Tcw"][b], np.float32), mp["usable"][pb, :npb].astype(np.uint8), mp["xw"][pb, :npb].astype(np.float32),
        mp["normal"][pb, :npb].astype(np.float32), mp["min_dist"][pb, :npb].astype(np.float32), mp["max_dist"][pb, :npb].astype(np.float32), mp["desc"][pb, :npb],
        state, kobs, np.asarray(mp.get("observations", np.ones_like(mp["usable"], dtype=np.int32))[pb, :npb], np.int32)]
    idx, nf = _run_ref_match("fuse", blocks, 2)
    return idx, int(nf[0])


def lsd_fuse_search(kf, lines, ml, th, log_scale_factor, n_levels, shared=False):
    """oracle/guided_oracle.cpp lsd_fuse_search = LSDmatcher::Fuse(pKF, vpMapLines, th), search half.  Returns (fuse_idx, fuse_dist, n_fused)."""
    from planarslam_amd._lib import KEYLINE_DTYPE
    G = _g(); L = lib()
    fv, keep = G.pose_view(kf)
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    p = lambda a: C.c_void_p(a.ctypes.data)
    a = [c(lines["n"], np.int32), c(lines["keylines"], KEYLINE_DTYPE), c(lines["ldesc"], np.uint8), c(ml["n"], np.int32), c(ml["usable"], np.uint8), c(ml["xw6"], np.float64),
         c(ml["normal"], np.float64), c(ml["min_dist"], np.float32), c(ml["max_dist"], np.float32), c(ml["desc"], np.uint8)]
    S = a[4].shape[-1]
    idx = np.full((fv.B, S), -1, np.int32); dist = np.full((fv.B, S), 2 ** 31 - 1, np.int32); nf = np.zeros(fv.B, np.int32)
    L.orc_lsd_fuse_search(C.byref(fv), C.c_float(log_scale_factor), int(n_levels), p(a[0]), a[1].shape[1], p(a[1]), p(a[2]), p(a[3]), S, int(shared), p(a[4]), p(a[5]), p(a[6]),
                          p(a[7]), p(a[8]), p(a[9]), C.c_float(th), p(idx), p(dist), p(nf))
    return idx, dist, nf


def ref_lsd_fuse(kf, lines, ml, b, th, log_scale_factor, n_levels, kf_state=None, kf_obs=None):
    """One key frame through the REAL LSDmatcher::Fuse (oracle/_ref/ref_match, mode lsd_fuse).  Returns (fuse_idx [n], nFused)."""
    from planarslam_amd._lib import KEYLINE_DTYPE
    nl, nm = int(lines["n"][b]), int(ml["n"][b])
    state = np.zeros(nl, np.uint8) if kf_state is None else np.asarray(kf_state, np.uint8)[:nl]
    kobs = np.zeros(nl, np.int32) if kf_obs is None else np.asarray(kf_obs, np.int32)[:nl]
    intr = np.array([kf["min_x"], kf["max_x"], kf["min_y"], kf["max_y"], kf["fx"], kf["fy"], kf["cx"], kf["cy"], kf["bf"]], np.float32)
    blocks = [np.array([th, log_scale_factor], np.float32), np.ascontiguousarray(lines["keylines"][b, :nl], KEYLINE_DTYPE), lines["ldesc"][b, :nl], intr,
              np.asarray(kf["scale_factors"], np.float32)[:n_levels], np.asarray(kf["Tcw"][b], np.float32), ml["usable"][b, :nm].astype(np.uint8),
              ml["xw6"][b, :nm].astype(np.float64), ml["normal"][b, :nm].astype(np.float64), ml["min_dist"][b, :nm].astype(np.float32),
              ml["max_dist"][b, :nm].astype(np.float32), ml["desc"][b, :nm], state, kobs, np.asarray(ml["observations"][b, :nm], np.int32)]
    idx, nf = _run_ref_match("lsd_fuse", blocks, 2)
    return idx, int(nf[0])


def is_in_frustum_lines(frame, ml, log_scale_factor, limit=0.5):
    G = _g(); L = lib()
    fv, keep = G.frame_view(frame)
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    p = lambda a: C.c_void_p(a.ctypes.data)
    a = [c(ml["n"], np.int32), c(ml["valid"], np.uint8), c(ml["xw6"], np.float64), c(ml["normal"], np.float64), c(ml["min_dist"], np.float32), c(ml["max_dist"], np.float32)]
    B, S = a[1].shape
    out = dict(in_view=np.zeros((B, S), np.uint8), proj=np.zeros((B, S, 4), np.float32), level=np.zeros((B, S), np.int32), view_cos=np.zeros((B, S), np.float32))
    L.orc_is_in_frustum_lines(C.byref(fv), C.c_float(log_scale_factor), p(a[0]), S, p(a[1]), p(a[2]), p(a[3]), p(a[4]), p(a[5]), C.c_float(limit),
                              p(out["in_view"]), p(out["proj"]), p(out["level"]), p(out["view_cos"]))
    return out


def ref_lsd_path():
    return os.path.join(ORACLE_DIR, "_ref", "ref_lsd")


def run_ref_lsd(gray, tie_order=1):
    """The REAL LineSegment::ExtractLineSegment (src/LSDextractor.cpp) over the restated LSD / LBD.  Returns (keylines, ldesc, eq)."""
    from planarslam_amd._lib import KEYLINE_DTYPE
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.raw"), os.path.join(d, "out.bin")
        gray.tofile(fin)
        subprocess.check_call([ref_lsd_path(), fin, str(W), str(H), str(int(tie_order)), fout])
        buf = open(fout, "rb").read()
    n = int(np.frombuffer(buf, "<i4", 1, 0)[0]); off = 4
    kl = np.frombuffer(buf, KEYLINE_DTYPE, n, off).copy(); off += 68 * n
    desc = np.frombuffer(buf, np.uint8, 32 * n, off).reshape(n, 32).copy(); off += 32 * n
    eq = np.frombuffer(buf, "<f8", 3 * n, off).reshape(n, 3).copy()
    return kl, desc, eq


def track_manhattan_frame(R_last, normals, lines):
    """Tracking::TrackManhattanFrame (src/Tracking.cc:963) for one frame -> dict(R, member, info, density)."""
    L = lib()
    L.orc_track_manhattan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    R_last = np.ascontiguousarray(R_last, np.float32)
    normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    lines = np.ascontiguousarray(lines, np.float64).reshape(-1, 3)
    n, nl = len(normals), len(lines)
    R = np.zeros((3, 3), np.float32); member = np.zeros(n + nl, np.uint8); info = np.zeros(8, np.int32); dens = np.zeros(3, np.float32)
    L.orc_track_manhattan(R_last.ctypes.data, normals.ctypes.data, n, lines.ctypes.data, nl, R.ctypes.data, member.ctypes.data,
                          info.ctypes.data, dens.ctypes.data)
    return dict(R=R, member=member, info=info, density=dens)


# ---- REAL reference optimiser (oracle/_ref/ref_opt: src/Optimizer.cc + vendored g2o + g2oAddition over the mini-Eigen stand-in) ----
def ref_opt_path():
    return BINARY_OVERRIDE.get("ref_opt") or os.path.join(ORACLE_DIR, "_ref", "ref_opt")


def _cam_cfg(params):
    cam = np.array([params["fx"], params["fy"], params["cx"], params["cy"], params["bf"]], np.float32)
    cfg = np.array([params["angle_info"], params["distance_info"], params["parallel_info"], params["vertical_info"], params["plane_chi"],
                    params["vp_chi"]], np.float64)
    return cam, cfg


def run_ref_pose(batch, params, mode=0):
    """The reference's own Optimizer::PoseOptimization (mode 0) / TranslationOptimization (mode 1) on a synth.pose_batch()."""
    B = len(batch["n_points"])
    MP, ML, MM = batch["pt_valid"].shape[1], batch["ln_valid"].shape[1], batch["pl_valid"].shape[1]
    cam, cfg = _cam_cfg(params)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([B, MP, ML, MM, mode], np.int32).tobytes()); f.write(cam.tobytes()); f.write(cfg.tobytes())
            for k, dt in (("n_points", np.int32), ("n_lines", np.int32), ("n_planes", np.int32), ("pt_valid", np.uint8), ("pt_xw", np.float32),
                          ("pt_obs", np.float32), ("pt_inv_sigma2", np.float32), ("ln_valid", np.uint8), ("ln_obs", np.float64), ("ln_xw", np.float64),
                          ("pl_meas", np.float32), ("pl_valid", np.uint8), ("pl_world", np.float32), ("Tcw", np.float32)):
                f.write(np.ascontiguousarray(batch[k], dt).tobytes())
        subprocess.check_call([ref_opt_path(), "pose", fin, fout], stdout=subprocess.DEVNULL)
        buf = open(fout, "rb").read()
    res = dict(Tcw=np.zeros((B, 16), np.float32), pt_outlier=np.zeros((B, MP), np.uint8), ln_outlier=np.zeros((B, ML), np.uint8),
               pl_outlier=np.zeros((B, MM, 3), np.uint8), n_inliers=np.zeros(B, np.int32))
    off = 0
    for b in range(B):
        res["n_inliers"][b] = np.frombuffer(buf, "<i4", 1, off)[0]; off += 4
        res["Tcw"][b] = np.frombuffer(buf, "<f4", 16, off); off += 64
        res["pt_outlier"][b] = np.frombuffer(buf, np.uint8, MP, off); off += MP
        res["ln_outlier"][b] = np.frombuffer(buf, np.uint8, ML, off); off += ML
        res["pl_outlier"][b] = np.frombuffer(buf, np.uint8, MM * 3, off).reshape(MM, 3); off += MM * 3
    assert off == len(buf)
    return res


def run_ref_local_ba(prob, params, cur_kf):
    """The reference's own Optimizer::LocalBundleAdjustment(pKF = cur_kf) on a synth.ba_problem(lines_on_kf=cur_kf).
    Returns kf_Tcw [K,16] f32, lm [NL,4] f64 (as the map objects hold them afterwards) and the per-edge erase verdict."""
    K, NL, NE = len(prob["kf_fixed"]), len(prob["lm_type"]), len(prob["e_kf"])
    cam, cfg = _cam_cfg(params)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([K, NL, NE, cur_kf], np.int32).tobytes()); f.write(cam.tobytes()); f.write(cfg.tobytes())
            for k, dt in (("kf_Tcw", np.float32), ("kf_fixed", np.uint8), ("lm_type", np.uint8), ("lm_init", np.float64), ("e_kf", np.int32),
                          ("e_obs_kf", np.int32), ("e_lm", np.int32), ("e_type", np.uint8), ("e_meas", np.float64), ("e_inv_sigma2", np.float32)):
                f.write(np.ascontiguousarray(prob[k], dt).tobytes())
        subprocess.check_call([ref_opt_path(), "ba", fin, fout], stdout=subprocess.DEVNULL)
        buf = open(fout, "rb").read()
    off = 0
    kf = np.frombuffer(buf, "<f4", K * 16, off).reshape(K, 16).copy(); off += K * 64
    lm = np.frombuffer(buf, "<f8", NL * 4, off).reshape(NL, 4).copy(); off += NL * 32
    er = np.frombuffer(buf, np.uint8, NE, off).copy(); off += NE
    assert off == len(buf)
    return dict(kf_Tcw=kf, lm=lm, e_outlier=er)


def _edge_cases(n, seed):
    """n seeded (pose, point, observation, line, map plane, measured plane) tuples for the edge-level checks."""
    from planarslam_amd import synth
    b = synth.pose_batch(B=n, n_points=4, n_lines=2, n_planes=1, seed=seed)
    return dict(Tcw=b["Tcw"].copy(), X=b["pt_xw"][:, 0].astype(np.float64), obs=np.abs(b["pt_obs"][:, 0]).astype(np.float64),
                lobs=b["ln_obs"][:, 0].copy(), pw=b["pl_world"][:, 0, 0].copy(), pm=b["pl_meas"][:, 0].copy())


def run_ref_edges(cases, params):
    """computeError() / chi2() / linearizeOplus() of the reference's 12 pose-only edge classes (oracle/_ref/ref_opt edges) ->
    [n, 12, 22] (err[3], chi2, J[3][6]) and the pose after VertexSE3Expmap::oplus of a seeded update [n, 4, 4]."""
    n = len(cases["Tcw"])
    cam, _ = _cam_cfg(params)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.int32(n).tobytes()); f.write(cam.tobytes())
            for i in range(n):
                f.write(np.ascontiguousarray(cases["Tcw"][i], np.float32).tobytes()); f.write(np.ascontiguousarray(cases["X"][i], np.float64).tobytes())
                f.write(np.ascontiguousarray(cases["obs"][i], np.float64).tobytes()); f.write(np.ascontiguousarray(cases["lobs"][i], np.float64).tobytes())
                f.write(np.ascontiguousarray(cases["pw"][i], np.float32).tobytes()); f.write(np.ascontiguousarray(cases["pm"][i], np.float32).tobytes())
        subprocess.check_call([ref_opt_path(), "edges", fin, fout], stdout=subprocess.DEVNULL)
        out = np.fromfile(fout, np.float64).reshape(n, 12 * 22 + 16)
    return out[:, :12 * 22].reshape(n, 12, 22).copy(), out[:, 12 * 22:].reshape(n, 4, 4).copy()


def pose_edges_eval(cases, params):
    L = lib()
    n = len(cases["Tcw"])
    out = np.zeros((n, 12, 22), np.float64)
    prm = pose_params(params)
    a = {k: np.ascontiguousarray(cases[k], np.float32 if k in ("Tcw", "pw", "pm") else np.float64) for k in ("Tcw", "X", "obs", "lobs", "pw", "pm")}
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    L.orc_pose_edges_eval(C.c_int(n), p(a["Tcw"]), p(a["X"]), p(a["obs"]), p(a["lobs"]), p(a["pw"]), p(a["pm"]), C.byref(prm), p(out))
    return out


# ---- REAL reference Frame / Tracking function bodies (oracle/_ref/ref_frame: line ranges of src/Frame.cc, MapPoint.cc, MapLine.cpp, Tracking.cc) ----
def ref_frame_path():
    return os.path.join(ORACLE_DIR, "_ref", "ref_frame")


def _run_ref_frame(mode, payload):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(payload)
        subprocess.check_call([ref_frame_path(), mode, fin, fout], stdout=subprocess.DEVNULL)
        return open(fout, "rb").read()


def run_ref_manhattan(R_last, normals, lines):
    """The reference's own Tracking::TrackManhattanFrame for one frame -> dict(R [3,3] f32, member [n+nl] u8)."""
    R_last = np.ascontiguousarray(R_last, np.float32); normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    lines = np.ascontiguousarray(lines, np.float64).reshape(-1, 3)
    n, nl = len(normals), len(lines)
    buf = _run_ref_frame("manhattan", np.int32(1).tobytes() + R_last.tobytes() + np.array([n, nl], np.int32).tobytes() + normals.tobytes() + lines.tobytes())
    return dict(R=np.frombuffer(buf, "<f4", 9, 0).reshape(3, 3).copy(), member=np.frombuffer(buf, np.uint8, n + nl, 36).copy())


def run_ref_manhattan_pose(Rcm0, MF_can, Tcw):
    """The reference's own statements src/Tracking.cc:251-253 + :1778 (oracle/_ref/ref_frame manhattan_pose): mRotation_wc = (Rotation_cm * MF_can^T)^T copied into mTcw."""
    Rcm0 = np.ascontiguousarray(Rcm0, np.float32).reshape(-1, 9); MF = np.ascontiguousarray(MF_can, np.float32).reshape(-1, 9); T = np.ascontiguousarray(Tcw, np.float32).reshape(-1, 16)
    n = len(T)
    pay = np.int32(n).tobytes() + b"".join(Rcm0[b].tobytes() + MF[b].tobytes() + T[b].tobytes() for b in range(n))
    return np.frombuffer(_run_ref_frame("manhattan_pose", pay), "<f4").reshape(n, 16).copy()


def manhattan_pose(Rcm0, MF_can, Tcw):
    """oracle/manhattan_oracle.cpp orc_manhattan_pose: [n,9], [n,9], [n,16] f32 -> [n,16] f32"""
    L = lib()
    Rcm0 = np.ascontiguousarray(Rcm0, np.float32).reshape(-1, 9); MF = np.ascontiguousarray(MF_can, np.float32).reshape(-1, 9); T = np.ascontiguousarray(Tcw, np.float32).reshape(-1, 16)
    out = np.zeros_like(T)
    L.orc_manhattan_pose(C.c_void_p(Rcm0.ctypes.data), C.c_void_p(MF.ctypes.data), C.c_void_p(T.ctypes.data), C.c_void_p(out.ctypes.data), len(T))
    return out


def _camera_block(frame, log_scale_factor, n_levels):
    cam = np.array([frame["fx"], frame["fy"], frame["cx"], frame["cy"], frame["bf"], frame["min_x"], frame["max_x"], frame["min_y"], frame["max_y"],
                    log_scale_factor], np.float32)
    return cam.tobytes() + np.int32(n_levels).tobytes()


def run_ref_frustum_points(frame, mp, b, log_scale_factor, n_levels, limit=0.5):
    """Frame::isInFrustum(MapPoint*, limit) of the reference for every valid map point of frame b."""
    n = int(mp["n"][b])
    idx = np.nonzero(mp["valid"][b, :n])[0]
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    pay = (_camera_block(frame, log_scale_factor, n_levels) + c(frame["Tcw"][b], np.float32).tobytes() + np.float32(limit).tobytes() + np.int32(len(idx)).tobytes() +
           c(mp["xw"][b, idx], np.float32).tobytes() + c(mp["normal"][b, idx], np.float32).tobytes() + c(mp["min_dist"][b, idx], np.float32).tobytes() +
           c(mp["max_dist"][b, idx], np.float32).tobytes())
    rec = np.frombuffer(_run_ref_frame("frustum_points", pay), np.dtype([("ret", "u1"), ("in_view", "u1"), ("px", "<f4"), ("py", "<f4"), ("pxr", "<f4"), ("level", "<i4"), ("vc", "<f4")]))
    return idx, rec


def run_ref_frustum_lines(frame, ml, b, log_scale_factor, limit=0.5):
    n = int(ml["n"][b])
    idx = np.nonzero(ml["valid"][b, :n])[0]
    c = lambda a, dt: np.ascontiguousarray(a, dt)
    pay = (_camera_block(frame, log_scale_factor, 8) + c(frame["Tcw"][b], np.float32).tobytes() + np.float32(limit).tobytes() + np.int32(len(idx)).tobytes() +
           c(ml["xw6"][b, idx], np.float64).tobytes() + c(ml["normal"][b, idx], np.float64).tobytes() + c(ml["min_dist"][b, idx], np.float32).tobytes() +
           c(ml["max_dist"][b, idx], np.float32).tobytes())
    rec = np.frombuffer(_run_ref_frame("frustum_lines", pay), np.dtype([("ret", "u1"), ("p", "<f4", 4), ("level", "<i4"), ("vc", "<f4")]))
    return idx, rec


def stereo_from_rgbd(keys, depth, Tcw, cam, depth_factor=1.0 / 5000.0, keys_un=None):
    """CPU oracle of Frame::ComputeStereoFromRGBD + UnprojectStereo for ONE frame: keys [n] KP_DTYPE (mvKeys; keys_un = mvKeysUn, default the same), depth [H, W] uint16."""
    L = lib()
    keys = np.ascontiguousarray(keys, KP_DTYPE); keys_un = keys if keys_un is None else np.ascontiguousarray(keys_un, KP_DTYPE); depth = np.ascontiguousarray(depth, np.uint16); Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    n = len(keys)
    out = dict(u_right=np.zeros(n, np.float32), depth=np.zeros(n, np.float32), xw=np.zeros((n, 3), np.float32), valid=np.zeros(n, np.uint8))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.orc_stereo_from_rgbd(p(keys), p(keys_un), C.c_int(n), p(depth), C.c_int(depth.shape[1]), C.c_float(np.float32(depth_factor)), C.c_float(cam["fx"]), C.c_float(cam["fy"]),
                           C.c_float(cam["cx"]), C.c_float(cam["cy"]), C.c_float(cam["bf"]), p(Tcw), p(out["u_right"]), p(out["depth"]), p(out["xw"]), p(out["valid"]))
    return out


def undistort_keypoints(keys, cam, dist_coef):
    """CPU oracle of Frame::UndistortKeyPoints for ONE frame (cv::undistortPoints' published loop; unpinned): keys [n] KP_DTYPE, dist_coef (k1, k2, p1, p2, k3)."""
    keys = np.ascontiguousarray(keys, KP_DTYPE); out = np.zeros_like(keys); d = np.ascontiguousarray(dist_coef, np.float32)
    assert d.shape == (5,)
    lib().orc_undistort_keypoints(keys.ctypes.data_as(C.c_void_p), C.c_int(len(keys)), C.c_float(cam["fx"]), C.c_float(cam["fy"]), C.c_float(cam["cx"]), C.c_float(cam["cy"]),
                                  d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def run_ref_stereo(keys, depth, Tcw, cam, depth_factor=1.0 / 5000.0):
    """The reference's own Frame::ComputeStereoFromRGBD + UnprojectStereo (oracle/_ref/ref_frame stereo) for one frame.  The float depth image
    is what imDepth.convertTo(CV_32F, factor) produces: float(u16) * float(factor)."""
    keys = np.ascontiguousarray(keys, KP_DTYPE); depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    imf = (depth.astype(np.float32) * np.float32(depth_factor)).astype(np.float32)
    frame = dict(fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], bf=cam["bf"], min_x=0.0, max_x=float(W), min_y=0.0, max_y=float(H))
    kp = np.stack([keys["x"], keys["y"]], 1).astype(np.float32)
    pay = (_camera_block(frame, 0.18, 8) + np.ascontiguousarray(Tcw, np.float32).tobytes() + np.array([W, H, len(keys)], np.int32).tobytes() + imf.tobytes() + kp.tobytes())
    rec = np.frombuffer(_run_ref_frame("stereo", pay), np.dtype([("u_right", "<f4"), ("depth", "<f4"), ("xw", "<f4", 3)]))
    return rec


def surface_normals(depth, cam=(535.4, 539.2, 320.1, 247.6), factor=1.0 / 5000.0, want_dist=False):
    """Frame::ComputePlanes' surface normals (oracle/normals_oracle.cpp) of one u16 depth frame -> normals [n,3], points [n,3] (+ chamfer map)."""
    L = lib()
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    gw, gh = -(-W // 3), -(-H // 3)
    cap = (gw // 2) * (gh // 2)
    nrm = np.zeros((cap, 3), np.float32); pts = np.zeros((cap, 3), np.float32); dist = np.zeros((gh, gw), np.float32)
    L.orc_surface_normals.restype = C.c_int
    n = L.orc_surface_normals(C.c_void_p(depth.ctypes.data), W, H, W, C.c_float(np.float32(factor)), C.c_float(cam[0]), C.c_float(cam[1]), C.c_float(cam[2]), C.c_float(cam[3]),
                              C.c_void_p(nrm.ctypes.data), C.c_void_p(pts.ctypes.data), cap, C.c_void_p(dist.ctypes.data) if want_dist else None)
    assert n == cap, (n, cap)
    return (nrm, pts, dist) if want_dist else (nrm, pts)


# ---- DBoW2 vocabulary transform (oracle/bow_oracle.cpp; the REAL Thirdparty/DBoW2 through oracle/_ref/ref_bow) ----
class VocabOracle:
    def __init__(self, voc):
        L = lib()
        L.orc_vocab_create.restype = C.c_void_p
        self.L = L
        a = {k: np.ascontiguousarray(voc[k]) for k in ("parent", "is_leaf", "desc", "weight")}
        self.h = C.c_void_p(L.orc_vocab_create(int(voc["k"]), int(voc["L"]), len(a["parent"]), C.c_void_p(a["parent"].ctypes.data), C.c_void_p(a["is_leaf"].ctypes.data),
                                               C.c_void_p(a["desc"].ctypes.data), C.c_void_p(a["weight"].ctypes.data)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_vocab_destroy(self.h); self.h = None

    def transform(self, desc, levelsup=4):
        """-> dict(word [n], weight [n], node [n] (-1: stopped word), bow_word [m], bow_value [m])"""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        word = np.zeros(n, np.int32); wt = np.zeros(n); node = np.zeros(n, np.int32); bw = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1))
        m = self.L.orc_bow_transform(self.h, C.c_void_p(desc.ctypes.data), n, int(levelsup), C.c_void_p(word.ctypes.data), C.c_void_p(wt.ctypes.data),
                                     C.c_void_p(node.ctypes.data), C.c_void_p(bw.ctypes.data), C.c_void_p(bv.ctypes.data))
        return dict(word=word, weight=wt, node=node, bow_word=bw[:m].copy(), bow_value=bv[:m].copy())


def ref_bow_path():
    return os.path.join(ORACLE_DIR, "_ref", "ref_bow")


def run_ref_bow(voc_txt, desc, levelsup=4):
    """The real DBoW2 TemplatedVocabulary::transform on a vocabulary text file -> same dict as VocabOracle.transform."""
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([levelsup, n], np.int32).tobytes()); f.write(desc.tobytes())
        subprocess.check_call([ref_bow_path(), voc_txt, fin, fout])
        buf = open(fout, "rb").read()
    m = int(np.frombuffer(buf, "<i4", 1, 0)[0]); off = 4
    pairs = np.frombuffer(buf, np.dtype([("w", "<u4"), ("v", "<f8")]), m, off); off += 12 * m
    node = np.frombuffer(buf, "<i4", n, off).copy(); off += 4 * n
    word = np.frombuffer(buf, "<u4", n, off).astype(np.int32); off += 4 * n
    wt = np.frombuffer(buf, "<f8", n, off).copy(); off += 8 * n
    assert off == len(buf)
    return dict(word=word, weight=wt, node=node, bow_word=pairs["w"].astype(np.int32), bow_value=pairs["v"].copy())


# ---- 3-D line back-projection (oracle/line3d_oracle.cpp: Frame::isLineGood + extract3dline_mahdist) ----
def glibc_rand(seed, count):
    out = np.zeros(count, np.int32)
    lib().orc_glibc_rand(C.c_uint32(seed), count, C.c_void_p(out.ctypes.data))
    return out


def is_line_good(keylines, depth, seed, cam=(535.4, 539.2, 320.1, 247.6), factor=1.0 / 5000.0):
    """Frame::isLineGood for one frame -> dict(depth_line [n] f32, lines3d [n,6] f64, good [n] u8, direction [n,3] f64, n_inliers [n], n_samples [n])."""
    from planarslam_amd._lib import KEYLINE_DTYPE
    L = lib()
    kl = np.ascontiguousarray(keylines, KEYLINE_DTYPE)
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    n = len(kl)
    m = max(n, 1)
    out = dict(depth_line=np.zeros(m, np.float32), lines3d=np.zeros((m, 6)), good=np.zeros(m, np.uint8), direction=np.zeros((m, 3)), n_inliers=np.zeros(m, np.int32),
               n_samples=np.zeros(m, np.int32))
    L.orc_is_line_good.restype = C.c_int
    L.orc_is_line_good(C.c_void_p(kl.ctypes.data), n, C.c_void_p(depth.ctypes.data), W, H, W, C.c_float(np.float32(factor)), C.c_float(cam[0]), C.c_float(cam[1]),
                       C.c_float(cam[2]), C.c_float(cam[3]), C.c_uint32(seed), *[C.c_void_p(out[k].ctypes.data) for k in ("depth_line", "lines3d", "good", "direction", "n_inliers", "n_samples")])
    return {k: v[:n] for k, v in out.items()}


def ref_line3d_path():
    return os.path.join(ORACLE_DIR, "_ref", "ref_line3d")


def run_ref_line3d(keylines, depth, seed, cam=(535.4, 539.2, 320.1, 247.6), factor=1.0 / 5000.0):
    """The reference's own Frame::isLineGood (+ src/LineExtractor.cpp helpers), srand(seed + i) before line i -> the fields of is_line_good() (no n_samples)."""
    from planarslam_amd._lib import KEYLINE_DTYPE
    kl = np.ascontiguousarray(keylines, KEYLINE_DTYPE)
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    n = len(kl)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([n, W, H], np.int32).tobytes()); f.write(np.array([seed], np.uint32).tobytes())
            f.write(np.array([cam[0], cam[1], cam[2], cam[3], factor], np.float32).tobytes()); f.write(kl.tobytes()); f.write(depth.tobytes())
        subprocess.check_call([ref_line3d_path(), fin, fout], stdout=subprocess.DEVNULL)
        buf = open(fout, "rb").read()
    rec = np.frombuffer(buf, np.dtype([("depth_line", "<f4"), ("lines3d", "<f8", 6), ("good", "u1"), ("direction", "<f8", 3), ("n_inliers", "<i4")]), n)
    return {k: rec[k].copy() for k in rec.dtype.names}


# ---- plane post-processing (oracle/planepost_oracle.cpp; PCL VoxelGrid + SACSegmentation restated, PARITY UNPINNED) ----
REFIT_INFO = ("iterations", "best_count", "s0", "s1", "s2", "n_inliers", "n_inliers_refined", "draws")


def _refit_info(raw):
    raw = np.asarray(raw, np.int32)
    d = {k: int(raw[i]) for i, k in enumerate(REFIT_INFO)}
    d["model"] = raw[8:12].copy().view(np.float32)
    return d


def voxel_grid(pts, leaf=0.1, want_exact=False):
    """pcl::VoxelGrid centroids [m,3] float32 of pts [n,3] (+ exact double centroids and point counts of the same voxels)."""
    L = lib()
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    n = len(pts)
    out = np.zeros((max(n, 1), 3), np.float32); ex = np.zeros((max(n, 1), 3)); cnt = np.zeros(max(n, 1), np.int32)
    L.orc_voxel_grid.restype = C.c_int
    m = L.orc_voxel_grid(C.c_void_p(pts.ctypes.data), n, C.c_float(leaf), C.c_void_p(out.ctypes.data), max(n, 1), C.c_void_p(ex.ctypes.data) if want_exact else None,
                         C.c_void_p(cnt.ctypes.data) if want_exact else None)
    assert m >= 0, m
    return (out[:m].copy(), ex[:m].copy(), cnt[:m].copy()) if want_exact else out[:m].copy()


def plane_refit(plane, pts, dis_th):
    """Frame::MaxPointDistanceFromPlane -> (state: 0 kept / 1 distance / 2 no inliers, plane [4] f32, info)."""
    L = lib()
    plane = np.array(plane, np.float32).reshape(4).copy(); pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    info = np.zeros(12, np.int32)
    L.orc_plane_refit.restype = C.c_int
    st = L.orc_plane_refit(C.c_void_p(plane.ctypes.data), C.c_void_p(pts.ctypes.data), len(pts), C.c_double(dis_th), C.c_void_p(info.ctypes.data))
    return st, plane, _refit_info(info)


def plane_clouds(depth, labels, planes, dis_th=0.05, leaf=0.1, cam=(535.4, 539.2, 320.1, 247.6), factor=1.0 / 5000.0):
    """The head of Frame::ComputePlanes for one frame: dict(n, coef [n,4], src [n], pt_off [n+1], points [m,3], state [P], nvox [P], info [P])."""
    L = lib()
    depth = np.ascontiguousarray(depth, np.uint16); labels = np.ascontiguousarray(labels, np.int32); planes = np.ascontiguousarray(planes, np.float64).reshape(-1, 8)
    H, W = depth.shape
    P = len(planes)
    capP, capN = max(P, 1), 1 << 16
    coef = np.zeros((capP, 4), np.float32); src = np.zeros(capP, np.int32); off = np.zeros(capP + 1, np.int32); pts = np.zeros((capN, 3), np.float32)
    state = np.zeros(capP, np.int32); nvox = np.zeros(capP, np.int32); info = np.zeros((capP, 12), np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    L.orc_plane_clouds.restype = C.c_int
    n = L.orc_plane_clouds(p(depth), W, H, W, C.c_float(np.float32(factor)), C.c_float(cam[0]), C.c_float(cam[1]), C.c_float(cam[2]), C.c_float(cam[3]), p(labels), p(planes), P,
                           C.c_double(dis_th), C.c_float(leaf), p(coef), p(src), p(off), p(pts), capP, capN, p(state), p(nvox), p(info))
    assert n >= 0, n
    return dict(n=n, coef=coef[:n].copy(), src=src[:n].copy(), pt_off=off[:n + 1].copy(), points=pts[:off[n]].copy(), state=state[:P].copy(), nvox=nvox[:P].copy(),
                info=[_refit_info(info[i]) for i in range(P)])


def flag_matched_plane_points(Tcw, coef, matched, xw):
    L = lib()
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); coef = np.ascontiguousarray(coef, np.float32).reshape(-1, 4)
    matched = np.ascontiguousarray(matched, np.uint8); xw = np.ascontiguousarray(xw, np.float32).reshape(-1, 3)
    flags = np.zeros(len(xw), np.uint8)
    L.orc_flag_matched_plane_points.restype = C.c_int
    nm = L.orc_flag_matched_plane_points(C.c_void_p(Tcw.ctypes.data), C.c_void_p(coef.ctypes.data), C.c_void_p(matched.ctypes.data), len(coef), C.c_void_p(xw.ctypes.data),
                                         len(xw), C.c_void_p(flags.ctypes.data))
    return flags, nm


def merge_plane_points(T, frame_pts, map_pts, leaf=0.1):
    L = lib()
    T = np.ascontiguousarray(T, np.float64).reshape(16); f = np.ascontiguousarray(frame_pts, np.float32).reshape(-1, 3); m = np.ascontiguousarray(map_pts, np.float32).reshape(-1, 3)
    out = np.zeros((len(f) + len(m) + 1, 3), np.float32)
    L.orc_merge_plane_points.restype = C.c_int
    n = L.orc_merge_plane_points(C.c_void_p(T.ctypes.data), C.c_void_p(f.ctypes.data), len(f), C.c_void_p(m.ctypes.data), len(m), C.c_float(leaf), C.c_void_p(out.ctypes.data), len(out))
    assert n >= 0
    return out[:n].copy()


def sac_rnd(n, seed=12345):
    out = np.zeros(n, np.int32)
    lib().orc_sac_rnd(C.c_uint32(seed), n, C.c_void_p(out.ctypes.data))
    return out


# ---- MapPoint::ComputeDistinctiveDescriptors (oracle/match_oracle.cpp; the REAL src/MapPoint.cc:259-324 through oracle/_ref/ref_frame) ----
def distinctive_descriptor(desc):
    """-> (index of the chosen descriptor (-1: none), its median distance)"""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    med = C.c_int()
    L = lib()
    L.orc_distinctive_descriptor.restype = C.c_int
    return L.orc_distinctive_descriptor(C.c_void_p(desc.ctypes.data), len(desc), C.byref(med)), med.value


def run_ref_distinctive(points, lines=False):
    """points: list of (desc [n,32] u8, bad [n] u8) -> [len(points), 32] u8: mDescriptor after the reference's own ComputeDistinctiveDescriptors (zeros if unset).
    lines=True: MapLine::ComputeDistinctiveDescriptors (src/MapLine.cpp:241-312) on LBD rows instead."""
    pay = np.int32(len(points)).tobytes()
    for desc, bad in points:
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); bad = np.ascontiguousarray(bad, np.uint8)
        pay += np.int32(len(desc)).tobytes() + bad.tobytes() + desc.tobytes()
    return np.frombuffer(_run_ref_frame("distinctive_lines" if lines else "distinctive", pay), np.uint8).reshape(len(points), 32).copy()


# ---- MapPoint::UpdateNormalAndDepth (oracle/guided_oracle.cpp; the REAL src/MapPoint.cc:347-388 + KeyFrame::SetPose through oracle/_ref/ref_frame) ----
def keyframe_center(Tcw):
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); ow = np.zeros(3, np.float32)
    lib().orc_keyframe_center(C.c_void_p(Tcw.ctypes.data), C.c_void_p(ow.ctypes.data))
    return ow


def update_normal_and_depth(pos, obs_ow, ref_ow, level, sf):
    pos = np.ascontiguousarray(pos, np.float32).reshape(3); obs_ow = np.ascontiguousarray(obs_ow, np.float32).reshape(-1, 3); ref_ow = np.ascontiguousarray(ref_ow, np.float32).reshape(3)
    sf = np.ascontiguousarray(sf, np.float32)
    nrm = np.zeros(3, np.float32); mn = C.c_float(); mx = C.c_float()
    lib().orc_update_normal_and_depth(C.c_void_p(pos.ctypes.data), C.c_void_p(obs_ow.ctypes.data), len(obs_ow), C.c_void_p(ref_ow.ctypes.data), int(level), C.c_void_p(sf.ctypes.data),
                                      len(sf), C.c_void_p(nrm.ctypes.data), C.byref(mn), C.byref(mx))
    return nrm, np.float32(mn.value), np.float32(mx.value)


def run_ref_normal_depth(Tcws, sf, points):
    """Tcws [nkf,16]; points: list of (pos [3], ref, level, obs list) -> [np, 5] f32 (normal, min, max) from the reference's own UpdateNormalAndDepth."""
    Tcws = np.ascontiguousarray(Tcws, np.float32).reshape(-1, 16); sf = np.ascontiguousarray(sf, np.float32)
    pay = np.array([len(Tcws), len(sf)], np.int32).tobytes() + sf.tobytes() + Tcws.tobytes() + np.int32(len(points)).tobytes()
    for pos, ref, level, obs in points:
        pay += np.ascontiguousarray(pos, np.float32).tobytes() + np.array([ref, level, len(obs)], np.int32).tobytes() + np.ascontiguousarray(obs, np.int32).tobytes()
    return np.frombuffer(_run_ref_frame("normal_depth", pay), np.float32).reshape(len(points), 5).copy()
