"""Synthetic inputs for tests and bench.  numpy only; deterministic per seed."""
from __future__ import annotations

import numpy as np


def _bilinear_up(grid: np.ndarray, h: int, w: int) -> np.ndarray:
    gh, gw = grid.shape
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.floor(ys).astype(int).clip(0, gh - 2)
    x0 = np.floor(xs).astype(int).clip(0, gw - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = grid[y0][:, x0]
    b = grid[y0][:, x0 + 1]
    c = grid[y0 + 1][:, x0]
    d = grid[y0 + 1][:, x0 + 1]
    return a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx


def gray_image(seed: int = 1234, w: int = 640, h: int = 480) -> np.ndarray:
    """4-octave value noise + 60 axis-aligned + 20 rotated rectangles + N(0,2) pixel noise, u8."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 96.0)
    for (gw, gh), amp in zip([(5, 4), (10, 8), (20, 15), (40, 30)], [64, 32, 16, 8]):
        img += amp * (_bilinear_up(rng.random((gh, gw)), h, w) - 0.5) * 2
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(60):
        x0, y0 = rng.integers(0, w - 8), rng.integers(0, h - 8)
        rw, rh = rng.integers(6, 120), rng.integers(6, 90)
        img[y0:y0 + rh, x0:x0 + rw] = rng.uniform(0, 255)
    for _ in range(20):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        rw, rh = rng.uniform(8, 80), rng.uniform(8, 60)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        img[(np.abs(u) < rw / 2) & (np.abs(v) < rh / 2)] = rng.uniform(0, 255)
    img += rng.normal(0, 2, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def gray_batch(n: int, seed: int = 1234, w: int = 640, h: int = 480) -> np.ndarray:
    return np.stack([gray_image(seed + i, w, h) for i in range(n)])


# ----------------------------------------------------------------------------------------------
# Pose-optimisation problems: SoA batch in the layout of
# include/planar_abi.h `planar_pose_batch`.
TUM3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, bf=40.0, angle_info=0.5, distance_info=50.0, parallel_info=0.1,
            vertical_info=0.1, plane_chi=100.0, vp_chi=50.0)
LEVEL_COUNTS = np.array([217, 181, 151, 126, 105, 87, 73, 60])


def _rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def pose_batch(B=4, n_points=1000, n_lines=75, n_planes=4, seed=7, outlier_frac=0.10, stereo_frac=0.85,
               max_points=None, max_lines=None, max_planes=None, rot_pert=0.02, trans_pert=0.05):
    """B synthetic frames.  Returns a dict of contiguous numpy arrays (+ 'T_gt' [B,4,4] float64)."""
    P = TUM3
    MP, ML, MM = max_points or n_points, max_lines or n_lines, max_planes or n_planes
    scale = 1.2 ** np.arange(8)
    out = dict(
        n_points=np.full(B, n_points, np.int32), n_lines=np.full(B, n_lines, np.int32), n_planes=np.full(B, n_planes, np.int32),
        pt_valid=np.zeros((B, MP), np.uint8), pt_xw=np.zeros((B, MP, 3), np.float32), pt_obs=np.zeros((B, MP, 3), np.float32),
        pt_inv_sigma2=np.ones((B, MP), np.float32), ln_valid=np.zeros((B, ML), np.uint8), ln_obs=np.zeros((B, ML, 3), np.float64),
        ln_xw=np.zeros((B, ML, 6), np.float64), pl_meas=np.zeros((B, MM, 4), np.float32), pl_valid=np.zeros((B, MM, 3), np.uint8),
        pl_world=np.zeros((B, MM, 3, 4), np.float32), Tcw=np.zeros((B, 16), np.float32), T_gt=np.zeros((B, 4, 4)))
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        w = rng.normal(size=3); w *= rng.uniform(0, 0.05) / np.linalg.norm(w)
        t = rng.normal(size=3); t *= rng.uniform(0, 0.1) / np.linalg.norm(t)
        R = _rodrigues(w)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        out["T_gt"][b] = T
        Rinv, tinv = R.T, -R.T @ t

        def backproject(u, v, z):
            return np.stack([(u - P["cx"]) * z / P["fx"], (v - P["cy"]) * z / P["fy"], z], -1)

        def project(Xw):
            Xc = Xw @ R.T + t
            return np.stack([Xc[..., 0] / Xc[..., 2] * P["fx"] + P["cx"], Xc[..., 1] / Xc[..., 2] * P["fy"] + P["cy"]], -1), Xc[..., 2]

        # points
        u, v, z = rng.uniform(20, 620, n_points), rng.uniform(20, 460, n_points), rng.uniform(0.5, 6, n_points)
        Xw = (backproject(u, v, z) @ Rinv.T + tinv).astype(np.float32)
        octave = rng.choice(8, n_points, p=LEVEL_COUNTS / LEVEL_COUNTS.sum())
        uv, zc = project(Xw.astype(np.float64))
        sig = scale[octave]
        obs = np.zeros((n_points, 3))
        obs[:, :2] = uv + rng.normal(size=(n_points, 2)) * sig[:, None]
        obs[:, 2] = obs[:, 0] - P["bf"] / zc + rng.normal(size=n_points) * sig * 0.5
        mono = rng.random(n_points) > stereo_frac
        obs[mono, 2] = -1
        bad = rng.random(n_points) < outlier_frac
        obs[bad, 0] = rng.uniform(0, 640, bad.sum()); obs[bad, 1] = rng.uniform(0, 480, bad.sum())
        out["pt_valid"][b, :n_points] = (rng.random(n_points) < 0.97).astype(np.uint8)    # a few NULL map points
        out["pt_xw"][b, :n_points] = Xw
        out["pt_obs"][b, :n_points] = obs.astype(np.float32)
        out["pt_inv_sigma2"][b, :n_points] = (1.0 / (scale[octave].astype(np.float32) ** 2)).astype(np.float32)
        # lines: two 3-D endpoints; observation = normalised cross product of noisy projected endpoints
        for i in range(n_lines):
            uu, vv, zz = rng.uniform(30, 610, 2), rng.uniform(30, 450, 2), rng.uniform(0.8, 5, 2)
            Xl = backproject(uu, vv, zz) @ Rinv.T + tinv
            pe, _ = project(Xl)
            pe = pe + rng.normal(size=(2, 2)) * 0.7
            if rng.random() < outlier_frac:
                pe = pe + rng.uniform(-60, 60, (2, 2))
            l = np.cross(np.append(pe[0], 1.0), np.append(pe[1], 1.0))
            out["ln_obs"][b, i] = l / np.linalg.norm(l)
            out["ln_xw"][b, i] = Xl.reshape(6)
            out["ln_valid"][b, i] = 1 if rng.random() < 0.95 else 0
        # planes: observed camera-frame plane i with a matched, a parallel and a vertical map plane
        for i in range(n_planes):
            nc = rng.normal(size=3); nc /= np.linalg.norm(nc)
            dc = rng.uniform(0.8, 3.0)
            nw = Rinv @ nc
            dw = dc + t @ nc            # d_c = d_w - t.n_c  (Plane3D operator*)
            def noisy(n, d, ang=np.radians(0.5), dd=0.005):
                a = rng.normal(size=3); a -= a.dot(n) * n; a /= np.linalg.norm(a)
                n2 = n * np.cos(ang) + a * np.sin(ang) * rng.normal()
                n2 /= np.linalg.norm(n2)
                return np.append(n2, d + rng.normal() * dd)
            out["pl_meas"][b, i] = noisy(nc, dc).astype(np.float32)
            out["pl_world"][b, i, 0] = np.append(nw, dw).astype(np.float32)
            out["pl_world"][b, i, 1] = np.append(nw, dw + rng.uniform(0.3, 1.0)).astype(np.float32)     # parallel, other offset
            a = rng.normal(size=3); a -= a.dot(nw) * nw; a /= np.linalg.norm(a)
            out["pl_world"][b, i, 2] = np.append(a, rng.uniform(0.5, 2.0)).astype(np.float32)          # perpendicular
            out["pl_valid"][b, i] = [1, 1, 1]
        # initial pose = ground truth perturbed
        dw_ = rng.normal(size=3); dw_ *= rot_pert / np.linalg.norm(dw_)
        dt_ = rng.normal(size=3); dt_ *= trans_pert / np.linalg.norm(dt_)
        T0 = np.eye(4); T0[:3, :3] = _rodrigues(dw_) @ R; T0[:3, 3] = t + dt_
        out["Tcw"][b] = T0.astype(np.float32).reshape(16)
    return out


# ----------------------------------------------------------------------------------------------
# Depth images: a box room of planes rendered by ray-plane intersection, TUM units.
def depth_image(seed: int = 4321, w: int = 640, h: int = 480, factor: float = 5000.0, noise: bool = True, holes: bool = True):
    """u16 depth (TUM scale: value / 5000 = metres): floor, ceiling, two or three walls and a table-top box seen
    from a random camera; depth-dependent noise; ~3 % of the pixels zeroed in blobs."""
    rng = np.random.default_rng(seed)
    P = TUM3
    yy, xx = np.mgrid[0:h, 0:w]
    rays = np.stack([(xx - P["cx"]) / P["fx"], (yy - P["cy"]) / P["fy"], np.ones_like(xx, dtype=np.float64)], -1)
    R = _rodrigues(rng.normal(size=3) * 0.15)
    rays = rays @ R.T
    planes = [  # n.X + d = 0 in the camera frame before the random rotation
        (np.array([0.0, -1.0, 0.0]), rng.uniform(0.9, 1.4)),      # floor (y down)
        (np.array([0.0, 1.0, 0.0]), rng.uniform(1.0, 1.5)),       # ceiling
        (np.array([0.0, 0.0, -1.0]), rng.uniform(2.5, 4.0)),      # back wall
        (np.array([1.0, 0.0, 0.0]), rng.uniform(1.2, 2.5)),       # left wall
        (np.array([-1.0, 0.0, 0.0]), rng.uniform(1.2, 2.5)),      # right wall
    ]
    z = np.full((h, w), np.inf)
    for n, d in planes:
        denom = rays @ n
        t = np.where(np.abs(denom) > 1e-9, -d / denom, np.inf)
        t = np.where(t > 0.05, t, np.inf)
        z = np.minimum(z, t * rays[..., 2])
    # a box (table) in front of the back wall: top face + front face
    bx0, bx1 = sorted(rng.uniform(-0.8, 0.8, 2)); by = rng.uniform(0.2, 0.6); bz0 = rng.uniform(1.2, 2.0); bz1 = bz0 + rng.uniform(0.4, 0.9)
    t = by / np.where(np.abs(rays[..., 1]) > 1e-9, rays[..., 1], np.inf)           # plane y = by (top)
    X = rays * t[..., None]
    ok = (t > 0) & (X[..., 0] > bx0) & (X[..., 0] < bx1) & (X[..., 2] > bz0) & (X[..., 2] < bz1)
    z = np.where(ok & (X[..., 2] < z), X[..., 2], z)
    t = bz0 / rays[..., 2]                                                         # plane z = bz0 (front)
    X = rays * t[..., None]
    ok = (X[..., 0] > bx0) & (X[..., 0] < bx1) & (X[..., 1] > by) & (X[..., 1] < planes[0][1])
    z = np.where(ok & (X[..., 2] < z), X[..., 2], z)
    z = np.where(np.isfinite(z), z, 0.0)
    if noise:
        z = z + rng.normal(size=z.shape) * 0.0012 * z * z
    d = np.clip(np.rint(z * factor), 0, 65535).astype(np.uint16)
    if holes:
        for _ in range(12):
            cx, cy, r = rng.integers(0, w), rng.integers(0, h), rng.integers(8, 40)
            d[(xx - cx) ** 2 + (yy - cy) ** 2 < r * r] = 0
    return d


# ----------------------------------------------------------------------------------------------
# Local bundle adjustment problems, layout of include/planar_abi.h `planar_ba_problem`.
BE_MONO, BE_STEREO, BE_LINE, BE_PLANE, BE_VER, BE_PAR = range(6)


def ba_problem(seed=99, n_kf=10, n_points=2400, n_lines=500, n_planes=100, n_fixed_extra=2, outlier_frac=0.03, pose_noise=(0.01, 0.03),
               point_noise=0.02, lines_on_kf=None):
    """Keyframes on a 2 m arc looking at a cloud of points / line segments / planes 2-6 m away.  KF 0 is fixed, plus
    `n_fixed_extra` fixed observers at the end.  Every landmark is seen by 4-10 keyframes.  Returns a dict of numpy arrays
    (+ 'T_gt' [n_kf,4,4], 'lm_gt' [n_lm,4]).
    lines_on_kf = k reproduces the graph the reference's LocalBundleAdjustment(pKF = k) really builds: every line edge hangs on
    keyframe k's vertex and measures k's line function (src/Optimizer.cc:2170-2194 use pKF, not the observer pKFi); the observing
    keyframe of each edge is kept in 'e_obs_kf' (it owns the erase verdict)."""
    rng = np.random.default_rng(seed)
    P = TUM3
    K = n_kf + n_fixed_extra
    T_gt = np.zeros((K, 4, 4))
    for k in range(K):
        ang = (k / max(1, K - 1) - 0.5) * 0.6
        Rwc = _rodrigues(np.array([0.0, ang, 0.0])) @ _rodrigues(rng.normal(size=3) * 0.03)
        c = np.array([2.0 * np.sin(ang), 0.05 * rng.normal(), -2.0 * (1 - np.cos(ang))])
        T = np.eye(4); T[:3, :3] = Rwc.T; T[:3, 3] = -Rwc.T @ c
        T_gt[k] = T
    fixed = np.zeros(K, np.uint8); fixed[0] = 1; fixed[n_kf:] = 1
    scale = 1.2 ** np.arange(8)

    def proj(T, X):
        Xc = T[:3, :3] @ X + T[:3, 3]
        return np.array([Xc[0] / Xc[2] * P["fx"] + P["cx"], Xc[1] / Xc[2] * P["fy"] + P["cy"]]), Xc[2]

    lm_type, lm_gt, e_kf, e_lm, e_type, e_meas, e_is2, e_obs = [], [], [], [], [], [], [], []

    def observers():
        n = rng.integers(4, min(10, K) + 1)
        return np.sort(rng.choice(K, n, replace=False))

    def sample_point():
        return np.array([rng.uniform(-2.5, 2.5), rng.uniform(-1.2, 1.2), rng.uniform(2.0, 6.0)])

    for _ in range(n_points):
        X = sample_point(); l = len(lm_gt); lm_type.append(0); lm_gt.append(np.append(X, 0))
        for k in observers():
            uv, z = proj(T_gt[k], X)
            if z < 0.3 or not (0 < uv[0] < 640 and 0 < uv[1] < 480):
                continue
            o = rng.integers(0, 8); s = scale[o]
            uv = uv + rng.normal(size=2) * s
            if rng.random() < outlier_frac:
                uv = uv + rng.uniform(-40, 40, 2)
            stereo = rng.random() < 0.85
            ur = (uv[0] - P["bf"] / z + rng.normal() * 0.5 * s) if stereo else -1
            stereo = stereo and ur >= 0          # the reference reads the edge class off mvuRight < 0 (src/Optimizer.cc:2063)
            e_kf.append(k); e_obs.append(k); e_lm.append(l); e_type.append(BE_STEREO if stereo else BE_MONO)
            e_meas.append([uv[0], uv[1], ur if stereo else -1, 0]); e_is2.append(1.0 / np.float32(s) ** 2)
    for _ in range(n_lines):
        A = sample_point(); Bp = A + rng.normal(size=3) * 0.4
        la = len(lm_gt); lm_type += [0, 0]; lm_gt += [np.append(A, 0), np.append(Bp, 0)]
        for k in observers():
            kk = k if lines_on_kf is None else lines_on_kf     # the vertex the edge hangs on (and whose view is measured)
            (pa, za), (pb, zb) = proj(T_gt[kk], A), proj(T_gt[kk], Bp)
            if min(za, zb) < 0.3:
                continue
            pa = pa + rng.normal(size=2) * 0.7; pb = pb + rng.normal(size=2) * 0.7
            ln = np.cross(np.append(pa, 1), np.append(pb, 1)); ln /= np.linalg.norm(ln)
            for lmi in (la, la + 1):          # start edge then end edge, consecutive (the reference pairs them)
                e_kf.append(kk); e_obs.append(k); e_lm.append(lmi); e_type.append(BE_LINE); e_meas.append([ln[0], ln[1], ln[2], 0]); e_is2.append(1.0)
    for _ in range(n_planes):
        n = rng.normal(size=3); n /= np.linalg.norm(n); d = rng.uniform(1.0, 4.0)
        l = len(lm_gt); lm_type.append(1); lm_gt.append(np.append(n, d))
        for k in observers():
            R, t = T_gt[k][:3, :3], T_gt[k][:3, 3]
            nc = R @ n; dc = d - t @ nc
            a = rng.normal(size=3); a -= a.dot(nc) * nc; a /= np.linalg.norm(a)
            nm = nc * np.cos(0.004) + a * np.sin(0.004) * rng.normal(); nm /= np.linalg.norm(nm)
            e_kf.append(k); e_obs.append(k); e_lm.append(l); e_type.append(BE_PLANE); e_meas.append([*nm, dc + rng.normal() * 0.004]); e_is2.append(1.0)
            if rng.random() < 0.3:      # an extra parallel / vertical association of the same map plane from this keyframe
                if rng.random() < 0.5:
                    e_kf.append(k); e_obs.append(k); e_lm.append(l); e_type.append(BE_PAR); e_meas.append([*nm, dc + rng.uniform(0.3, 1.0)]); e_is2.append(1.0)
                else:
                    v = np.cross(nm, a); v /= np.linalg.norm(v)
                    e_kf.append(k); e_obs.append(k); e_lm.append(l); e_type.append(BE_VER); e_meas.append([*v, rng.uniform(0.5, 2.0)]); e_is2.append(1.0)
    lm_gt = np.array(lm_gt)
    lm_init = lm_gt.copy()
    pts = np.array(lm_type) == 0
    lm_init[pts, :3] += rng.normal(size=(pts.sum(), 3)) * point_noise
    lm_init[~pts, :3] += rng.normal(size=((~pts).sum(), 3)) * 0.01
    kf_T = T_gt.copy()
    for k in range(K):
        if not fixed[k]:
            kf_T[k, :3, :3] = _rodrigues(rng.normal(size=3) * pose_noise[0]) @ T_gt[k, :3, :3]
            kf_T[k, :3, 3] += rng.normal(size=3) * pose_noise[1]
    return dict(kf_Tcw=kf_T.astype(np.float32).reshape(K, 16), kf_fixed=fixed, lm_type=np.array(lm_type, np.uint8), lm_init=np.ascontiguousarray(lm_init),
                e_kf=np.array(e_kf, np.int32), e_obs_kf=np.array(e_obs, np.int32), e_lm=np.array(e_lm, np.int32), e_type=np.array(e_type, np.uint8),
                e_meas=np.array(e_meas, np.float64), e_inv_sigma2=np.array(e_is2, np.float32), T_gt=T_gt, lm_gt=lm_gt)


# ---- guided matcher problems (SearchByProjection x2, SearchByBoW, line projection, plane association) ----------
def scale_factors(levels=8, s=1.2):
    """ORBextractor::mvScaleFactor (src/ORBextractor.cc:417-424): float32 running product."""
    out = np.ones(levels, np.float32)
    for i in range(1, levels):
        out[i] = np.float32(out[i - 1] * np.float32(s))
    return out


def _flip_bits(rng, desc, max_bits):
    """copy of desc [..., 32] uint8 with up to max_bits random bits flipped per row"""
    out = desc.copy().reshape(-1, 32)
    for r in range(out.shape[0]):
        k = int(rng.integers(0, max_bits + 1))
        for bit in rng.integers(0, 256, k):
            out[r, bit >> 3] ^= np.uint8(1 << (bit & 7))
    return out.reshape(desc.shape)


def guided_frame(B=2, N=1000, stride=None, seed=5, crowd=0.3):
    """The current-frame side: keypoints with depth, stereo coordinate, descriptors, TUM3 intrinsics.
    `crowd` = fraction of keypoints placed in tight clusters (many candidates per search window)."""
    from ._lib import KP_DTYPE
    rng = np.random.default_rng(seed)
    S = stride or N
    K = TUM3
    keys = np.zeros((B, S), KP_DTYPE)
    uR = np.full((B, S), -1, np.float32)
    depth = np.zeros((B, S), np.float32)
    desc = rng.integers(0, 256, (B, S, 32), dtype=np.uint8)
    n = np.zeros(B, np.int32)
    for b in range(B):
        nb = N if b == 0 else int(rng.integers(N // 2, N + 1))
        n[b] = nb
        x = rng.uniform(2, 637, nb); y = rng.uniform(2, 477, nb)
        nc = int(crowd * nb)
        if nc:
            centers = rng.uniform([60, 60], [580, 420], (8, 2))
            which = rng.integers(0, 8, nc)
            x[:nc] = np.clip(centers[which, 0] + rng.normal(0, 12, nc), 0, 639.5)
            y[:nc] = np.clip(centers[which, 1] + rng.normal(0, 12, nc), 0, 479.5)
        keys["x"][b, :nb] = x; keys["y"][b, :nb] = y
        keys["octave"][b, :nb] = np.minimum(rng.geometric(0.35, nb) - 1, 7)
        keys["angle"][b, :nb] = rng.uniform(0, 360, nb)
        keys["size"][b, :nb] = 31
        z = rng.uniform(0.6, 6.0, nb).astype(np.float32)
        depth[b, :nb] = z
        has = rng.random(nb) < 0.85
        uR[b, :nb] = np.where(has, keys["x"][b, :nb] - np.float32(K["bf"]) / z, -1).astype(np.float32)
    return dict(n=n, keys_un=keys, u_right=uR, desc=desc, depth=depth, min_x=0.0, max_x=640.0, min_y=0.0, max_y=480.0, fx=K["fx"], fy=K["fy"],
                cx=K["cx"], cy=K["cy"], bf=K["bf"], b=K["bf"] / K["fx"], scale_factors=scale_factors())


def _se3(rng, rot, trans):
    R = _rodrigues(rng.normal(0, rot, 3))
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = trans
    return T


def guided_last_frame(cur, seed=6, motion=(0.0, 0.0, 0.0), dup=0.15, pix_noise=2.0, bits=30):
    """LastFrame + poses for SearchByProjection(Cur, Last): every last-frame map point is the back-projection of a
    current keypoint (plus `dup` duplicates competing for the same keypoint), seen from a camera displaced by `motion`."""
    rng = np.random.default_rng(seed)
    B, S = cur["keys_un"].shape
    Tc = np.zeros((B, 16), np.float32); Tl = np.zeros((B, 16), np.float32)
    L = S
    out = dict(n=np.zeros(B, np.int32), usable=np.zeros((B, L), np.uint8), xw=np.zeros((B, L, 3), np.float32), octave=np.zeros((B, L), np.int32),
               angle=np.zeros((B, L), np.float32), mp_desc=rng.integers(0, 256, (B, L, 32), dtype=np.uint8), mp_observed=np.zeros((B, L), np.uint8))
    for b in range(B):
        nb = int(cur["n"][b])
        Tcw = _se3(rng, 0.2, rng.normal(0, 0.5, 3))
        Tlw = _se3(rng, 0.01, np.asarray(motion, float) + rng.normal(0, 0.005, 3)) @ Tcw
        Tc[b] = Tcw.astype(np.float32).ravel(); Tl[b] = Tlw.astype(np.float32).ravel()
        src = rng.permutation(nb)
        nd = int(dup * nb)
        src = np.concatenate([src[: nb - nd], rng.choice(src[: max(1, nb - nd)], nd)]) if nd else src
        src = src[rng.permutation(len(src))]
        k = cur["keys_un"][b, src]
        z = cur["depth"][b, src].astype(np.float64)
        u = k["x"] + rng.normal(0, pix_noise, len(src)); v = k["y"] + rng.normal(0, pix_noise, len(src))
        Xc = np.stack([(u - cur["cx"]) * z / cur["fx"], (v - cur["cy"]) * z / cur["fy"], z], 1)
        Xw = (np.linalg.inv(Tcw) @ np.concatenate([Xc, np.ones((len(src), 1))], 1).T).T[:, :3]
        m = len(src)
        out["n"][b] = m
        out["xw"][b, :m] = Xw
        out["usable"][b, :m] = rng.random(m) < 0.9
        out["octave"][b, :m] = np.clip(k["octave"] + rng.integers(-1, 2, m), 0, 7)
        out["angle"][b, :m] = (k["angle"] + rng.normal(0, 6, m) + (rng.random(m) < 0.08) * rng.uniform(0, 360, m)) % 360
        out["mp_desc"][b, :m] = _flip_bits(rng, cur["desc"][b, src], bits)
        out["mp_observed"][b, :m] = rng.random(m) < 0.8
    cur = dict(cur); cur["Tcw"] = Tc
    cur["blocked"] = (rng.random((B, S)) < 0.05).astype(np.uint8)
    out["Tcw"] = Tl
    return cur, out


def guided_map_probes(frame, seed=8, n_probes=2000, pix_noise=1.5, bits=30):
    """Local-map probes for SearchByProjection(F, vpMapPoints, th): the fields Frame::isInFrustum leaves on a MapPoint."""
    rng = np.random.default_rng(seed)
    B, S = frame["keys_un"].shape
    P = n_probes
    out = dict(n=np.zeros(B, np.int32), in_view=np.zeros((B, P), np.uint8), proj_x=np.zeros((B, P), np.float32), proj_y=np.zeros((B, P), np.float32),
               proj_xr=np.zeros((B, P), np.float32), level=np.zeros((B, P), np.int32), view_cos=np.zeros((B, P), np.float32),
               desc=rng.integers(0, 256, (B, P, 32), dtype=np.uint8), observed=np.zeros((B, P), np.uint8))
    for b in range(B):
        nb = int(frame["n"][b])
        m = P if b == 0 else int(rng.integers(P // 2, P + 1))
        src = rng.integers(0, nb, m)
        k = frame["keys_un"][b, src]
        z = frame["depth"][b, src]
        out["n"][b] = m
        out["in_view"][b, :m] = rng.random(m) < 0.9
        out["proj_x"][b, :m] = k["x"] + rng.normal(0, pix_noise, m)
        out["proj_y"][b, :m] = k["y"] + rng.normal(0, pix_noise, m)
        out["proj_xr"][b, :m] = out["proj_x"][b, :m] - np.float32(frame["bf"]) / z
        out["level"][b, :m] = np.clip(k["octave"] + rng.integers(0, 2, m), 0, 7)
        out["view_cos"][b, :m] = rng.uniform(0.99, 1.0, m)
        out["desc"][b, :m] = _flip_bits(rng, frame["desc"][b, src], bits)
        out["observed"][b, :m] = rng.random(m) < 0.8
    frame = dict(frame)
    frame["blocked"] = (rng.random((B, S)) < 0.05).astype(np.uint8)
    return frame, out


def guided_bow(B=2, N=1000, seed=9, n_nodes=90, bits=18):
    """Key frame / frame pair for SearchByBoW: one vocabulary node id per feature (DBoW2 level-4 nodes: ~100 ids)."""
    rng = np.random.default_rng(seed)
    kf = dict(n=np.zeros(B, np.int32), node=np.full((B, N), -1, np.int32), usable=np.zeros((B, N), np.uint8), angle=np.zeros((B, N), np.float32),
              desc=rng.integers(0, 256, (B, N, 32), dtype=np.uint8))
    f = dict(n=np.zeros(B, np.int32), node=np.full((B, N), -1, np.int32), angle=np.zeros((B, N), np.float32),
             desc=rng.integers(0, 256, (B, N, 32), dtype=np.uint8))
    for b in range(B):
        nk = N if b == 0 else int(rng.integers(N // 2, N + 1))
        nf = N if b == 0 else int(rng.integers(N // 2, N + 1))
        kf["n"][b], f["n"][b] = nk, nf
        kf["node"][b, :nk] = rng.integers(-1, n_nodes, nk) * 7 + 11
        kf["node"][b, :nk][kf["node"][b, :nk] < 11] = -1
        kf["usable"][b, :nk] = rng.random(nk) < 0.8
        kf["angle"][b, :nk] = rng.uniform(0, 360, nk)
        # most frame features are noisy copies of key-frame features (same node), some twice
        src = rng.integers(0, nk, nf)
        f["node"][b, :nf] = kf["node"][b, src]
        f["desc"][b, :nf] = _flip_bits(rng, kf["desc"][b, src], bits)
        f["angle"][b, :nf] = (kf["angle"][b, src] + rng.normal(0, 5, nf) + (rng.random(nf) < 0.1) * rng.uniform(0, 360, nf)) % 360
        stray = rng.random(nf) < 0.15
        f["node"][b, :nf][stray] = rng.integers(0, n_nodes, int(stray.sum())) * 7 + 11
    return kf, f


def guided_lines(B=3, n_lines=40, n_ml=120, seed=10, bits=30):
    """Frame lines + projected map lines for LSDmatcher::SearchByProjection."""
    from ._lib import KEYLINE_DTYPE
    rng = np.random.default_rng(seed)
    kl = np.zeros((B, n_lines), KEYLINE_DTYPE)
    lines = dict(n=np.zeros(B, np.int32), keylines=kl, ldesc=rng.integers(0, 256, (B, n_lines, 32), dtype=np.uint8),
                 blocked=(rng.random((B, n_lines)) < 0.05).astype(np.uint8))
    ml = dict(n=np.zeros(B, np.int32), in_view=np.zeros((B, n_ml), np.uint8), proj=np.zeros((B, n_ml, 4), np.float32), level=np.zeros((B, n_ml), np.int32),
              view_cos=np.zeros((B, n_ml), np.float32), desc=rng.integers(0, 256, (B, n_ml, 32), dtype=np.uint8), observed=np.zeros((B, n_ml), np.uint8))
    for b in range(B):
        nl = n_lines if b == 0 else int(rng.integers(n_lines // 2, n_lines + 1))
        lines["n"][b] = nl
        kl["pt_x"][b, :nl] = rng.uniform(20, 620, nl); kl["pt_y"][b, :nl] = rng.uniform(20, 460, nl)
        kl["angle"][b, :nl] = rng.uniform(-np.pi, np.pi, nl)
        kl["octave"][b, :nl] = rng.integers(0, 3, nl) * (b % 2)   # frame 0: all octave 0 like the reference's 1-octave LSD
        kl["line_length"][b, :nl] = rng.uniform(30, 200, nl)
        m = n_ml if b == 0 else int(rng.integers(n_ml // 2, n_ml + 1))
        ml["n"][b] = m
        src = rng.integers(0, nl, m)
        ang = kl["angle"][b, src] + rng.normal(0, 0.3, m)
        half = rng.uniform(15, 100, m)
        cx = kl["pt_x"][b, src] + rng.normal(0, 2.0, m); cy = kl["pt_y"][b, src] + rng.normal(0, 2.0, m)
        ml["proj"][b, :m] = np.stack([cx - half * np.cos(ang), cy - half * np.sin(ang), cx + half * np.cos(ang), cy + half * np.sin(ang)], 1)
        ml["in_view"][b, :m] = rng.random(m) < 0.9
        ml["level"][b, :m] = np.clip(kl["octave"][b, src] + rng.integers(0, 2, m), 0, 7)
        ml["view_cos"][b, :m] = rng.uniform(0.99, 1.0, m)
        ml["desc"][b, :m] = _flip_bits(rng, lines["ldesc"][b, src], bits)
        ml["observed"][b, :m] = rng.random(m) < 0.7
    return lines, ml


def guided_planes(B=4, n_planes=8, n_map=40, n_pts=600, seed=12, shared=False):
    """Frame planes (camera frame) + map planes (world frame, with boundary clouds) for PlaneMatcher::SearchMapByCoefficients."""
    rng = np.random.default_rng(seed)
    MB = 1 if shared else B
    mp = dict(n=np.zeros(MB, np.int32), valid=np.zeros((MB, n_map), np.uint8), coef=np.zeros((MB, n_map, 4), np.float32),
              npts=np.zeros((MB, n_map), np.int32), pts=np.zeros((MB, n_map, n_pts, 3), np.float32), shared=shared)
    for m in range(MB):
        nm = n_map if m == 0 else int(rng.integers(n_map // 2, n_map + 1))
        mp["n"][m] = nm
        axes = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        for j in range(nm):
            nrm = axes[j % 3] * rng.choice([-1, 1]) + rng.normal(0, 0.02 if j % 4 else 0.3, 3)
            nrm /= np.linalg.norm(nrm)
            d = rng.uniform(-3, 3)
            mp["coef"][m, j] = [*nrm, d]
            k = int(rng.integers(0, n_pts + 1)) if j % 7 == 0 else n_pts
            mp["npts"][m, j] = k
            # points on the plane (n.x + d = 0) with a little noise
            t1 = np.cross(nrm, [1, 0, 0.3]); t1 /= np.linalg.norm(t1); t2 = np.cross(nrm, t1)
            ab = rng.uniform(-2, 2, (k, 2))
            mp["pts"][m, j, :k] = (-d * nrm)[None] + ab[:, :1] * t1 + ab[:, 1:] * t2 + rng.normal(0, 0.01, (k, 3))
            mp["valid"][m, j] = rng.random() < 0.9
    fr = dict(n=np.zeros(B, np.int32), coef=np.zeros((B, n_planes, 4), np.float32), Tcw=np.zeros((B, 16), np.float32))
    for b in range(B):
        m = 0 if shared else b
        T = _se3(rng, 0.3, rng.normal(0, 0.5, 3))
        fr["Tcw"][b] = T.astype(np.float32).ravel()
        np_ = n_planes if b == 0 else int(rng.integers(1, n_planes + 1))
        fr["n"][b] = np_
        for i in range(np_):
            j = int(rng.integers(0, mp["n"][m]))
            pw = mp["coef"][m, j].astype(np.float64) + np.r_[rng.normal(0, 0.02, 3), rng.normal(0, 0.03)]
            pw[:3] /= np.linalg.norm(pw[:3])
            # camera-frame coefficients: pi_c = Tcw^-T pi_w  (so that Tcw^T pi_c = pi_w)
            fr["coef"][b, i] = np.linalg.inv(T).T @ pw
    return fr, mp


def guided_local_map(frame, seed=14, n_points=2500, n_lines=300):
    """Local-map points / lines in world coordinates around a posed frame, for Frame::isInFrustum: most project into the image,
    some fall behind the camera, outside the image, outside the scale-invariance range or are seen too obliquely."""
    rng = np.random.default_rng(seed)
    B = frame["keys_un"].shape[0]
    Tcw = np.zeros((B, 16), np.float32)
    mp = dict(n=np.zeros(B, np.int32), valid=np.zeros((B, n_points), np.uint8), xw=np.zeros((B, n_points, 3), np.float32),
              normal=np.zeros((B, n_points, 3), np.float32), min_dist=np.zeros((B, n_points), np.float32), max_dist=np.zeros((B, n_points), np.float32),
              desc=rng.integers(0, 256, (B, n_points, 32), dtype=np.uint8), observed=(rng.random((B, n_points)) < 0.8).astype(np.uint8))
    ml = dict(n=np.zeros(B, np.int32), valid=np.zeros((B, n_lines), np.uint8), xw6=np.zeros((B, n_lines, 6)), normal=np.zeros((B, n_lines, 3)),
              min_dist=np.zeros((B, n_lines), np.float32), max_dist=np.zeros((B, n_lines), np.float32),
              desc=rng.integers(0, 256, (B, n_lines, 32), dtype=np.uint8), observed=(rng.random((B, n_lines)) < 0.8).astype(np.uint8))
    for b in range(B):
        T = _se3(rng, 0.3, rng.normal(0, 0.5, 3))
        Tcw[b] = T.astype(np.float32).ravel()
        Twc = np.linalg.inv(T)
        Ow = Twc[:3, 3]

        def world(n):
            u = rng.uniform(-120, 760, n); v = rng.uniform(-100, 580, n); z = rng.uniform(-1.0, 7.0, n)
            z[np.abs(z) < 0.2] = 0.5
            Xc = np.stack([(u - frame["cx"]) * z / frame["fx"], (v - frame["cy"]) * z / frame["fy"], z, np.ones(n)], 1)
            return (Twc @ Xc.T).T[:, :3]
        m = n_points if b == 0 else int(rng.integers(n_points // 2, n_points + 1))
        X = world(m)
        d = np.linalg.norm(X - Ow, axis=1)
        nrm = (Ow - X) / d[:, None] * -1.0            # PO = P - Ow; make the stored normal roughly parallel to PO, with outliers
        nrm = nrm + rng.normal(0, 0.5, nrm.shape); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        mp["n"][b] = m; mp["valid"][b, :m] = rng.random(m) < 0.95
        mp["xw"][b, :m] = X; mp["normal"][b, :m] = nrm
        mp["max_dist"][b, :m] = d * rng.uniform(0.6, 4.0, m); mp["min_dist"][b, :m] = mp["max_dist"][b, :m] / (1.2 ** 7) * rng.uniform(0.8, 1.5, m)
        k = n_lines if b == 0 else int(rng.integers(n_lines // 2, n_lines + 1))
        S = world(k); E = S + rng.normal(0, 0.4, (k, 3))
        M = 0.5 * (S + E); dm = np.linalg.norm(M - Ow, axis=1)
        ln = (M - Ow) / dm[:, None] + rng.normal(0, 0.5, (k, 3)); ln /= np.linalg.norm(ln, axis=1, keepdims=True)
        ml["n"][b] = k; ml["valid"][b, :k] = rng.random(k) < 0.95
        ml["xw6"][b, :k] = np.concatenate([S, E], 1); ml["normal"][b, :k] = ln
        ml["max_dist"][b, :k] = dm * rng.uniform(0.6, 4.0, k); ml["min_dist"][b, :k] = ml["max_dist"][b, :k] / (1.2 ** 7) * rng.uniform(0.8, 1.5, k)
    frame = dict(frame); frame["Tcw"] = Tcw
    return frame, mp, ml


def guided_fuse_points(frame, seed=15, n_points=2500, hit=0.6, bits=60, pix_noise=1.5, inv_level_sigma2=None):
    """Map points to fuse into posed key frames (ORBmatcher::Fuse): `hit` of them are back-projections of the key frame's own keypoints
    (descriptor = the keypoint's with up to `bits` flipped, pixel noise, a predicted level at or one above the keypoint's octave), the rest the
    generic local map of guided_local_map (behind the camera, outside the image, out of range, oblique).  Returns (frame with Tcw, mp) with
    mp: n, usable, xw, normal, min_dist, max_dist, desc [+ observations for the map edits]."""
    rng = np.random.default_rng(seed)
    frame, mp, _ = guided_local_map(frame, seed=seed + 1, n_points=n_points, n_lines=4)
    B = frame["keys_un"].shape[0]
    mp = {k: v.copy() for k, v in mp.items()}
    mp["usable"] = mp.pop("valid")
    for b in range(B):
        T = frame["Tcw"][b].reshape(4, 4).astype(np.float64)
        Twc = np.linalg.inv(T); Ow = Twc[:3, 3]
        m, nk = int(mp["n"][b]), int(frame["n"][b])
        sel = np.flatnonzero(rng.random(m) < hit)
        kp = rng.integers(0, nk, len(sel))
        keys = frame["keys_un"][b]
        ur = frame["u_right"][b, kp]
        z = np.where(ur >= 0, frame["bf"] / np.maximum(keys["x"][kp] - ur, 1e-3), rng.uniform(0.6, 6.0, len(sel)))
        u = keys["x"][kp] + rng.normal(0, pix_noise, len(sel)); v = keys["y"][kp] + rng.normal(0, pix_noise, len(sel))
        Xc = np.stack([(u - frame["cx"]) * z / frame["fx"], (v - frame["cy"]) * z / frame["fy"], z, np.ones(len(sel))], 1)
        X = (Twc @ Xc.T).T[:, :3]
        d = np.linalg.norm(X - Ow, axis=1)
        nrm = (X - Ow) / d[:, None] + rng.normal(0, 0.25, X.shape); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        lvl = keys["octave"][kp] + rng.integers(0, 2, len(sel))                   # the key point's level must be lvl - 1 or lvl
        mx = d * 1.2 ** (lvl - rng.uniform(0.05, 0.95, len(sel)))                  # PredictScale = ceil(log(max / d) / log 1.2) = lvl
        mp["xw"][b, sel] = X; mp["normal"][b, sel] = nrm; mp["max_dist"][b, sel] = mx; mp["min_dist"][b, sel] = mx / 1.2 ** 7 * 0.5
        mp["desc"][b, sel] = _flip_bits(rng, frame["desc"][b, kp], bits)
    mp["observations"] = rng.integers(1, 9, mp["usable"].shape).astype(np.int32)
    return frame, mp


def guided_fuse_lines(B=3, n_lines=60, n_ml=300, seed=17, hit=0.6, bits=60):
    """Key frames with key lines + map lines to fuse into them (LSDmatcher::Fuse): `hit` of the map lines are back-projections of a key line's end
    points (descriptor with up to `bits` flipped, predicted level at or one above the key line's octave), the rest random segments around the
    camera (behind it, outside the image, out of range, oblique).  Returns (kf dict: Tcw + intrinsics + scale factors, lines dict, ml dict)."""
    from ._lib import KEYLINE_DTYPE
    rng = np.random.default_rng(seed)
    K = TUM3
    kf = dict(B=B, Tcw=np.zeros((B, 16), np.float32), min_x=0.0, max_x=640.0, min_y=0.0, max_y=480.0, fx=K["fx"], fy=K["fy"], cx=K["cx"], cy=K["cy"], bf=K["bf"],
              b=K["bf"] / K["fx"], scale_factors=scale_factors())
    kl = np.zeros((B, n_lines), KEYLINE_DTYPE)
    lines = dict(n=np.zeros(B, np.int32), keylines=kl, ldesc=rng.integers(0, 256, (B, n_lines, 32), dtype=np.uint8))
    ml = dict(n=np.zeros(B, np.int32), usable=np.zeros((B, n_ml), np.uint8), xw6=np.zeros((B, n_ml, 6)), normal=np.zeros((B, n_ml, 3)),
              min_dist=np.zeros((B, n_ml), np.float32), max_dist=np.zeros((B, n_ml), np.float32), desc=rng.integers(0, 256, (B, n_ml, 32), dtype=np.uint8),
              observations=rng.integers(1, 9, (B, n_ml)).astype(np.int32))
    for b in range(B):
        T = _se3(rng, 0.3, rng.normal(0, 0.5, 3))
        kf["Tcw"][b] = T.astype(np.float32).ravel()
        Twc = np.linalg.inv(T); Ow = Twc[:3, 3]
        nl = n_lines if b == 0 else int(rng.integers(n_lines // 2, n_lines + 1))
        lines["n"][b] = nl
        cx_ = rng.uniform(60, 580, nl); cy_ = rng.uniform(60, 420, nl); ang = rng.uniform(-1.2, 1.2, nl); half = rng.uniform(15, 55, nl)
        kl["start_x"][b, :nl] = cx_ - half * np.cos(ang); kl["start_y"][b, :nl] = cy_ - half * np.sin(ang)
        kl["end_x"][b, :nl] = cx_ + half * np.cos(ang); kl["end_y"][b, :nl] = cy_ + half * np.sin(ang)
        kl["pt_x"][b, :nl] = cx_; kl["pt_y"][b, :nl] = cy_
        # GetLinesInArea compares the SLOPE of the projected segment with KeyLine::angle, one-sided (slope - angle <= r / 100)
        kl["angle"][b, :nl] = np.tan(ang) + rng.uniform(-0.02, 0.6, nl)
        kl["octave"][b, :nl] = rng.integers(0, 3, nl) * (b % 2)          # key frame 0: one octave, like the reference's LSD
        kl["line_length"][b, :nl] = 2 * half

        def back(u, v, z):
            Xc = np.stack([(u - K["cx"]) * z / K["fx"], (v - K["cy"]) * z / K["fy"], z, np.ones(len(u))], 1)
            return (Twc @ Xc.T).T[:, :3]
        m = n_ml if b == 0 else int(rng.integers(n_ml // 2, n_ml + 1))
        ml["n"][b] = m
        ml["usable"][b, :m] = rng.random(m) < 0.93
        # misses: random segments
        u = rng.uniform(-120, 760, m); v = rng.uniform(-100, 580, m); z = rng.uniform(-1.0, 7.0, m); z[np.abs(z) < 0.2] = 0.5
        S = back(u, v, z); E = S + rng.normal(0, 0.3, (m, 3))
        src = rng.integers(0, nl, m)
        isel = rng.random(m) < hit
        zs = rng.uniform(0.8, 5.0, m); ze = zs + rng.normal(0, 0.15, m)
        S[isel] = back(kl["start_x"][b, src] + rng.normal(0, 1.0, m), kl["start_y"][b, src] + rng.normal(0, 1.0, m), zs)[isel]
        E[isel] = back(kl["end_x"][b, src] + rng.normal(0, 1.0, m), kl["end_y"][b, src] + rng.normal(0, 1.0, m), ze)[isel]
        M = 0.5 * (S + E); dm = np.linalg.norm(M - Ow, axis=1)
        nrm = (M - Ow) / dm[:, None] + rng.normal(0, 0.3, (m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        lvl = kl["octave"][b, src] + rng.integers(0, 2, m)
        mx = np.where(isel, dm * 1.2 ** (lvl - rng.uniform(0.05, 0.95, m)), dm * rng.uniform(0.4, 4.5, m))   # misses: some predicted levels out of the pyramid
        ml["xw6"][b, :m] = np.concatenate([S, E], 1); ml["normal"][b, :m] = nrm
        ml["max_dist"][b, :m] = mx; ml["min_dist"][b, :m] = mx / 1.2 ** 7 * np.where(isel, 0.5, rng.uniform(0.5, 1.6, m))
        d = ml["desc"][b, :m]; d[isel] = _flip_bits(rng, lines["ldesc"][b, src[isel]], bits); ml["desc"][b, :m] = d
    return kf, lines, ml


def manhattan_scene(B=4, n_normals=8500, n_lines=40, seed=21, tilt_deg=4.0, noise=0.04, clutter=0.25, drop_axis=None):
    """Surface normals and vanishing directions of a Manhattan world seen from B cameras (Tracking::TrackManhattanFrame input).

    Per frame: a true rotation R_true (camera <- Manhattan frame), `n_normals` unit normals (float32, as PCL's integral-image
    normals feed `Frame::vSurfaceNormal`) clustered around +-columns of R_true plus `clutter` uniformly random ones, `n_lines`
    3-D line directions (float64, `FrameLine::direction`), and the previous estimate R_last = R_true perturbed by `tilt_deg`.
    drop_axis (0..2, or a tuple of them): those axes get no support (exercises the two-axes + cross-product branch, and the one-axis case).  Counts are ragged."""
    rng = np.random.default_rng(seed)

    def rand_rot(max_deg):
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        a = np.deg2rad(rng.uniform(0.3 * max_deg, max_deg))
        K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)

    normals = np.zeros((B, n_normals, 3), np.float32)
    lines = np.zeros((B, n_lines, 3), np.float64)
    nn = np.zeros(B, np.int32); nl = np.zeros(B, np.int32)
    R_true = np.zeros((B, 3, 3), np.float32); R_last = np.zeros((B, 3, 3), np.float32)
    for b in range(B):
        Rt = rand_rot(180.0)
        R_true[b] = Rt.astype(np.float32)
        R_last[b] = (Rt @ rand_rot(tilt_deg)).astype(np.float32)
        n = int(n_normals * rng.uniform(0.6, 1.0)) if b else n_normals
        m = int(n_lines * rng.uniform(0.3, 1.0)) if b else n_lines
        nn[b], nl[b] = n, m
        dropped = () if drop_axis is None else (tuple(drop_axis) if isinstance(drop_axis, (tuple, list)) else (drop_axis,))
        axes = [a for a in range(3) if a not in dropped]
        pick = rng.choice(axes, size=n)
        sign = rng.choice([-1.0, 1.0], size=n)
        v = Rt[:, pick].T * sign[:, None] + rng.normal(scale=noise, size=(n, 3))
        rnd = rng.random(n) < clutter
        v[rnd] = rng.normal(size=(int(rnd.sum()), 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        normals[b, :n] = v.astype(np.float32)
        pick = rng.choice(axes, size=m)
        d = Rt[:, pick].T * rng.choice([-1.0, 1.0], size=m)[:, None] + rng.normal(scale=noise * 0.5, size=(m, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        lines[b, :m] = d
    return dict(normals=normals, n_normals=nn, lines=lines, n_lines=nl, R_last=R_last, R_true=R_true)


def ba_local_only(prob):
    """Drop the edges of landmarks no optimised (non-fixed) keyframe observes: Optimizer::LocalBundleAdjustment collects its landmarks
    from the local keyframes' matches (src/Optimizer.cc:1869-1921), so such landmarks never enter the reference's graph."""
    fixed = np.asarray(prob["kf_fixed"]).astype(bool)
    seen = np.zeros(len(prob["lm_type"]), bool)
    np.logical_or.at(seen, prob["e_lm"], ~fixed[prob["e_obs_kf"]])
    # the two end points of a line are one map object
    is_line = np.zeros(len(seen), bool); is_line[prob["e_lm"][prob["e_type"] == BE_LINE]] = True
    idx = np.nonzero(is_line)[0]
    for a, b in zip(idx[0::2], idx[1::2]):
        seen[a] = seen[b] = seen[a] or seen[b]
    keep = seen[prob["e_lm"]]
    out = dict(prob)
    for k in ("e_kf", "e_obs_kf", "e_lm", "e_type", "e_meas", "e_inv_sigma2"):
        out[k] = np.ascontiguousarray(prob[k][keep])
    return out


# ---- camera streams for bench.py: larger canvases a 640x480 window pans over (consecutive frames of a stream overlap like video) --------
def _canvas_job(args):
    kind, seed, w, h = args
    return gray_image(seed, w, h) if kind == 0 else depth_image(seed, w, h)


def stream_canvases(P, seed, w, h, procs=None):
    """P gray (uint8) and P depth (uint16) canvases of w x h, generated on `procs` worker processes (call before CUDA is initialised)."""
    import multiprocessing as mp
    import os
    jobs = [(0, 1234 + seed * 4096 + i, w, h) for i in range(P)] + [(1, 4321 + seed * 4096 + i, w, h) for i in range(P)]
    procs = procs or min(32, os.cpu_count() or 1)
    if procs <= 1:
        res = [_canvas_job(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_canvas_job, jobs, chunksize=max(1, len(jobs) // (4 * procs)))
    return np.stack(res[:P]), np.stack(res[P:])


def pan_offset(i, margin=48, amp=20):
    """Top-left corner of the window at step i: a slow Lissajous path inside the margin (<= ~8 px per step)."""
    ox = margin // 2 + int(round(amp * np.sin(2 * np.pi * i / 17.0)))
    oy = margin // 2 + int(round(amp * np.sin(2 * np.pi * i / 23.0 + 1.0)))
    return ox, oy


def vocabulary(k=10, L=3, seed=77, bits=40, stop_frac=0.02):
    """A synthetic DBoW2 vocabulary tree (complete k-ary, depth L) in the node order of the text format TemplatedVocabulary::loadFromTextFile reads
    (node ids 1.. in file order, breadth first; word ids in order of leaf appearance).  A child's descriptor is its parent's with `bits` random bit flips;
    leaf weights are idf-like positive doubles, a few words are stopped (weight 0).  Returns dict(k, L, parent [n], is_leaf [n], desc [n,32], weight [n])
    for nodes 1..n (node 0, the root, is implicit)."""
    rng = np.random.default_rng(seed)
    parent, is_leaf, desc, weight = [], [], [], []
    frontier = [(0, rng.integers(0, 256, 32, dtype=np.uint8))]
    nid = 0
    for level in range(1, L + 1):
        nxt = []
        for pid, pdesc in frontier:
            for _ in range(k):
                nid += 1
                d = _flip_bits(rng, pdesc[None], bits if level > 1 else 128)[0]
                leaf = level == L
                parent.append(pid); is_leaf.append(1 if leaf else 0); desc.append(d)
                weight.append(0.0 if not leaf or rng.random() < stop_frac else float(rng.uniform(0.5, 9.0)))
                nxt.append((nid, d))
        frontier = nxt
    return dict(k=k, L=L, parent=np.array(parent, np.int32), is_leaf=np.array(is_leaf, np.uint8), desc=np.stack(desc), weight=np.array(weight, np.float64))


def write_vocabulary_text(voc, path):
    """ORBvoc.txt's format: header 'k L scoring weighting' (0 0 = L1_NORM, TF_IDF), then per node 'parent isLeaf d0 .. d31 weight'.  No trailing newline: the
    reference's reader would append a node for an empty last line."""
    lines = [f"{voc['k']} {voc['L']} 0 0"]
    for p, lf, d, w in zip(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"]):
        lines.append(f"{int(p)} {int(lf)} " + " ".join(str(int(x)) for x in d) + f" {float(w)!r}")
    with open(path, "w") as f:
        f.write("\n".join(lines))


def vocabulary_queries(voc, n=1000, seed=5, bits=25):
    """Descriptors near the vocabulary's leaves (noisy copies), so the tree descent is not arbitrary; some leaves are hit several times."""
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["is_leaf"])[0]
    pick = leaves[rng.integers(0, len(leaves), n)]
    pick[: n // 5] = pick[n // 5: 2 * (n // 5)]          # repeated words: term frequency > 1
    return _flip_bits(rng, voc["desc"][pick], bits)
