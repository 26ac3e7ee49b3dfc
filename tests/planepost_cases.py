"""Seeded inputs shared by the CPU and GPU tests of the plane post-processing (Frame::ComputePlanes head, MaxPointDistanceFromPlane)."""
import numpy as np


def plane_cloud(seed, n=600, th=0.05, spread=0.9, extent=2.0, outliers=0):
    """n points on a random plane through ~(0, 0, 2) with uniform noise of +-spread*th along the normal (all pass the distance gate for spread < 1);
    `outliers` extra points at 0.98 th.  -> (plane [4] f32 with unit normal, pts [n,3] f32)"""
    rng = np.random.default_rng(seed)
    nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
    c = np.array([0.0, 0.0, 2.0]) + rng.normal(scale=0.2, size=3)
    a = np.cross(nrm, [1.0, 0.0, 0.0]); a /= np.linalg.norm(a); b = np.cross(nrm, a)
    uv = rng.uniform(-extent / 2, extent / 2, size=(n + outliers, 2))
    off = rng.uniform(-spread * th, spread * th, size=n + outliers)
    off[n:] = 0.98 * th * np.sign(off[n:])
    pts = c + uv[:, :1] * a + uv[:, 1:] * b + off[:, None] * nrm
    plane = np.r_[nrm, -nrm @ c].astype(np.float32)
    return plane, pts.astype(np.float32)


def refit_cases():
    cases = []
    for i, (n, th, spread) in enumerate([(600, 0.05, 0.9), (300, 0.03, 0.95), (2500, 0.05, 0.5), (64, 0.05, 0.9), (5, 0.05, 0.5), (1200, 0.01, 0.97)]):
        plane, pts = plane_cloud(100 + i, n=n, th=th, spread=spread)
        cases.append(dict(name=f"noisy{n}_{th}", plane=plane, pts=pts, th=th))
    plane, pts = plane_cloud(200, n=400, th=0.05, spread=0.3, outliers=40)
    cases.append(dict(name="outliers", plane=plane, pts=pts, th=0.05))
    plane, pts = plane_cloud(201, n=400, th=0.05, spread=0.5)
    pts[17] += 0.2 * plane[:3]
    cases.append(dict(name="gate_fails", plane=plane, pts=pts, th=0.05))
    plane, pts = plane_cloud(202, n=2, th=0.05)
    cases.append(dict(name="two_points", plane=plane, pts=pts, th=0.05))
    # collinear points along a diagonal: the three coordinate ratios are equal, no good sample in 1000 tries
    t = np.linspace(-1, 1, 50, dtype=np.float32)
    cases.append(dict(name="collinear_diag", plane=np.array([0.70710678, -0.70710678, 0, 0], np.float32), pts=np.stack([t, t, t], 1), th=0.05))
    # collinear along an axis: 0/0 = NaN makes isSampleGood accept every sample; the models are NaN, nothing is ever an inlier
    cases.append(dict(name="collinear_axis", plane=np.array([0, 0, 1, -2], np.float32), pts=np.stack([t, np.zeros_like(t), np.full_like(t, 2.0)], 1), th=0.05))
    # flipped sign of d: the refit must come back with the sign of the input
    plane, pts = plane_cloud(203, n=500, th=0.05, spread=0.6)
    cases.append(dict(name="negated", plane=-plane, pts=pts, th=0.05))
    return cases


def world_points(seed, n=3000):
    rng = np.random.default_rng(seed)
    return rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)


def pose(seed):
    rng = np.random.default_rng(seed)
    w = rng.normal(scale=0.2, size=3); th = np.linalg.norm(w); k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.normal(scale=0.5, size=3)
    return T
