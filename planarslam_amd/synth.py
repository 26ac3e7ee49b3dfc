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
