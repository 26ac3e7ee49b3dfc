"""Seeded observation sets for MapPoint::ComputeDistinctiveDescriptors (shared by the fixture generator, the CPU and the GPU tests)."""
import numpy as np


def cases(seed=3, n_points=60):
    """list of desc [n,32] u8: noisy copies of a base descriptor (1..80 observations), duplicates, one outlier-heavy set, one empty set"""
    rng = np.random.default_rng(seed)
    out = []
    for p in range(n_points):
        n = int(rng.integers(1, 81)) if p % 7 else int(rng.integers(1, 4))
        base = rng.integers(0, 256, size=32, dtype=np.uint8)
        d = np.repeat(base[None], n, 0)
        flips = rng.integers(0, 40, size=n)
        for i in range(n):
            bits = rng.choice(256, size=int(flips[i]), replace=False)
            for b in bits:
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        if p % 5 == 0 and n > 3:
            d[rng.integers(0, n)] = rng.integers(0, 256, size=32, dtype=np.uint8)      # a wrong association
        if p % 11 == 0 and n > 2:
            d[n - 1] = d[0]                                                          # exact duplicates: ties between medians
        out.append(d)
    out.append(np.zeros((0, 32), np.uint8))
    out.append(np.repeat(rng.integers(0, 256, size=(1, 32), dtype=np.uint8), 5, 0))   # all equal: every median 0, index 0 wins
    return out
