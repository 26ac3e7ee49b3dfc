"""The product's wavefront eigen-solver (planarslam_amd/csrc/peac_eig.h: one instruction stream for 64 lanes) against the oracle's
restatement of Eigen's SelfAdjointEigenSolver<Matrix3d> (oracle/eigprim.cpp), bit for bit, on the CPU: the header compiles with g++."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "host_shim", "libpeac_eig_host.so")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(ROOT, "tests", "host_shim", "peac_eig_host.cpp")
    deps = [src, os.path.join(ROOT, "planarslam_amd", "csrc", "peac_eig.h"), os.path.join(ROOT, "oracle", "eigprim.cpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", SO, src, os.path.join(ROOT, "oracle", "eigprim.cpp")])
    L = ctypes.CDLL(SO)
    L.peac_eig_compare.restype = ctypes.c_long
    L.peac_eig_compare.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    return L


def _covariances(rng, n):
    """scatter matrices of noisy planar patches (what PlaneSeg::Stats::compute feeds the solver), lower triangles"""
    out = np.empty((n, 6))
    for i in range(n):
        N = int(rng.integers(4, 4000))
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        basis = np.linalg.svd(nrm[None])[2][1:]
        pts = (rng.normal(size=(N, 2)) * rng.uniform(0.01, 2.0, 2)) @ basis + nrm * rng.normal(size=(N, 1)) * rng.choice([0.0, 1e-6, 1e-3, 0.05]) + rng.normal(size=3) * 3
        if rng.random() < 0.2:
            pts = np.rint(pts * 5000) / 5000
        s = pts.sum(0); K = pts.T @ pts - np.outer(s, s) / N
        out[i] = [K[0, 0], K[1, 0], K[1, 1], K[2, 0], K[2, 1], K[2, 2]]
    return out


def test_wavefront_solver_is_bit_identical_to_the_oracle(lib):
    rng = np.random.default_rng(7)
    sets = [_covariances(rng, 4000)]
    sets.append(rng.normal(size=(200000, 6)) * np.exp(rng.normal(size=(200000, 1)) * 6))            # generic symmetric
    d = np.zeros((20000, 6)); d[:, [0, 2, 5]] = rng.normal(size=(20000, 3)); sets.append(d)          # diagonal
    t = rng.normal(size=(20000, 6)); t[:, 3] = 0; sets.append(t)                                    # already tridiagonal (a20 == 0)
    z = rng.normal(size=(20000, 6)); z[rng.random(z.shape) < 0.4] = 0; sets.append(z)               # many exact zeros
    r1 = rng.normal(size=(20000, 3)); sets.append(np.stack([r1[:, 0] ** 2, r1[:, 1] * r1[:, 0], r1[:, 1] ** 2, r1[:, 2] * r1[:, 0], r1[:, 2] * r1[:, 1], r1[:, 2] ** 2], 1))  # rank one
    e = np.ones((1000, 6)) * rng.normal(size=(1000, 1)); sets.append(e)                             # all entries equal
    sets.append(rng.normal(size=(20000, 6)) * 1e-160)                                               # squares underflow (e2 == 0 branch)
    sets.append(np.zeros((4, 6)))
    q = np.rint(rng.normal(size=(50000, 6)) * 4) / 4; sets.append(q)                                # small dyadic numbers: exact ties, td == 0
    for s in sets:
        s = np.ascontiguousarray(s, np.float64)
        first = ctypes.c_long(-1)
        bad = lib.peac_eig_compare(s.ctypes.data, len(s), ctypes.byref(first))
        assert bad == 0, f"{bad} of {len(s)} matrices differ; first: {s[first.value]}"
