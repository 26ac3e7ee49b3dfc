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
    L.peac_lb_check.restype = ctypes.c_long
    L.peac_lb_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
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


def _moments(rng, n, unit):
    """moments of noisy planar patches at camera distances (unit: 1.0 = metres, 1000.0 = millimetres, as PEAC sees them), float32 points like a depth
    map's back-projection; some thin (line-like), some exactly planar, some tiny"""
    st = np.empty((n, 9)); Ns = np.empty(n, np.int32)
    for i in range(n):
        N = int(rng.choice([4, 16, 100, 400, 3000, 30000]))
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        basis = np.linalg.svd(nrm[None])[2][1:]
        ext = rng.uniform(0.01, 1.5, 2) * (rng.choice([1.0, 1e-2, 1e-4]), 1.0)
        noise = rng.choice([0.0, 1e-5, 1e-3, 0.01, 0.2])
        pts = (rng.normal(size=(N, 2)) * ext) @ basis + nrm * rng.normal(size=(N, 1)) * noise + np.array([rng.normal(), rng.normal(), rng.uniform(0.5, 6)])
        pts = (pts * unit).astype(np.float32).astype(np.float64)
        s = pts.sum(0); q = (pts * pts).sum(0)
        st[i] = [s[0], s[1], s[2], q[0], q[1], q[2], (pts[:, 0] * pts[:, 1]).sum(), (pts[:, 1] * pts[:, 2]).sum(), (pts[:, 0] * pts[:, 2]).sum()]
        Ns[i] = N
    return st, Ns


def test_mse_lower_bound_never_exceeds_the_solver(lib):
    """merged_mse_lower_bound (peac_eig.h) is what lets the clustering skip eigen-solves: it must never lie above the mse the solver returns, and it
    should be tight where it matters (plane-like regions)."""
    rng = np.random.default_rng(11)
    for unit in (1.0, 1000.0):
        st, Ns = _moments(rng, 6000, unit)
        lb = np.empty(len(st)); mse = np.empty(len(st))
        bad = lib.peac_lb_check(st.ctypes.data, Ns.ctypes.data, len(st), lb.ctypes.data, mse.ctypes.data)
        assert bad == 0, f"{bad} bounds above the solver's mse; e.g. {np.flatnonzero(lb > mse)[:5]}"
        fin = np.isfinite(lb) & (mse > 0)
        assert fin.mean() > 0.5
        assert np.median(lb[fin] / mse[fin]) > 0.98
    # sums of two patches (what a candidate merge is), degenerate inputs
    st, Ns = _moments(rng, 4000, 1000.0)
    st2 = st[:2000] + st[2000:]; N2 = (Ns[:2000] + Ns[2000:]).astype(np.int32)
    z = np.zeros((3, 9)); zN = np.array([1, 4, 100], np.int32)
    for a, b in ((st2, N2), (z, zN)):
        a = np.ascontiguousarray(a); lb = np.empty(len(a)); mse = np.empty(len(a))
        assert lib.peac_lb_check(a.ctypes.data, b.ctypes.data, len(a), lb.ctypes.data, mse.ctypes.data) == 0
