"""CPU: the wavefront version of LSD's NFA tail (planarslam_amd/csrc/lsd_nfa.h: the (term, tail) recurrence travels through the lanes, one multiplier per lane,
the stopping rule evaluated by the lane of its own iteration) compiled with g++ and run on the wave64 emulator (tests/host_shim/wave_emul.h), bit for bit against the
sequential loop the library runs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "host_shim")
SO = os.path.join(SHIM, "liblsd_nfa_host.so")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(SHIM, "lsd_nfa_host.cpp")
    deps = [src, os.path.join(SHIM, "wave_emul.h"), os.path.join(ROOT, "planarslam_amd", "csrc", "lsd_nfa.h"), os.path.join(ROOT, "planarslam_amd", "csrc", "wave_ops.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", SO, src])
    L = C.CDLL(SO)
    L.lsd_nfa_tail_compare.restype = C.c_long
    L.lsd_nfa_tail_compare.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    return L


def _cases(rng, count, nmax):
    from math import lgamma
    out = []
    while len(out) < count:
        n = int(rng.integers(2, nmax))
        k = int(rng.integers(1, n))                       # k < n: the loop runs
        p = 0.125 / 2 ** int(rng.integers(0, 6))          # LSD's p = ang_th / 180 and its halvings
        log1term = lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1) + k * np.log(p) + (n - k) * np.log(1 - p)
        term = float(np.exp(log1term))
        if term == 0.0:
            continue                                      # nfa() returns before the loop
        out.append((term, n, k, p / (1 - p)))
    return np.array(out, np.float64)


@pytest.mark.parametrize("nmax,count", [(40, 300), (200, 300), (1500, 120), (9000, 25)])
def test_wavefront_nfa_tail_is_the_sequential_loop(lib, nmax, count):
    rng = np.random.default_rng(nmax)
    cases = _cases(rng, count, nmax)
    got = np.zeros(len(cases)); want = np.zeros(len(cases))
    err = C.create_string_buffer(256)
    log_nt = 5 * (np.log10(512.0) + np.log10(384.0)) / 2 + np.log10(11.0)
    bad = lib.lsd_nfa_tail_compare(cases.ctypes.data, len(cases), log_nt, got.ctypes.data, want.ctypes.data, err, 256)
    assert bad >= 0, err.value.decode()
    assert bad == 0, f"{bad} of {len(cases)} cases differ, e.g. {cases[np.flatnonzero(got.view(np.uint64) != want.view(np.uint64))[:3]]}"
    assert np.isfinite(want).all()
