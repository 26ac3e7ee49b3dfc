"""GPU parity (bit-exact, integer work): HIP Hamming matchers vs the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _descs(rng, n, base=None, flips=20):
    """Random 256-bit descriptors; with `base`, noisy copies of base rows (realistic matches + ties)."""
    if base is None:
        return rng.integers(0, 256, (n, 32)).astype(np.uint8)
    src = base[rng.integers(0, len(base), n)].copy()
    bits = np.unpackbits(src, axis=1)
    for i in range(n):
        bits[i, rng.integers(0, 256, rng.integers(0, flips))] ^= 1
    return np.packbits(bits, axis=1)


@pytest.mark.parametrize("k", [1, 2])
def test_hamming_knn_matches_bfmatcher_semantics(k):
    from planarslam_amd import hamming_knn
    rng = np.random.default_rng(0)
    B, qs, ts = 5, 1100, 1030
    nq = np.array([1000, 1100, 1, 257, 513], np.int32)
    nt = np.array([1000, 1030, 700, 2, 256], np.int32)
    t = np.stack([_descs(rng, ts) for _ in range(B)])
    q = np.stack([_descs(rng, qs, t[b][:nt[b]], 40) for b in range(B)])
    t[0, 10] = t[0, 3]            # exact duplicates: ties must resolve to the lowest train index
    q[0, 0] = t[0, 3]
    idx, dist = hamming_knn(q, nq, t, nt, k)
    for b in range(B):
        oi, od = ol.bf_knn(q[b][:nq[b]], t[b][:nt[b]], k)
        assert np.array_equal(idx[b, :nq[b]], oi) and np.array_equal(dist[b, :nq[b]], od)
    assert idx[0, 0, 0] == 3 and dist[0, 0, 0] == 0


def test_match_orb_points():
    from planarslam_amd import ORBmatcher
    rng = np.random.default_rng(1)
    B, cs, ls = 4, 1024, 1010
    n_cur = np.array([1003, 1024, 400, 30], np.int32)
    n_last = np.array([1001, 1010, 380, 25], np.int32)
    last = np.stack([_descs(rng, ls) for _ in range(B)])
    cur = np.stack([_descs(rng, cs, last[b][:n_last[b]], 12 + 10 * b) for b in range(B)])
    has = (rng.random((B, ls)) < 0.7).astype(np.uint8)
    outl = (rng.random((B, ls)) < 0.2).astype(np.uint8)
    init = np.full((B, cs), -1, np.int32); init[:, ::7] = 12345     # pre-existing assignments must survive
    m, npair = ORBmatcher().MatchORBPoints(cur, n_cur, last, n_last, has, outl, init)
    for b in range(B):
        om, on = ol.match_orb_points(cur[b][:n_cur[b]], last[b][:n_last[b]], has[b], outl[b], init[b][:n_cur[b]])
        assert npair[b] == on
        assert np.array_equal(m[b, :n_cur[b]], om)
        assert np.array_equal(m[b, n_cur[b]:], init[b, n_cur[b]:])


def test_lsd_search_by_descriptor():
    from planarslam_amd import LSDmatcher
    rng = np.random.default_rng(2)
    B, ks, cs = 6, 40, 48
    n_kf = np.array([40, 40, 13, 1, 40, 0], np.int32)
    n_cur = np.array([40, 48, 40, 40, 1, 40], np.int32)        # n_cur < 2: no matches (reference would read OOB)
    cur = np.stack([_descs(rng, cs) for _ in range(B)])
    kf = np.stack([_descs(rng, ks, cur[b][:max(n_cur[b], 1)], 30) for b in range(B)])
    has = (rng.random((B, ks)) < 0.8).astype(np.uint8)
    m, nm = LSDmatcher().SearchByDescriptor(kf, n_kf, cur, n_cur, has)
    for b in range(B):
        om, on = ol.lsd_search_by_descriptor(kf[b][:n_kf[b]], cur[b][:n_cur[b]], has[b])
        assert nm[b] == on and np.array_equal(m[b, :n_cur[b]], om)
