"""GPU parity (bit-exact): MapPoint::ComputeDistinctiveDescriptors through planar_distinctive_descriptors against the oracle and the committed outputs of the
reference's own function (tests/golden/distinctive_ref.npz, oracle/_ref/ref_frame)."""
import os

import numpy as np
import pytest

import distinctive_cases as dc
import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distinctive_ref.npz"))


def test_matches_reference_fixture_and_oracle():
    from planarslam_amd import distinctive_descriptors
    cs = dc.cases()
    n = GOLD["n"]; off = np.r_[0, np.cumsum(n)]
    keep = [d[GOLD["bad"][off[i]:off[i + 1]] == 0] for i, d in enumerate(cs)]      # the reference skips observations in bad key frames
    best, med = distinctive_descriptors(keep)
    for i, k in enumerate(keep):
        idx, m = ol.distinctive_descriptor(k)
        assert best[i] == idx and (idx < 0 or med[i] == m), i
        if idx >= 0:
            assert np.array_equal(k[best[i]], GOLD["chosen"][i]), i
        else:
            assert not GOLD["chosen"][i].any()


def test_map_line_version_matches_its_reference_fixture():
    """MapLine::ComputeDistinctiveDescriptors (src/MapLine.cpp:241-312) is the same selection on the key frames' LBD rows: planar_distinctive_descriptors against the
    outputs of the reference's own function (tests/golden/distinctive_lines_ref.npz)."""
    from planarslam_amd import distinctive_descriptors
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distinctive_lines_ref.npz"))
    cs = dc.cases(seed=17, n_points=40)
    off = np.r_[0, np.cumsum(G["n"])]
    keep = [d[G["bad"][off[i]:off[i + 1]] == 0] for i, d in enumerate(cs)]
    best, _ = distinctive_descriptors(keep)
    for i, k in enumerate(keep):
        assert (not G["chosen"][i].any()) if best[i] < 0 else np.array_equal(k[best[i]], G["chosen"][i]), i


def test_many_observations_and_large_batch():
    from planarslam_amd import distinctive_descriptors
    rng = np.random.default_rng(5)
    obs = [rng.integers(0, 256, size=(int(n), 32), dtype=np.uint8) for n in [1, 2, 63, 64, 65, 200, 700] + list(rng.integers(1, 40, size=3000))]
    best, med = distinctive_descriptors(obs)
    for i in list(range(7)) + list(range(7, len(obs), 97)):
        idx, m = ol.distinctive_descriptor(obs[i])
        assert best[i] == idx and med[i] == m, i
