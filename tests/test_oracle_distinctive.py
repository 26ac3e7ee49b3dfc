"""CPU: the oracle's MapPoint::ComputeDistinctiveDescriptors against the reference's own function (committed outputs of oracle/_ref/ref_frame, and the binary
itself when it is present)."""
import os

import numpy as np

import distinctive_cases as dc
import oracle_lib as ol

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distinctive_ref.npz"))
GOLD_LINES = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distinctive_lines_ref.npz"))


def _split_bad(G=None):
    G = GOLD if G is None else G
    n = G["n"]; off = np.r_[0, np.cumsum(n)]
    return [G["bad"][off[i]:off[i + 1]] for i in range(len(n))]


def test_oracle_matches_the_map_line_fixture():
    """MapLine::ComputeDistinctiveDescriptors (src/MapLine.cpp:241-312: cv::norm NORM_HAMMING on LBD rows, the same first-minimum median rule) - outputs of the
    reference's own function compiled into oracle/_ref/ref_frame."""
    cs, bad = dc.cases(seed=17, n_points=40), _split_bad(GOLD_LINES)
    assert len(cs) == len(GOLD_LINES["chosen"])
    for i, (d, b) in enumerate(zip(cs, bad)):
        keep = d[b == 0]
        idx, _ = ol.distinctive_descriptor(keep)
        want = GOLD_LINES["chosen"][i]
        assert (not want.any()) if idx < 0 else np.array_equal(keep[idx], want), i
    if os.path.exists(ol.ref_frame_path()):
        assert np.array_equal(ol.run_ref_distinctive(list(zip(cs, bad)), lines=True), GOLD_LINES["chosen"])


def test_oracle_matches_reference_fixture():
    cs, bad = dc.cases(), _split_bad()
    assert len(cs) == len(GOLD["chosen"])
    ties = 0
    for i, (d, b) in enumerate(zip(cs, bad)):
        keep = d[b == 0]                          # the reference skips observations in bad key frames
        idx, med = ol.distinctive_descriptor(keep)
        want = GOLD["chosen"][i]
        if idx < 0:
            assert not want.any()
            continue
        assert np.array_equal(keep[idx], want), i
        # the choice is the FIRST minimum: brute-force medians agree
        D = np.unpackbits(keep[:, None, :] ^ keep[None, :, :], axis=2).sum(2)
        meds = np.sort(D, 1)[:, int(0.5 * (len(keep) - 1))]
        assert med == meds.min() and idx == int(np.argmin(meds))
        ties += int((meds == meds.min()).sum() > 1)
    assert ties >= 3


def test_reference_binary_when_present():
    if not os.path.exists(ol.ref_frame_path()):
        import pytest
        pytest.skip("oracle/_ref/ref_frame not built (no /root/reference here)")
    cs, bad = dc.cases(), _split_bad()
    out = ol.run_ref_distinctive(list(zip(cs, bad)))
    assert np.array_equal(out, GOLD["chosen"])
