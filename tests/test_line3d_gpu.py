"""3-D line back-projection (Frame::isLineGood + extract3dline_mahdist, src/Frame.cc:189-267, src/LineExtractor.cpp:1157-1470) on the GPU vs the oracle
restatement (oracle/line3d_oracle.cpp; cv::SVD restated, parity unpinned below it): which lines get a 3-D segment, their inlier counts, end points and
mvDepthLine identical; directions within 1e-9 (the Jacobi rotations go through hypot(), whose last bit differs between libm and the device library)."""
import numpy as np
import pytest

import oracle_lib as O
from planarslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from planarslam_amd._lib import Context
    return Context(0)


def _frames(seeds):
    from planarslam_amd._lib import KEYLINE_DTYPE
    B = len(seeds)
    kl = np.zeros((B, 40), KEYLINE_DTYPE); n = np.zeros(B, np.int32); depth = np.zeros((B, 480, 640), np.uint16)
    for b, s in enumerate(seeds):
        k = O.extract_line_segment(synth.gray_image(s), tie_order=0)[0]
        kl[b, :len(k)] = k; n[b] = len(k)
        depth[b] = synth.depth_image(100 + s, noise=(b % 2 == 0), holes=(b % 3 != 2))
    return kl, n, depth


def test_is_line_good_batch_equals_oracle(ctx):
    from planarslam_amd.lines import is_line_good
    kl, n, depth = _frames([11, 12, 13, 14, 15, 16])
    depth[5, :, 300:] = 0                                   # half a frame without depth: lines lose samples, some drop below 10
    n[4] = 7
    seeds = np.array([7, 0, 123456, 4000000000, 99, 5], np.uint32)
    got = is_line_good(kl, n, depth, seeds, ctx=ctx)
    total_good = 0
    for b in range(len(n)):
        want = O.is_line_good(kl[b, :n[b]], depth[b], int(seeds[b]))
        k = int(n[b])
        np.testing.assert_array_equal(got["good"][b, :k], want["good"], err_msg=f"frame {b}")
        np.testing.assert_array_equal(got["n_inliers"][b, :k], want["n_inliers"])
        np.testing.assert_array_equal(got["depth_line"][b, :k], want["depth_line"])
        np.testing.assert_array_equal(got["lines3d"][b, :k], want["lines3d"])          # end points are input points: exact once the inlier sets agree
        g = want["good"] > 0
        assert np.abs(got["direction"][b, :k][g] - want["direction"][g]).max(initial=0) <= 1e-9
        assert got["n_good"][b] == g.sum()
        np.testing.assert_array_equal(got["packed_dirs"][b, :g.sum()], got["direction"][b, :k][g])
        assert (got["good"][b, k:] == 0).all() and (got["depth_line"][b, k:] == -1).all()
        total_good += int(g.sum())
    assert total_good > 100 and (O.is_line_good(kl[5, :n[5]], depth[5], 5)["n_samples"] < 10).any()


def test_is_line_good_depends_on_the_seed_like_rand(ctx):
    """Different seeds draw different point pairs; the emulated generator is glibc's (checked against libc on the CPU side)."""
    from planarslam_amd.lines import is_line_good
    kl, n, depth = _frames([21])
    depth[0] = synth.depth_image(300, noise=True)
    a = is_line_good(kl, n, depth, np.array([1], np.uint32), ctx=ctx)
    for s in (2, 77):
        want = O.is_line_good(kl[0, :n[0]], depth[0], s)
        got = is_line_good(kl, n, depth, np.array([s], np.uint32), ctx=ctx)
        np.testing.assert_array_equal(got["n_inliers"][0, :n[0]], want["n_inliers"])
        np.testing.assert_array_equal(got["lines3d"][0, :n[0]], want["lines3d"])
    assert a["n_good"][0] > 10


def test_directions_feed_the_manhattan_tracker(ctx):
    from planarslam_amd.lines import is_line_good
    from planarslam_amd.manhattan import Tracking
    from planarslam_amd.planes import SurfaceNormals
    kl, n, depth = _frames([31])
    r = is_line_good(kl, n, depth, np.array([3], np.uint32), ctx=ctx)
    nrm, _ = SurfaceNormals(640, 480, 1, ctx).compute(depth[0])
    R0 = np.eye(3, dtype=np.float32)[None]
    got = Tracking(ctx).TrackManhattanFrame(R0, nrm[None], np.array([len(nrm)], np.int32), r["packed_dirs"], r["n_good"])
    want = O.track_manhattan_frame(R0[0], O.surface_normals(depth[0])[0], r["packed_dirs"][0, :r["n_good"][0]])
    assert np.abs(got["R"][0] - want["R"]).max() <= 1e-6


import os  # noqa: E402

import line3d_cases as cases  # noqa: E402


@pytest.mark.parametrize("name", list(cases.CASES))
def test_is_line_good_hip_equals_real_reference_fixture(ctx, golden_dir, name):
    """HIP vs what the reference's own code returned (tests/golden/line3d_ref.npz): which lines are good, end points, mvDepthLine, inlier counts identical;
    directions within 1e-9."""
    from planarslam_amd._lib import KEYLINE_DTYPE
    from planarslam_amd.lines import is_line_good
    g = np.load(os.path.join(golden_dir, "line3d_ref.npz"))
    kl, d, seed = cases.build(name)
    n = len(kl)
    klb = np.zeros((1, 40), KEYLINE_DTYPE); klb[0, :n] = kl
    r = is_line_good(klb, np.array([n], np.int32), d[None], np.array([seed], np.uint32), ctx=ctx)
    good = g[f"{name}/good"] > 0
    np.testing.assert_array_equal(r["good"][0, :n], g[f"{name}/good"])
    np.testing.assert_array_equal(r["depth_line"][0, :n], g[f"{name}/depth_line"])
    np.testing.assert_array_equal(r["lines3d"][0, :n], g[f"{name}/lines3d"])
    np.testing.assert_array_equal(r["n_inliers"][0, :n][good], g[f"{name}/n_inliers"][good])
    assert np.abs(r["direction"][0, :n] - g[f"{name}/direction"]).max() <= 1e-9
