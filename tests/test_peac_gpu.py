"""GPU parity (bit-exact): HIP PEAC plane segmentation vs (a) committed outputs of the REAL reference plane extractor
and (b) the CPU oracle on seeded synthetic depth."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from planarslam_amd.synth import depth_image

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "peac_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[5:-4])
def test_hip_matches_reference_golden(path):
    from planarslam_amd import PlaneDetection
    z = np.load(path)
    planes, labels = PlaneDetection(640, 480).run(z["depth"])
    assert np.array_equal(labels, z["labels"].astype(np.int32))
    assert planes.shape == z["planes"].shape and np.array_equal(planes, z["planes"])


def test_hip_batch_matches_oracle():
    from planarslam_amd import PlaneDetection
    depths = np.stack([depth_image(50 + i, noise=(i % 2 == 0), holes=(i % 3 != 0)) for i in range(6)])
    res = PlaneDetection(640, 480, max_batch=8).run(depths)
    for b in range(6):
        op, olab = ol.peac_run(depths[b])
        assert np.array_equal(res[b][1], olab), f"labels frame {b}"
        assert res[b][0].shape == op.shape and np.array_equal(res[b][0], op), f"planes frame {b}"


def test_hip_empty_and_single_plane():
    from planarslam_amd import PlaneDetection
    pd = PlaneDetection(640, 480, max_batch=2)
    d = np.zeros((2, 480, 640), np.uint16)
    d[1] = 10000
    (p0, l0), (p1, l1) = pd.run(d)
    assert len(p0) == 0 and (l0 == -1).all()
    assert len(p1) == 1 and p1[0][0] == 307200 and (l1 == 0).all()


def test_hip_other_size():
    from planarslam_amd import PlaneDetection
    d = depth_image(77, 320, 240)
    op, olab = ol.peac_run(d)
    planes, labels = PlaneDetection(320, 240).run(d)
    assert np.array_equal(labels, olab) and np.array_equal(planes, op)


@pytest.mark.parametrize("w,h,B", [(848, 480, 3), (1280, 720, 2), (325, 247, 2)])
def test_hip_frames_beyond_640x480(w, h, B):
    """PlaneDetection takes any frame size (reference: src/PlaneExtractor.cpp:59, include/peac/AHCPlaneFitter.hpp:225 - blocks of 10x10; the pixels outside the block grid are only reached by the flood fill).  848x480 = 4 032 blocks (RealSense); 1280x720 = 9 216 blocks: the clustering wavefront's queue and merge-parent table take 114 KB of LDS,
    the tournament's columns hold five ids per lane; 325x247: neither side a multiple of the block size.  Labels and plane doubles bit-exact against the oracle."""
    from planarslam_amd import PlaneDetection
    depths = np.stack([depth_image(300 + 7 * i + w, w, h, noise=(i % 2 == 0), holes=True) for i in range(B)])
    res = PlaneDetection(w, h, max_batch=B).run(depths)
    for b in range(B):
        op, olab = ol.peac_run(depths[b])
        assert len(op) >= 1
        assert np.array_equal(res[b][1], olab), f"labels frame {b}"
        assert res[b][0].shape == op.shape and np.array_equal(res[b][0], op), f"planes frame {b}"


def test_hip_more_frames_than_cus_and_repeated_calls():
    """Workgroups take their frame from a start-order counter: more frames than CUs (several dispatch rounds) and a second call on the
    same handle (counter reset) must give, frame by frame, what a single-frame call gives."""
    from planarslam_amd import PlaneDetection
    src = np.stack([depth_image(900 + i) for i in range(3)])
    one = PlaneDetection(640, 480, max_batch=1)
    want = [one.run(src[i]) for i in range(3)]
    B = 300
    pd = PlaneDetection(640, 480, max_batch=B)
    depths = src[np.arange(B) % 3]
    for _ in range(2):
        res = pd.run(depths)
        for b in range(B):
            wp, wl = want[b % 3]
            assert np.array_equal(res[b][1], wl), f"labels frame {b}"
            assert res[b][0].shape == wp.shape and np.array_equal(res[b][0], wp), f"planes frame {b}"


def test_refinement_rare_paths_give_the_same_labels():
    """peac_refine folds a pixel's pairs through a fast path (up to four pairs of one pixel in one flood-fill step, sorted in registers) and keeps two rare ones: the generic
    fold (more than four pairs) and the `extra_connects` replay (the pixel changed hands twice within a step).  The test build `-DPLANAR_REFINE_PARANOID`
    (make -C planarslam_amd/csrc paranoid) sends EVERY pixel through the generic fold and replays every pixel that changed hands at all - both must be no-ops on the
    result: labels and planes of that build equal the product's (which equal the oracle's) on golden, synthetic and SE3-rendered frames."""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "planarslam_amd", "libplanar_hip_paranoid.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "planarslam_amd", "csrc"), "paranoid"])
    import torch
    from planarslam_amd import PlaneDetection, synth_se3
    from planarslam_amd.synth import TUM3, gray_image
    tex = torch.from_numpy(np.stack([gray_image(1234 + i, 736, 576) for i in range(4)])).cuda()
    _, loop_d, _ = synth_se3.render_streams(torch, tex, 8, 2, TUM3, seed=5)
    depths = [np.load(p)["depth"] for p in GOLD] + [depth_image(50 + i, noise=(i % 2 == 0), holes=(i % 3 != 0)) for i in range(4)]
    depths += [loop_d[b, 1].cpu().numpy().view(np.uint16) for b in range(8)]
    depths = np.stack(depths)
    res = PlaneDetection(640, 480, max_batch=len(depths)).run(depths)
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "d.npy"), depths)
        code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
                "from planarslam_amd import PlaneDetection\n"
                "d = np.load(%r)\n"
                "r = PlaneDetection(640, 480, max_batch=len(d)).run(d)\n"
                "np.savez(%r, labels=np.stack([x[1] for x in r]), n=np.array([len(x[0]) for x in r]), planes=np.concatenate([x[0] for x in r]))\n") % (root, os.path.join(td, "d.npy"), os.path.join(td, "o.npz"))
        subprocess.check_call([sys.executable, "-W", "ignore", "-c", code], env=dict(os.environ, PLANAR_HIP_LIB=lib))
        z = np.load(os.path.join(td, "o.npz"))
    assert np.array_equal(z["n"], [len(x[0]) for x in res])
    assert np.array_equal(z["labels"], np.stack([x[1] for x in res]))
    assert np.array_equal(z["planes"], np.concatenate([x[0] for x in res]))
    # frames whose sides are not multiples of the 10-pixel block (pixels outside the block grid take part in the fill): the two builds agree there too
    odd = np.stack([depth_image(900 + i, 325, 247, noise=(i % 2 == 0), holes=True) for i in range(3)])
    res_odd = PlaneDetection(325, 247, max_batch=3).run(odd)
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "d.npy"), odd)
        code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
                "from planarslam_amd import PlaneDetection\n"
                "d = np.load(%r)\n"
                "r = PlaneDetection(325, 247, max_batch=len(d)).run(d)\n"
                "np.savez(%r, labels=np.stack([x[1] for x in r]))\n") % (root, os.path.join(td, "d.npy"), os.path.join(td, "o.npz"))
        subprocess.check_call([sys.executable, "-W", "ignore", "-c", code], env=dict(os.environ, PLANAR_HIP_LIB=lib))
        assert np.array_equal(np.load(os.path.join(td, "o.npz"))["labels"], np.stack([x[1] for x in res_odd]))
    for b in (len(GOLD) + 4, len(depths) - 1):                       # (and the product's are the oracle's on the SE3 frames too)
        op, olab = ol.peac_run(depths[b])
        assert np.array_equal(res[b][1], olab) and np.array_equal(res[b][0], op)
