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
