"""GPU parity tests (run with -m gpu on the MI355X box): the HIP ORB path through the C ABI against
(a) the CPU oracle on seeded inputs, stage by stage, and (b) the committed outputs of the REAL
reference ORBextractor (tests/golden).  Bit-exact: integer/byte work, and float fields that are
produced by the same IEEE operation sequence."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from planarslam_amd.synth import gray_image

pytestmark = pytest.mark.gpu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_*.npz")))


def _mk(W, H, B=1, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
    from planarslam_amd import ORBextractor
    return ORBextractor(nfeatures, scale, nlevels, ini, mn, width=W, height=H, max_batch=B)


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[4:-4])
def test_hip_matches_reference_golden(path):
    z = np.load(path)
    p = z["params"]
    img = z["image"]
    ex = _mk(img.shape[1], img.shape[0], 1, int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4]))
    kps, desc = ex(img)
    assert len(kps) == len(z["kps"])
    assert kps.tobytes() == z["kps"].tobytes()
    assert np.array_equal(desc, z["desc"])


@pytest.mark.parametrize("seed,W,H", [(1234, 640, 480), (77, 640, 480), (5, 400, 304), (9, 333, 257)])
def test_hip_stages_match_oracle(seed, W, H):
    img = gray_image(seed, W, H)
    if seed == 77:   # heavy texture: tens of thousands of FAST candidates, deep octree
        rng = np.random.default_rng(seed)
        img = np.clip(img.astype(np.int32) + rng.integers(-40, 41, img.shape), 0, 255).astype(np.uint8)
    o = ol.OrbOracle()
    okps, odesc = o.extract(img)
    ex = _mk(W, H)
    kps, desc = ex(img)
    for l in range(8):
        assert np.array_equal(ex.read_level(0, l), o.level(l)), f"pyramid level {l}"
    for l in range(8):
        assert np.array_equal(ex.read_candidates(0, l), o.candidates(l)), f"FAST candidates level {l}"
    for l in range(8):
        ob = o.blurred(l)
        if ob is not None:
            assert np.array_equal(ex.read_level(0, l, blurred=True), ob), f"blur level {l}"
    assert len(kps) == len(okps)
    for f in ("x", "y", "octave", "response", "size"):
        assert np.array_equal(kps[f], okps[f]), f
    assert np.array_equal(kps["angle"], okps["angle"])
    assert np.array_equal(desc, odesc)


def test_hip_batch_equals_single_and_is_deterministic():
    imgs = np.stack([gray_image(100 + i) for i in range(5)])
    ex = _mk(640, 480, B=8)
    res1 = ex(imgs)
    res2 = ex(imgs[::-1].copy())[::-1]
    o = ol.OrbOracle()
    for b in range(5):
        okps, odesc = o.extract(imgs[b])
        for res in (res1, res2):
            assert res[b][0].tobytes() == okps.tobytes()
            assert np.array_equal(res[b][1], odesc)


def test_hip_uniform_noise_worst_case_candidates():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    o = ol.OrbOracle()
    okps, odesc = o.extract(img)
    kps, desc = _mk(640, 480)(img)
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)


def test_hip_flat_image_has_no_keypoints():
    img = np.full((480, 640), 128, np.uint8)
    kps, desc = _mk(640, 480)(img)
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_errors_are_reported_not_raised_from_c():
    from planarslam_amd import PlanarError
    with pytest.raises(PlanarError):
        _mk(32, 32)                       # below the supported size
    ex = _mk(640, 480)
    with pytest.raises(ValueError):
        ex(np.zeros((100, 100), np.uint8))
    with pytest.raises(TypeError):
        ex(np.zeros((480, 640), np.float32))
