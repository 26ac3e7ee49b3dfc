"""CPU tests: the ORB oracle against (a) committed outputs of the REAL reference ORBextractor
(tests/golden/orb_*.npz, made by tools/gen_golden_orb.py from oracle/_ref/ref_orb) and (b) the
live oracle/_ref/ref_orb binary when it is present."""
import ctypes
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from planarslam_amd.synth import gray_image


def _params(z):
    p = z["params"]
    return dict(nfeatures=int(p[0]), scale=float(p[1]), nlevels=int(p[2]), ini=int(p[3]), mn=int(p[4]))


def golden_cases():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(glob.glob(os.path.join(here, "orb_*.npz")))


@pytest.mark.parametrize("path", golden_cases(), ids=lambda p: os.path.basename(p)[4:-4])
def test_oracle_matches_reference_golden(path):
    z = np.load(path)
    o = ol.OrbOracle(**_params(z))
    kps, desc = o.extract(z["image"])
    assert len(kps) == len(z["kps"])
    assert kps.tobytes() == z["kps"].tobytes()          # bit-exact x,y,size,angle,response,octave,class_id
    assert np.array_equal(desc, z["desc"])              # bit-exact 256-bit rBRIEF
    shapes = np.array([o.level(l).shape for l in range(o.nlevels)], np.int32)
    assert np.array_equal(shapes, z["level_shapes"])
    sums = np.array([int(o.level(l).astype(np.int64).sum()) for l in range(o.nlevels)], np.int64)
    assert np.array_equal(sums, z["level_sums"])
    assert np.array_equal(o.level(o.nlevels - 1), z["last_level"])


@pytest.mark.skipif(not os.path.exists(ol.ref_orb_path()), reason="oracle/_ref/ref_orb not built")
@pytest.mark.parametrize("seed,w,h", [(21, 640, 480), (22, 400, 300), (23, 333, 257)])
def test_oracle_matches_live_reference(seed, w, h):
    img = gray_image(seed, w, h)
    rng = np.random.default_rng(seed)
    img = np.clip(img.astype(np.int32) + rng.integers(-25, 26, img.shape), 0, 255).astype(np.uint8)
    o = ol.OrbOracle()
    kps, desc = o.extract(img)
    rk, rd, pyr = ol.run_ref_orb(img)
    assert kps.tobytes() == rk.tobytes()
    assert np.array_equal(desc, rd)
    for l in range(8):
        assert np.array_equal(o.level(l), pyr[l])


def test_constructor_tables():
    o = ol.OrbOracle()
    assert o.features_per_level() == [217, 181, 151, 126, 105, 87, 73, 60]
    umax = [o.L.orc_orb_umax(o.h, v) for v in range(16)]
    assert umax == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_fast_atan2_range_and_quadrants():
    L = ol.lib()
    assert L.orc_fast_atan2(0.0, 1.0) == 0.0
    assert abs(L.orc_fast_atan2(1.0, 0.0) - 90.0) < 1e-4
    assert abs(L.orc_fast_atan2(0.0, -1.0) - 180.0) < 1e-4
    assert abs(L.orc_fast_atan2(-1.0, 0.0) - 270.0) < 1e-4
    rng = np.random.default_rng(0)
    for y, x in rng.normal(size=(200, 2)):
        a = L.orc_fast_atan2(float(y), float(x))
        t = np.degrees(np.arctan2(y, x)) % 360
        assert abs((a - t + 180) % 360 - 180) < 0.02     # documented accuracy ~0.3 deg worst-case is far looser


def test_cv_round_half_even():
    L = ol.lib()
    assert [L.orc_cv_round_d(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_blur_constant_and_impulse():
    L = ol.lib()
    img = np.full((20, 24), 200, np.uint8)
    out = np.zeros_like(img)
    L.orc_gaussian7_s2_u8(img.ctypes.data, 24, 20, 24, out.ctypes.data, 24)
    assert (out == 202).all()    # taps sum to 257/256: 200*257^2/65536 = 201.57 -> 202 (library quirk, see cvprim.cpp)
    img[:] = 0; img[10, 12] = 255
    L.orc_gaussian7_s2_u8(img.ctypes.data, 24, 20, 24, out.ctypes.data, 24)
    taps = np.array([18, 34, 49, 55, 49, 34, 18])
    want = (np.outer(taps, taps) * 255 + 32768) >> 16
    assert np.array_equal(out[7:14, 9:16], want)


def test_resize_identity_and_half():
    L = ol.lib()
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (30, 40)).astype(np.uint8)
    out = np.zeros_like(img)
    L.orc_resize_linear_u8(img.ctypes.data, 40, 30, 40, out.ctypes.data, 40, 30, 40)
    assert np.array_equal(out, img)
    half = np.zeros((15, 20), np.uint8)
    L.orc_resize_linear_u8(img.ctypes.data, 40, 30, 40, half.ctypes.data, 20, 15, 20)
    blk = img.reshape(15, 2, 20, 2).astype(np.int32)
    want = (blk.sum(axis=(1, 3)) + 2) >> 2
    assert np.abs(half.astype(np.int32) - want).max() <= 1


def test_both_gaussian_variants_of_survey_a4_return_the_same_bytes():
    """OpenCV <= 3.4.0 (integer SymmSmall filters, one rounding at the end) and 3.4.1 (ufixedpoint16 per pass, saturating adds) are two code paths.
    With the Q8 taps {18, 34, 49, 55, 49, 34, 18} (sum 257) the horizontal sums never exceed 65535 = 257 * 255, so neither saturation nor a per-pass rounding
    occurs and both reduce to the same integers - checked here on the inputs where they could differ (saturated, alternating, random, tiny images)."""
    L = ol.lib()
    L.orc_gaussian7_variant.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(3)
    imgs = [np.full((40, 56), 255, np.uint8), np.zeros((9, 9), np.uint8), (np.indices((33, 47)).sum(0) % 2 * 255).astype(np.uint8),
            rng.integers(0, 256, (64, 80), dtype=np.uint8), rng.integers(250, 256, (31, 29), dtype=np.uint8), np.full((7, 7), 255, np.uint8)]
    imgs[1][4, 4] = 255
    for img in imgs:
        h, w = img.shape
        a, b = np.zeros_like(img), np.zeros_like(img)
        L.orc_gaussian7_variant(0, img.ctypes.data, w, h, w, a.ctypes.data, w)
        L.orc_gaussian7_variant(1, img.ctypes.data, w, h, w, b.ctypes.data, w)
        np.testing.assert_array_equal(a, b)
    assert a.max() == 255       # 257 / 256 gain: an all-255 image stays at 255 after saturation
