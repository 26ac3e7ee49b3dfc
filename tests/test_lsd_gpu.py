"""GPU parity: the HIP line extractor through the C ABI against oracle/lsd_oracle.cpp (both tie orders; 0 = the real std::sort).
Rows a10, a11, a12 of SURVEY.md §8.  Integer stages, the visiting order, keyline fields and LBD bytes must be
identical; segment endpoints are float32 roundings of FP64 results whose only cross-platform difference is the
last-bit behaviour of libm (cos/sin/log/exp/pow), so they are compared exactly as well and any mismatch is reported."""
import numpy as np
import pytest

import oracle_lib as O
from planarslam_amd import synth

pytestmark = pytest.mark.gpu
DEG = np.pi / 180


@pytest.fixture(scope="module")
def ctx():
    from planarslam_amd._lib import Context
    return Context(0)


def test_device_std_sort_matches_libstdcxx():
    """lsd_keylines' device emulation of libstdc++ std::sort(greater-by-response) on raw keys, through the hook only the TEST build of the library exports
    (libplanar_hip_paranoid.so, `make paranoid`: -DPLANAR_TEST_HOOKS); run in a child process so that this process keeps the product library."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_path = os.path.join(root, "planarslam_amd", "libplanar_hip_paranoid.so")
    if not os.path.exists(lib_path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "planarslam_amd", "csrc"), "paranoid"])
    rng = np.random.default_rng(5)
    cases = [rng.random(n).astype(np.float32) for n in (0, 1, 2, 16, 17, 41, 172, 600, 2048)]
    cases += [rng.integers(0, 4, 500).astype(np.float32), rng.integers(0, 30, 2048).astype(np.float32), np.zeros(300, np.float32),
              np.arange(700, dtype=np.float32), np.arange(700, dtype=np.float32)[::-1].copy(),
              np.concatenate([np.arange(350), np.arange(350)[::-1]]).astype(np.float32)]
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), *cases)
        code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
                "from planarslam_amd._lib import Context, check, lib\n"
                "ctx = Context(0); z = np.load(%r); out = {}\n"
                "for name in z.files:\n"
                "    k = z[name].copy(); perm = np.zeros(len(k), np.int32)\n"
                "    check(lib().planar_debug_std_sort_desc(ctx.h, k.ctypes.data, perm.ctypes.data, len(k)))\n"
                "    out['k_' + name] = k; out['p_' + name] = perm\n"
                "np.savez(%r, **out)\n") % (root, os.path.join(td, "in.npz"), os.path.join(td, "out.npz"))
        subprocess.check_call([sys.executable, "-W", "ignore", "-c", code], env=dict(os.environ, PLANAR_HIP_LIB=lib_path))
        z = np.load(os.path.join(td, "out.npz"))
        for i, keys in enumerate(cases):
            rk, rp = O.std_sort_desc(keys)
            np.testing.assert_array_equal(z[f"k_arr_{i}"], rk)
            np.testing.assert_array_equal(z[f"p_arr_{i}"], rp)


@pytest.mark.parametrize("tie", [0, 1])
@pytest.mark.parametrize("seed", [1234, 77])
def test_lsd_stages_and_segments(ctx, seed, tie):
    """tie = 0: pixels of one gradient bin in the order the REAL libstdc++ std::sort leaves them (oracle runs std::sort); 1: raster order."""
    from planarslam_amd.lines import LineSegment
    img = synth.gray_image(seed)
    ref = O.lsd_detect(img, tie_order=tie, want_stages=True)
    ls = LineSegment(640, 480, 1, ctx, tie_order=tie)
    ls.ExtractLineSegment(img)
    ang = ls.read_stage(0, 0)
    ref_deg_defined = ref["angles"] != -1024.0
    np.testing.assert_array_equal(ang != -1024.0, ref_deg_defined)
    np.testing.assert_array_equal((ang.astype(np.float64) * DEG)[ref_deg_defined], ref["angles"][ref_deg_defined])
    order = ls.read_stage(0, 2)
    ref_order = ref["order"][ref["order"] >= 0]
    ref_order = ref_order[ref_deg_defined.ravel()[ref_order]]          # the device list drops undefined pixels up front
    np.testing.assert_array_equal(order, ref_order)
    segs = ls.read_stage(0, 3)
    assert len(segs) == len(ref["xy"]) > 100
    got = np.stack([segs["x1"], segs["y1"], segs["x2"], segs["y2"]], 1)
    np.testing.assert_array_equal(got, ref["xy"])
    np.testing.assert_allclose(segs["nfa"], ref["wpn"][:, 2], rtol=1e-9)
    np.testing.assert_array_equal(segs["p"], ref["wpn"][:, 1])
    np.testing.assert_allclose(segs["width"], ref["wpn"][:, 0], rtol=1e-12)


def test_extract_line_segment_batch(ctx):
    from planarslam_amd.lines import LineSegment
    imgs = np.stack([synth.gray_image(1234), synth.gray_image(5), np.full((480, 640), 90, np.uint8), synth.gray_image(9)])
    imgs[1, 100:300, 150:450] = 220
    ls = LineSegment(640, 480, 4, ctx, tie_order=1)
    kl, desc, eq, n = ls.ExtractLineSegment(imgs)
    for b in range(4):
        rk, rd, re, _, nd = O.extract_line_segment(imgs[b], tie_order=1)
        assert n[b] == len(rk)
        for f in rk.dtype.names:
            np.testing.assert_array_equal(kl[b, :n[b]][f], rk[f], err_msg=f"frame {b} field {f}")
        np.testing.assert_array_equal(desc[b, :n[b]], rd)
        np.testing.assert_array_equal(eq[b, :n[b]], re)
    assert n[2] == 0 and n[0] == 40


def test_few_lines_keep_detection_order(ctx):
    from planarslam_amd.lines import LineSegment
    img = np.full((480, 640), 40, np.uint8)
    img[100:300, 150:450] = 200
    ls = LineSegment(640, 480, 1, ctx)
    kl, desc, eq, n = ls.ExtractLineSegment(img)
    rk, rd, re, _, nd = O.extract_line_segment(img, tie_order=0)
    assert n[0] == len(rk) == nd and 1 <= nd <= 40
    np.testing.assert_array_equal(kl[0, :nd]["class_id"], np.arange(nd))
    np.testing.assert_array_equal(desc[0, :nd], rd)


def test_noisy_image_uses_global_used_tail(ctx):
    """More than 32768 defined pixels: the `used` flags beyond the LDS-resident head live in global memory."""
    from planarslam_amd.lines import LineSegment
    rng = np.random.default_rng(17)
    img = synth.gray_image(3).astype(np.int32) + rng.integers(-40, 41, (480, 640))
    img = np.clip(img, 0, 255).astype(np.uint8)
    ref = O.lsd_detect(img, tie_order=0, want_stages=True)
    assert (ref["angles"] != -1024.0).sum() > 40000
    ls = LineSegment(640, 480, 1, ctx)
    kl, desc, eq, n = ls.ExtractLineSegment(img)
    segs = ls.read_stage(0, 3)
    got = np.stack([segs["x1"], segs["y1"], segs["x2"], segs["y2"]], 1)
    np.testing.assert_array_equal(got, ref["xy"])
    rk, rd, re, _, nd = O.extract_line_segment(img, tie_order=0)
    assert n[0] == len(rk)
    np.testing.assert_array_equal(desc[0, :n[0]], rd)


def test_many_segments_sort_in_global_scratch(ctx):
    """> 1024 raw segments: the response sort leaves LDS; many equal-length edges exercise std::sort's tie order."""
    from planarslam_amd.lines import LineSegment
    rng = np.random.default_rng(23)
    img = np.full((480, 640), 60, np.uint8)
    for gy in range(0, 480, 24):
        for gx in range(0, 640, 24):
            if rng.random() < 0.9:
                w, h = rng.integers(12, 20, 2)
                img[gy + 2:gy + 2 + h, gx + 2:gx + 2 + w] = rng.integers(120, 255)
    ls = LineSegment(640, 480, 1, ctx)
    kl, desc, eq, n = ls.ExtractLineSegment(img)
    segs = ls.read_stage(0, 3)
    assert len(segs) > 1024
    rk, rd, re, _, nd = O.extract_line_segment(img, tie_order=0)
    assert nd == len(segs) and n[0] == 40
    for f in rk.dtype.names:
        np.testing.assert_array_equal(kl[0][f], rk[f], err_msg=f)
    np.testing.assert_array_equal(desc[0], rd)


@pytest.mark.parametrize("size", [(752, 480), (320, 240), (500, 375), (1280, 720)])
def test_other_image_sizes(ctx, size):
    """(1280 x 720, round 6: 589 000 gradient pixels at the 0.8x working scale do not fit the sort's LDS stop bitmaps: the top levels of the std::sort arrangement go through
    isort::wg_partition_long - rank prefixes in LDS, ballots recomputed, swap partners through a scratch array)"""
    from planarslam_amd.lines import LineSegment
    W, H = size
    img = synth.gray_image(31, w=W, h=H)
    ls = LineSegment(W, H, 2, ctx)
    kl, desc, eq, n = ls.ExtractLineSegment(np.stack([img, img[::-1].copy()]))
    for b, im in enumerate((img, img[::-1].copy())):
        rk, rd, re, _, nd = O.extract_line_segment(im, tie_order=0)
        assert n[b] == len(rk) > 5
        assert kl[b, :n[b]].tobytes() == rk.tobytes()
        np.testing.assert_array_equal(desc[b, :n[b]], rd)
        np.testing.assert_array_equal(eq[b, :n[b]], re)


def test_top_only_mode_gives_the_same_key_lines(ctx):
    """planar_lsd_set_top_only: the NFA stage for the longest regions first, the rest only where that cannot settle the frame.  Key lines, descriptors and equations are the default
    mode's bit for bit - on ordinary images (settled by the longest regions), on an image with fewer than 41 segments (all regions from the start or after the check), on a grid of
    equal-length edges (equal responses among the kept lines: redone with all regions) and on an empty image."""
    from planarslam_amd.lines import LineSegment
    rng = np.random.default_rng(23)
    grid = np.full((480, 640), 60, np.uint8)
    for gy in range(0, 480, 24):
        for gx in range(0, 640, 24):
            if rng.random() < 0.9:
                w, h = rng.integers(12, 20, 2)
                grid[gy + 2:gy + 2 + h, gx + 2:gx + 2 + w] = rng.integers(120, 255)
    few = np.full((480, 640), 40, np.uint8); few[100:300, 150:450] = 200
    noisy = np.clip(synth.gray_image(3).astype(np.int32) + rng.integers(-40, 41, (480, 640)), 0, 255).astype(np.uint8)
    bars = np.full((480, 640), 50, np.uint8)                          # noise-free bars: the two long edges of a bar are mirror images - equal responses among the kept lines
    for i in range(26):
        y0, x0 = 8 + 18 * i, 20 + 3 * (i % 5)
        bars[y0:y0 + 9, x0:x0 + 120 + 17 * i] = 210
    imgs = np.stack([synth.gray_image(1234), synth.gray_image(5), grid, few, np.full((480, 640), 90, np.uint8), noisy, synth.gray_image(77), synth.gray_image(9), bars])
    B = len(imgs)
    full = LineSegment(640, 480, B, ctx)
    top = LineSegment(640, 480, B, ctx, top_only=True)
    kf, df, ef, nf = full.ExtractLineSegment(imgs)
    kt, dt, et, nt = top.ExtractLineSegment(imgs)
    np.testing.assert_array_equal(nt, nf)
    stats = np.stack([top.read_stage(b, 6) for b in range(B)])
    for b in range(B):
        n = int(nf[b])
        assert kt[b, :n].tobytes() == kf[b, :n].tobytes(), (b, stats[b])
        np.testing.assert_array_equal(dt[b, :n], df[b, :n]); np.testing.assert_array_equal(et[b, :n], ef[b, :n])
    print("top-only [settled, redone, all regions, regions]:", stats.tolist())
    redo = LineSegment(640, 480, B, ctx, top_only=2)                           # the path taken when kept lines have equal responses, forced for every settled frame
    kr, dr, er, nr = redo.ExtractLineSegment(imgs)
    np.testing.assert_array_equal(nr, nf)
    stats_r = np.stack([redo.read_stage(b, 6) for b in range(B)])
    assert (stats_r[:, 1] == stats_r[:, 0]).all() and stats_r[:, 1].sum() >= 5
    for b in range(B):
        n = int(nf[b])
        assert kr[b, :n].tobytes() == kf[b, :n].tobytes() and np.array_equal(dr[b, :n], df[b, :n]) and np.array_equal(er[b, :n], ef[b, :n]), b
    assert len(redo.read_stage(0, 3)) == len(full.read_stage(0, 3))            # after the redo every region has been through the NFA stage
    assert stats[0, 0] == 1 and stats[1, 0] == 1 and stats[6, 0] == 1            # ordinary images: the longest regions settle the frame ...
    assert (stats[[0, 1, 6], 1] == 0).all()                                    # ... without equal responses among the kept lines
    assert stats[3, 0] == 0 and nf[3] <= 40 and nf[4] == 0                      # too few segments: every region
    assert len(top.read_stage(0, 3)) < len(full.read_stage(0, 3))               # and fewer regions went through the NFA stage
