"""The product's std::sort-arrangement engine (planarslam_amd/csrc/isort.h: the very source hipcc compiles for gfx950) compiled with g++ and run on the
host-side wave64 emulator (tests/host_shim/wave_emul.h) against the REAL std::sort of this libstdc++ with a key-only comparator - the call
pcl::VoxelGrid::applyFilter (voxel index) and OpenCV's LSD (1024-bin gradient norm) make.  Word-for-word equality: where std::sort leaves elements of
equal key is exactly what the product has to reproduce (the float summation order of a voxel's points; the order LSD visits its seeds in)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "host_shim")
SO = os.path.join(SHIM, "libisort_host.so")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(SHIM, "isort_host.cpp")
    deps = [src, os.path.join(SHIM, "wave_emul.h"), os.path.join(ROOT, "planarslam_amd", "csrc", "isort.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-o", SO, src])
    E = C.CDLL(SO)
    vp = C.c_void_p
    E.isort_emul.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_char_p, C.c_int]
    E.isort_std_sort.argtypes = [vp, vp, C.c_int, C.c_int]
    E.isort_emul_skip.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_uint, vp, vp, C.c_char_p, C.c_int]
    return E


def _check(E, keys, bounds, shift, config, n_stage=0):
    n = len(keys)
    assert n < (1 << shift)
    arr = ((keys.astype(np.uint32) << np.uint32(shift)) | np.arange(n, dtype=np.uint32)).astype(np.uint32)
    want = arr.copy()
    b = np.asarray(bounds, np.int32)
    E.isort_std_sort(want.ctypes.data, b.ctypes.data, len(b) - 1, shift)
    status = C.c_int(-1); stats = np.zeros(8, np.int64); err = C.create_string_buffer(512)
    rc = E.isort_emul(arr.ctypes.data, b.ctypes.data, len(b) - 1, shift, config, n_stage, C.byref(status), stats.ctypes.data, err, 512)
    assert rc == 0, err.value.decode()
    assert status.value == 0, f"engine status {status.value}"
    bad = np.nonzero(arr != want)[0]
    assert bad.size == 0, f"{bad.size} words differ from std::sort, first at {bad[0]}: engine {arr[bad[0]]:#x} std::sort {want[bad[0]]:#x} (stats {stats})"
    return stats


def _voxel_like(rng, n, run, nkeys):
    """keys as a plane's voxel indices look in raster order: runs of equal keys that drift"""
    base = np.cumsum(rng.integers(-2, 3, size=n // run + 2))
    k = np.repeat(base, run)[:n] + rng.integers(0, 2, size=n) * rng.integers(0, nkeys // 8 + 1)
    return (k - k.min()) % nkeys


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_small_shapes_many_levels(lib, seed):
    """256-thread workgroups, 5 elements per thread: the global tier runs on arrays of a few thousand elements, the LDS tier sees every piece layout"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3000, 9000))
    for keys in (rng.integers(0, 1 << 10, n), rng.integers(0, 7, n), _voxel_like(rng, n, 26, 300), np.arange(n) % 1024, np.zeros(n, np.int64),
                 np.sort(rng.integers(0, 500, n)), np.sort(rng.integers(0, 500, n))[::-1].copy()):
        _check(lib, keys, [0, n], 20 if seed & 1 else 19, 1)


def test_many_ranges_and_tiny_ranges(lib):
    rng = np.random.default_rng(7)
    sizes = [0, 1, 2, 3, 15, 16, 17, 18, 33, 100, 1279, 1280, 1281, 4000, 1, 16, 17, 2500, 0, 40]
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    n = int(bounds[-1])
    _check(lib, rng.integers(0, 64, n), bounds, 19, 1)
    _check(lib, _voxel_like(rng, n, 9, 200), bounds, 19, 1)
    _check(lib, rng.integers(0, 1 << 12, n), bounds, 19, 2)       # 128 threads x 32 elements: full 32-bit chunk masks


@pytest.mark.parametrize("seed", [11, 12])
def test_production_shapes(lib, seed):
    """1024 threads, 23 elements per thread; a wall-sized plane (global tier), mid-sized and small planes packed into blocks"""
    rng = np.random.default_rng(seed)
    sizes = [70000, 23552, 23553, 9000, 300, 12, 5000] if seed == 11 else [120000, 40, 700, 17]
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    n = int(bounds[-1])
    keys = np.concatenate([_voxel_like(rng, s, 26, 2000) for s in sizes])
    st = _check(lib, keys, bounds, 19, 0)
    assert st[1] >= 3


def test_ranges_longer_than_the_stop_bitmaps(lib):
    """Frames beyond ~390 000 sort words (1280x720: LSD's 589 000 gradient pixels, a wall of 900 000 plane points) do not fit the global tier's LDS bitmaps: such a range is
    partitioned by wg_partition_long (rank prefixes only in LDS, ballots recomputed, swap partners through a scratch array).  Configuration 5 gives the bitmaps 16 384 words, so
    the top levels of these arrays take that path and the rest the ordinary one: word for word std::sort's arrangement."""
    rng = np.random.default_rng(31)
    sizes = [150000, 9000, 40000, 16385, 16384, 700]
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    n = int(bounds[-1])
    keys = np.concatenate([_voxel_like(rng, s, 26, 1500) for s in sizes])
    _check(lib, keys, bounds, 19, 5)
    _check(lib, rng.integers(0, 1 << 10, n), [0, n], 20, 5)
    g = np.abs(rng.normal(0, 1, n)) ** 3
    _check(lib, 1023 - np.minimum((g / g.max() * 1023).astype(np.int64), 1023), [0, n], 20, 5)


def test_lsd_like_keys(lib):
    """the 1024-bin gradient norm of an image: most pixels in the lowest bins, a long tail; descending order = ascending 1023 - bin"""
    rng = np.random.default_rng(5)
    n = 511 * 383
    g = np.abs(rng.normal(0, 1, n)) ** 3
    bins = np.minimum((g / g.max() * 1023).astype(np.int64), 1023)
    _check(lib, 1023 - bins, [0, n], 20, 0)


def _plane_voxel_keys(seed, noise):
    """voxel ranks (PCL's ascending voxel index = lexicographic (z, y, x) voxel coordinates) of every detected plane's pixels in raster order"""
    import oracle_lib as ol
    from planarslam_amd.synth import depth_image
    d = depth_image(seed, noise=noise)
    planes, labels = ol.peac_run(d)
    fx, fy, cx, cy = [np.float64(np.float32(v)) for v in (535.4, 539.2, 320.1, 247.6)]
    fac = np.float64(np.float32(1 / 5000.0))
    out = []
    for p in range(len(planes)):
        ys, xs = np.nonzero(labels == p)
        z = d[ys, xs].astype(np.float64) * fac
        pts = np.stack([(xs - cx) * z / fx, (ys - cy) * z / fy, z], 1).astype(np.float32)
        ijk = np.floor(pts * (np.float32(1.0) / np.float32(0.1))).astype(np.int64)
        _, r = np.unique((ijk[:, 2] * 100000 + ijk[:, 1]) * 100000 + ijk[:, 0], return_inverse=True)
        out.append(r)
    return out


def test_heap_sort_fallback_on_real_plane_keys(lib):
    """The 220 417-pixel plane of synthetic frame 60: its saw-tooth voxel keys make 34 median-of-three partitions in a row peel off a few percent each, and
    libstdc++ heap-sorts what is left (424 ranges, the longest 45 161 elements: tools/ trace in DESIGN.md).  The engine must end in the same place."""
    ks = _plane_voxel_keys(60, True)
    sizes = [len(k) for k in ks]
    assert max(sizes) > 200000
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    off = np.concatenate([[0], np.cumsum([k.max() + 1 for k in ks])])[:-1]
    keys = np.concatenate([k + o for k, o in zip(ks, off)])
    st = _check(lib, keys, bounds, 19, 0)
    assert st[7] > 100 and st[6] > 50000, st                      # hundreds of fallback ranges, the longest tens of thousands of elements
    _check(lib, keys, bounds, 19, 3)                             # the same with only 1000 words of every fallback range in LDS (the rest in the array)
    # and with the fallback inside LDS blocks (small capacity -> short ranges reach depth 0 there)
    ks = _plane_voxel_keys(705, True)
    bounds = np.concatenate([[0], np.cumsum([len(k) for k in ks])])
    off = np.concatenate([[0], np.cumsum([k.max() + 1 for k in ks])])[:-1]
    _check(lib, np.concatenate([k + o for k, o in zip(ks, off)]), bounds, 19, 0)


@pytest.mark.parametrize("config", [0, 1])
def test_skip_key_leaves_every_wanted_element_where_std_sort_puts_it(lib, config):
    """LSD's use of the engine: 88 % of a frame's gradient pixels have no defined angle (the lowest bins = the highest keys) and are dropped after the sort.  With a skip key
    the right part of a partition whose pivot is above it is never sorted; every element at or below the key still has to be exactly where std::sort leaves it, the
    rest only has to be the same multiset in the same index ranges."""
    rng = np.random.default_rng(21 + config)
    n = 511 * 383 if config == 0 else 9000
    bins = np.where(rng.random(n) < 0.88, rng.integers(0, 6, n), np.minimum(1023, (-60.0 * np.log(rng.random(n))).astype(np.int64) + 6))
    keys = (1023 - bins).astype(np.uint32)
    skip = int(1023 - 6)                                                  # wanted: bins >= 6
    shift = 20
    arr = ((keys << np.uint32(shift)) | np.arange(n, dtype=np.uint32)).astype(np.uint32)
    want = arr.copy()
    b = np.asarray([0, n], np.int32)
    lib.isort_std_sort(want.ctypes.data, b.ctypes.data, 1, shift)
    status = C.c_int(-1); stats = np.zeros(8, np.int64); err = C.create_string_buffer(512)
    rc = lib.isort_emul_skip(arr.ctypes.data, b.ctypes.data, 1, shift, config, skip, C.byref(status), stats.ctypes.data, err, 512)
    assert rc == 0 and status.value == 0, (err.value.decode(), status.value)
    wanted = (want >> np.uint32(shift)) <= skip
    n_w = int(wanted.sum())
    assert 0.08 * n < n_w < 0.2 * n and wanted[:n_w].all()               # they are the front of the sorted array
    assert np.array_equal(arr[:n_w], want[:n_w])
    assert np.array_equal(np.sort(arr[n_w:]), np.sort(want[n_w:]))       # the rest: the same elements, in some order
    full = np.zeros(8, np.int64)
    arr2 = ((keys << np.uint32(shift)) | np.arange(n, dtype=np.uint32)).astype(np.uint32)
    lib.isort_emul_skip(arr2.ctypes.data, b.ctypes.data, 1, shift, config, 0xFFFFFFFF, C.byref(status), full.ctypes.data, err, 512)
    assert np.array_equal(arr2, want)
    assert stats[1] < 0.5 * full[1]                                        # and most LDS-tier blocks never exist
