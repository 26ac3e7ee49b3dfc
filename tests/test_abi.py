"""CPU test: libplanar_hip.so loads and exports every symbol include/planar_abi.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "planar_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef PLANAR_TEST_HOOKS.*?#endif", "", src, flags=re.S)      # hooks only the test build (libplanar_hip_paranoid.so) exports
    return sorted(set(re.findall(r"\b(planar_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from planarslam_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 20
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from planarslam_amd import _lib
    assert sorted(_lib.exported_symbols()) == declared_symbols()


def test_version_and_error_string_without_gpu():
    from planarslam_amd import _lib
    L = _lib.lib()
    assert L.planar_abi_version() >= 100
    assert isinstance(L.planar_last_error(), bytes)


def test_argument_errors_are_reported_without_touching_a_device():
    """Error convention of the boundary: negative code + planar_last_error(), never a crash.  Null / out-of-range arguments are
    rejected before any HIP call, so this runs on a machine without a GPU."""
    import numpy as np
    from planarslam_amd import _lib
    from planarslam_amd._lib import FrameView, LastFrameView, MapProbes
    L = _lib.lib()
    EINVAL = -1
    one = np.zeros(16, np.int32)
    p = one.ctypes.data
    fv, lv, mp = FrameView(), LastFrameView(), MapProbes()
    calls = [
        lambda: L.planar_hamming_knn(None, p, p, 1, p, p, 1, 1, 1, p, p),
        lambda: L.planar_match_orb_points(None, p, p, 1, p, p, 1, p, p, 1, p, p),
        lambda: L.planar_search_by_projection_frame(None, ctypes.byref(fv), ctypes.byref(lv), 15.0, 0, 1, p, p),
        lambda: L.planar_search_by_projection_map(None, ctypes.byref(fv), ctypes.byref(mp), 1.0, 0.8, p, p),
        lambda: L.planar_search_by_bow(None, 1, p, 1, p, p, p, p, p, 1, p, p, p, 0.7, 1, p, p),
        lambda: L.planar_lsd_search_by_projection(None, 1, p, 1, p, p, p, p, 1, p, p, p, p, p, p, p, 8, 1.0, 0.6, p, p),
        lambda: L.planar_plane_search_by_coefficients(None, 1, p, 1, p, p, 0, p, 1, p, p, p, 1, p, p, p, p, p, p),
        lambda: L.planar_is_in_frustum_points(None, ctypes.byref(fv), 0.18, 8, p, 1, p, p, p, p, p, 0.5, p, p, p, p, p, p),
        lambda: L.planar_lsd_extract(None, p, 1, 640, 640 * 480, 40, p, p, p, p),
        lambda: L.planar_lsd_create(None, 640, 480, 1, ctypes.byref(ctypes.c_void_p())),
        lambda: L.planar_peac_create(None, 640, 480, 1, ctypes.byref(ctypes.c_void_p())),
        lambda: L.planar_pose_opt(None, None, None, 0, 4, 10),
        lambda: L.planar_local_ba(None, None, None, 5, 10, None, None, None),
        # round-2 entry points
        lambda: L.planar_normals_create(None, 640, 480, 1, ctypes.byref(ctypes.c_void_p())),
        lambda: L.planar_plane_clouds_create(None, 640, 480, 1, 4096, ctypes.byref(ctypes.c_void_p())),
        lambda: L.planar_plane_clouds_compute(None, p, 1, 640, 640 * 480, 535.4, 539.2, 320.1, 247.6, 0.0002, p, p, p, 0.05, 0.1, p, p, p, p, p, None, None, None),
        lambda: L.planar_plane_refit(None, 1, p, p, 0.05, p, p, None),
        lambda: L.planar_flag_matched_plane_points(None, 1, p, p, p, p, 1, p, 1, 1, p, None),
        lambda: L.planar_merge_plane_points(None, p, p, 1, p, 1, 0.1, p, 1, p),
        lambda: L.planar_distinctive_descriptors(None, 1, p, p, p, None),
        lambda: L.planar_update_normal_and_depth(None, 1, p, 1, p, None, p, p, None, None, p, 8, p, p, p),
        lambda: L.planar_is_line_good(None, 1, p, p, 40, p, 640, 480, 640, 640 * 480, 0.0002, 535.4, 539.2, 320.1, 247.6, p, p, p, p, p, p, p, p),
        lambda: L.planar_bow_transform(None, None, p, 1, 1, 4, p, p, p, p, p, p),
        lambda: L.planar_track_manhattan_frame(None, 1, p, p, p, 1, p, p, 1, p, None, None, None),
        # round-3 entry points
        lambda: L.planar_fuse_search(None, ctypes.byref(fv), p, 0.18, 8, p, 1, 0, p, p, p, p, p, p, 3.0, p, None, p),
        lambda: L.planar_fuse_search_dev(None, ctypes.byref(fv), p, 0.18, 8, p, 1, 0, p, p, p, p, p, p, 3.0, p, None, p),
        lambda: L.planar_lsd_fuse_search(None, ctypes.byref(fv), 0.18, 8, p, 1, p, p, p, 1, 0, p, p, p, p, p, p, 3.0, p, None, p),
        lambda: L.planar_lsd_fuse_search_dev(None, ctypes.byref(fv), 0.18, 8, p, 1, p, p, p, 1, 0, p, p, p, p, p, p, 3.0, p, None, p),
    ]
    for i, c in enumerate(calls):
        rc = c()
        assert rc == EINVAL, (i, rc)
        assert len(L.planar_last_error()) > 0


def test_product_library_carries_no_test_hooks():
    """planar_debug_* entry points live in the test build only (make paranoid, -DPLANAR_TEST_HOOKS)"""
    from planarslam_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    assert not any(hasattr(L, s) for s in _lib.TEST_HOOKS)
