"""ctypes binding of libplanar_hip.so (the C ABI in include/planar_abi.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible the calls fail
loudly (PlanarError / OSError)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLANAR_HIP_LIB: a developer override (e.g. the -DPLANAR_PEAC_TIMING build made by `make -C planarslam_amd/csrc timing`); never a fallback
LIB_PATH = os.path.join(_HERE, "libplanar_hip.so")
if os.environ.get("PLANAR_HIP_LIB"):
    _ov = os.path.abspath(os.environ["PLANAR_HIP_LIB"])
    if os.path.dirname(_ov) != _HERE:      # only builds that sit next to the product library (in-tree: what the driver records as loaded native code)
        raise ImportError(f"PLANAR_HIP_LIB={_ov}: a developer override must be a library inside {_HERE}")
    import warnings
    warnings.warn(f"planarslam_amd: loading the developer build {_ov} instead of libplanar_hip.so (PLANAR_HIP_LIB)")
    LIB_PATH = _ov

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


class PlanarError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libplanar_hip error {code}: {msg}")
        self.code = code


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class PoseParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("angle_info", C.c_double), ("distance_info", C.c_double), ("parallel_info", C.c_double),
                ("vertical_info", C.c_double), ("plane_chi", C.c_double), ("vp_chi", C.c_double)]


class PoseBatch(C.Structure):
    _fields_ = [("B", C.c_int32), ("max_points", C.c_int32), ("max_lines", C.c_int32), ("max_planes", C.c_int32)] + [
        (n, C.c_void_p) for n in ("n_points", "n_lines", "n_planes", "pt_valid", "pt_xw", "pt_obs", "pt_inv_sigma2", "ln_valid",
                                  "ln_obs", "ln_xw", "pl_meas", "pl_valid", "pl_world", "Tcw_in", "Tcw_out", "pt_outlier",
                                  "ln_outlier", "pl_outlier", "n_inliers", "lm_iters")]


class BAProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("kf_Tcw", C.c_void_p), ("kf_fixed", C.c_void_p), ("n_lm", C.c_int32), ("lm_type", C.c_void_p),
                ("lm_init", C.c_void_p), ("n_edges", C.c_int32), ("e_kf", C.c_void_p), ("e_lm", C.c_void_p), ("e_type", C.c_void_p),
                ("e_meas", C.c_void_p), ("e_inv_sigma2", C.c_void_p)]


KEYLINE_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"), ("response", "<f4"),
                          ("size", "<f4"), ("start_x", "<f4"), ("start_y", "<f4"), ("end_x", "<f4"), ("end_y", "<f4"), ("s_oct_x", "<f4"),
                          ("s_oct_y", "<f4"), ("e_oct_x", "<f4"), ("e_oct_y", "<f4"), ("line_length", "<f4"), ("num_pixels", "<i4")])
MAX_LEVELS = 16


class FrameView(C.Structure):
    _fields_ = [("B", C.c_int32), ("stride", C.c_int32), ("n", C.c_void_p), ("keys_un", C.c_void_p), ("u_right", C.c_void_p),
                ("desc", C.c_void_p), ("blocked", C.c_void_p), ("Tcw", C.c_void_p), ("min_x", C.c_float), ("max_x", C.c_float),
                ("min_y", C.c_float), ("max_y", C.c_float), ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float), ("fx", C.c_float),
                ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("b", C.c_float),
                ("scale_factors", C.c_float * MAX_LEVELS)]


class LastFrameView(C.Structure):
    _fields_ = [("stride", C.c_int32)] + [(n, C.c_void_p) for n in ("n", "Tcw", "usable", "xw", "octave", "angle", "mp_desc", "mp_observed")]


class MapProbes(C.Structure):
    _fields_ = [("stride", C.c_int32)] + [(n, C.c_void_p) for n in ("n", "in_view", "proj_x", "proj_y", "proj_xr", "level", "view_cos", "desc",
                                                                   "observed")]


class TrackMatches(C.Structure):
    _fields_ = [("B", C.c_int32), ("stride", C.c_int32), ("mp_stride", C.c_int32), ("n_levels", C.c_int32), ("n", C.c_void_p), ("keys_un", C.c_void_p),
                ("u_right", C.c_void_p), ("pt_match", C.c_void_p), ("mp_xw", C.c_void_p), ("mp_valid", C.c_void_p), ("inv_level_sigma2", C.c_float * MAX_LEVELS),
                ("ln_stride", C.c_int32), ("ml_stride", C.c_int32), ("n_lines", C.c_void_p), ("line_eq", C.c_void_p), ("ln_match", C.c_void_p), ("ml_xw6", C.c_void_p),
                ("pl_stride", C.c_int32), ("mpl_stride", C.c_int32), ("mpl_shared", C.c_int32), ("n_planes", C.c_void_p), ("pl_coef", C.c_void_p),
                ("pl_match", C.c_void_p), ("mpl_coef", C.c_void_p), ("Tcw", C.c_void_p)]


class BAResult(C.Structure):
    _fields_ = [("kf_Tcw", C.c_void_p), ("lm", C.c_void_p), ("e_outlier", C.c_void_p), ("lm_iterations", C.c_int32), ("stopped", C.c_int32)]


_SIGS = {
    # name: (restype, argtypes)
    "planar_last_error": (C.c_char_p, []),
    "planar_abi_version": (C.c_int, []),
    "planar_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "planar_ctx_destroy": (None, [C.c_void_p]),
    "planar_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "planar_ctx_get_stream": (C.c_void_p, [C.c_void_p]),
    "planar_cu_stream_create": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_cu_stream_destroy": (None, [C.c_void_p]),
    "planar_ctx_set_seq_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "planar_ctx_sync": (C.c_int, [C.c_void_p]),
    "planar_orb_create": (C.c_int, [C.c_void_p, C.POINTER(OrbParams), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_orb_destroy": (None, [C.c_void_p]),
    "planar_orb_check": (C.c_int, [C.c_void_p]),
    "planar_peac_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_peac_get_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "planar_orb_max_keypoints": (C.c_int, [C.c_void_p]),
    "planar_orb_get_scale_factors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_orb_level_size": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "planar_orb_features_per_level": (C.c_int, [C.c_void_p, C.c_void_p]),
    "planar_orb_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_orb_extract_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_orb_read_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "planar_orb_read_blurred": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "planar_orb_read_candidates": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "planar_orb_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_orb_profile_num_launches": (C.c_int, [C.c_void_p]),
    "planar_orb_profile_launch_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "planar_orb_get_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "planar_hamming_knn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_hamming_knn_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_match_orb_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_match_orb_points_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_lsd_search_by_descriptor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_lsd_search_by_descriptor_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_search_by_projection_frame": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.POINTER(LastFrameView), C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_search_by_projection_frame_dev": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.POINTER(LastFrameView), C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_search_by_projection_map": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.POINTER(MapProbes), C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "planar_search_by_projection_map_dev": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.POINTER(MapProbes), C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "planar_fuse_search": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3),
    "planar_fuse_search_dev": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3),
    "planar_lsd_fuse_search": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3),
    "planar_lsd_fuse_search_dev": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3),
    "planar_is_in_frustum_points": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_float, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 6),
    "planar_is_in_frustum_points_dev": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_float, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 6),
    "planar_is_in_frustum_lines": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_float, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 4),
    "planar_is_in_frustum_lines_dev": (C.c_int, [C.c_void_p, C.POINTER(FrameView), C.c_float, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 4),
    "planar_search_by_bow": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] +
                             [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_search_by_bow_dev": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] +
                                 [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_lsd_search_by_projection": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] +
                                        [C.c_void_p] * 6 + [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "planar_lsd_search_by_projection_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] +
                                            [C.c_void_p] * 6 + [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "planar_plane_search_by_coefficients": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_void_p] * 4),
    "planar_plane_search_by_coefficients_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_void_p] * 4),
    "planar_lsd_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_lsd_destroy": (None, [C.c_void_p]),
    "planar_lsd_max_segments": (C.c_int, []),
    "planar_lsd_set_tie_order": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_lsd_set_top_only": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_undistort_keypoints": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p]),
    "planar_undistort_keypoints_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p]),
    "planar_plane_clouds_sort_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "planar_plane_clouds_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_plane_clouds_get_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_lsd_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_lsd_get_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_lsd_scaled_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "planar_lsd_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_lsd_extract_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_lsd_preprocess_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64]),
    "planar_lsd_detect_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_lsd_check": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_lsd_read_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "planar_debug_std_sort_desc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "planar_peac_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_peac_destroy": (None, [C.c_void_p]),
    "planar_peac_max_planes": (C.c_int, []),
    "planar_peac_segment": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_peac_segment_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_peac_check": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_peac_set_variant": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "planar_track_manhattan_frame": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_track_manhattan_frame_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_peac_read_timing": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "planar_reset_matches_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "planar_blocked_mask_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "planar_merge_matches_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "planar_manhattan_pose_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_keypoint_fields_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "planar_add_scalar_i32_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "planar_copy_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "planar_peac_debug_layout": (C.c_int, [C.c_void_p, C.c_void_p]),
    "planar_peac_debug_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "planar_comm_unique_id": (C.c_int, [C.c_void_p]),
    "planar_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_vocab_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "planar_vocab_destroy": (None, [C.c_void_p]),
    "planar_vocab_words": (C.c_int, [C.c_void_p]),
    "planar_bow_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "planar_bow_transform_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "planar_is_line_good": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 8),
    "planar_is_line_good_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 8),
    "planar_normals_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_normals_destroy": (None, [C.c_void_p]),
    "planar_normals_count": (C.c_int, [C.c_void_p]),
    "planar_normals_grid": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "planar_normals_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "planar_normals_compute_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                            C.c_void_p, C.c_int]),
    "planar_plane_clouds_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_plane_clouds_destroy": (None, [C.c_void_p]),
    "planar_plane_clouds_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "planar_plane_clouds_read_timing": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "planar_plane_clouds_stride": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "planar_plane_clouds_set_plane_window": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "planar_plane_clouds_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_double, C.c_float] + [C.c_void_p] * 8),
    "planar_plane_clouds_compute_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_double, C.c_float] + [C.c_void_p] * 9),
    "planar_plane_clouds_last_status": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "planar_plane_refit": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_flag_matched_plane_points": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_flag_matched_plane_points_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                      C.c_void_p]),
    "planar_merge_plane_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "planar_distinctive_descriptors": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_distinctive_descriptors_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "planar_update_normal_and_depth": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 3),
    "planar_update_normal_and_depth_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 3),
    "planar_comm_create_hosted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "planar_comm_destroy": (None, [C.c_void_p]),
    "planar_local_ba": (C.c_int, [C.c_void_p, C.POINTER(BAProblem), C.POINTER(PoseParams), C.c_int, C.c_int, C.POINTER(BAResult), C.c_void_p, C.c_void_p]),
    "planar_stereo_from_rgbd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64] + [C.c_float] * 6 + [C.c_void_p] * 5),
    "planar_stereo_from_rgbd_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64] + [C.c_float] * 6 + [C.c_void_p] * 5),
    "planar_pose_assemble": (C.c_int, [C.c_void_p, C.POINTER(TrackMatches), C.POINTER(PoseBatch)]),
    "planar_pose_assemble_dev": (C.c_int, [C.c_void_p, C.POINTER(TrackMatches), C.POINTER(PoseBatch)]),
    "planar_discard_outliers": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_discard_outliers_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "planar_pose_opt": (C.c_int, [C.c_void_p, C.POINTER(PoseBatch), C.POINTER(PoseParams), C.c_int, C.c_int, C.c_int]),
    "planar_pose_opt_dev": (C.c_int, [C.c_void_p, C.POINTER(PoseBatch), C.POINTER(PoseParams), C.c_int, C.c_int, C.c_int]),
}

TEST_HOOKS = ("planar_debug_std_sort_desc",)     # declared under PLANAR_TEST_HOOKS in include/planar_abi.h
_LIB = None


def exported_symbols():
    """Every symbol include/planar_abi.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(n for n in _SIGS if n not in TEST_HOOKS)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if name in TEST_HOOKS and not hasattr(L, name):
                continue                       # only the test build (libplanar_hip_paranoid.so) exports these
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(rc: int):
    if rc < 0:
        raise PlanarError(rc, lib().planar_last_error().decode(errors="replace"))
    return rc


def cu_stream_create(device: int, n_cus: int, first: int = 0, total: int = 256) -> int:
    """A HIP stream restricted to n_cus compute units, bits first .. first + n_cus - 1 of the driver's CU numbering (which deals consecutive bits round-robin to the
    XCDs, so any contiguous run is spread evenly over them) -> the hipStream_t as an int (planar_cu_stream_destroy frees it)."""
    import numpy as np
    words = np.zeros((total + 31) // 32, np.uint32)
    for i in range(first, min(first + n_cus, total)):
        words[i >> 5] |= np.uint32(1 << (i & 31))
    out = C.c_void_p()
    check(lib().planar_cu_stream_create(device, words.ctypes.data, len(words), C.byref(out)))
    return out.value


class Context:
    """planar_ctx: one HIP device + stream.  Not re-entrant (one per host thread)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.L = lib()
        h = C.c_void_p()
        check(self.L.planar_ctx_create(C.byref(h), device))
        self.h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, hip_stream: int | None):
        check(self.L.planar_ctx_set_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_seq_stream(self, hip_stream: int | None):
        """the side stream (e.g. a CU-masked one, cu_stream_create) the context launches its one-wavefront-per-frame kernels on; None: its own stream"""
        check(self.L.planar_ctx_set_seq_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None))

    def sync(self):
        check(self.L.planar_ctx_sync(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
