"""Seeded problems shared by tools/gen_golden_frame.py (which runs the REAL reference function bodies, oracle/_ref/ref_frame, on them and stores
their outputs in tests/golden/frame_ref.npz), tests/test_oracle_frame_ref.py (oracle vs those outputs) and the GPU tests (HIP vs those outputs)."""
from planarslam_amd import synth

# Tracking::TrackManhattanFrame: name -> manhattan_scene kwargs
MANHATTAN_CASES = {
    "three_axes": dict(B=12, seed=31),
    "no_z": dict(B=4, seed=5, drop_axis=2, clutter=0.0),                       # v3 = v1 x v2
    "no_x_small": dict(B=4, seed=6, drop_axis=0, clutter=0.1, n_normals=700, n_lines=7),   # v1 = v3 x v2
    "no_y_tiny": dict(B=3, seed=8, drop_axis=1, n_normals=300, n_lines=3),     # v2 = v1 x v3
    "one_axis_only": dict(B=4, seed=13, drop_axis=(0, 2), clutter=0.0),        # numDirectionFound < 2: the aliased R_cm comes back (Tracking.cc:1066-1074)
    "mostly_clutter": dict(B=4, seed=9, n_normals=300, clutter=0.9),
    "big_tilt": dict(B=4, seed=17, tilt_deg=11.0),
}


def frustum_case():
    fr = synth.guided_frame(B=3, N=1000, seed=121)
    return synth.guided_local_map(fr, seed=122, n_points=4000, n_lines=500)


def frustum_scale():
    import numpy as np
    return float(np.float32(np.log(np.float32(1.2)))), 8       # Frame::mfLogScaleFactor = log(mfScaleFactor) (float), mnScaleLevels


def stereo_case():
    """Keypoints (incl. image-border positions and zero-depth holes), a depth frame and a pose for Frame::ComputeStereoFromRGBD / UnprojectStereo."""
    import numpy as np
    from planarslam_amd._lib import KP_DTYPE
    rng = np.random.default_rng(33)
    B, S = 3, 1100
    depth = np.stack([synth.depth_image(4321 + i) for i in range(B)])
    keys = np.zeros((B, S), KP_DTYPE)
    keys["x"] = rng.uniform(0, 639.99, (B, S)).astype(np.float32); keys["y"] = rng.uniform(0, 479.99, (B, S)).astype(np.float32)
    keys["x"][:, :4] = [0.0, 639.0, 0.49, 638.51]; keys["y"][:, :4] = [0.0, 479.0, 478.51, 0.49]
    keys["octave"] = rng.integers(0, 8, (B, S))
    n = np.array([S, 1000, 37], np.int32)
    Tcw = np.stack([synth._se3(rng, 0.3, rng.normal(0, 0.5, 3)).astype(np.float32).ravel() for _ in range(B)])
    return keys, n, depth, Tcw


# Camera.k1 .. Camera.k3 of the reference's Examples/RGB-D yaml files: TUM1, TUM2, and TUM3 (all zero -> mvKeysUn = mvKeys)
DIST = {"TUM1": ((517.306408, 516.469215, 318.643040, 255.313989), (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)),
        "TUM2": ((520.908620, 521.007327, 325.141442, 249.701764), (0.231222, -0.784899, -0.003257, -0.000105, 0.917205)),
        "TUM3": ((535.4, 539.2, 320.1, 247.6), (0.0, 0.0, 0.0, 0.0, 0.0))}


def undistort_case(seed=35):
    """Keypoints over the whole image incl. its four corners (Frame::ComputeImageBounds runs the same call on them) for Frame::UndistortKeyPoints"""
    import numpy as np
    from planarslam_amd._lib import KP_DTYPE
    rng = np.random.default_rng(seed)
    B, S = 3, 1300
    keys = np.zeros((B, S), KP_DTYPE)
    keys["x"] = rng.uniform(0, 639.99, (B, S)).astype(np.float32); keys["y"] = rng.uniform(0, 479.99, (B, S)).astype(np.float32)
    keys["x"][:, :4] = [0.0, 640.0, 0.0, 640.0]; keys["y"][:, :4] = [0.0, 0.0, 480.0, 480.0]
    for f in ("size", "angle", "response"):
        keys[f] = rng.uniform(0, 300, (B, S)).astype(np.float32)
    keys["octave"] = rng.integers(0, 8, (B, S)); keys["class_id"] = -1
    return keys, np.array([S, 1024, 5], np.int32)


def manhattan_pose_case(n=64, seed=31):
    """Rotation_cm [n,9], MF_can [n,9] (unit-quaternion rotations, a few of them only slightly apart as on consecutive frames) and mTcw [n,16]"""
    import numpy as np
    rng = np.random.default_rng(seed)

    def rot(q):
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
        w, x, y, z = q.T
        return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).astype(np.float32)
    q0 = rng.normal(size=(n, 4))
    q1 = np.where((np.arange(n) % 2 == 0)[:, None], q0 + rng.normal(scale=0.01, size=(n, 4)), rng.normal(size=(n, 4)))
    T = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (n, 1)); T[:, [3, 7, 11]] = rng.normal(size=(n, 3)).astype(np.float32)
    T[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]] = rot(rng.normal(size=(n, 4)))
    return rot(q0), rot(q1), T
