"""The SE3-rendered synthetic streams (planarslam_amd/synth_se3.py; bench.py's default input): a frame's depth and gray belong to ONE camera pose, so the
depth of frame 0 carried through the true relative pose lands on the depth (and the gray value) frame j shows there.  CPU torch, same code the bench runs on the GPU."""
import numpy as np
import torch

from planarslam_amd import synth, synth_se3 as S3


def test_frames_are_consistent_with_the_camera_motion():
    cam = synth.TUM3
    tex = torch.from_numpy(np.stack([synth.gray_image(1234 + i, 736, 576) for i in range(2)]))
    g, d, Twc = S3.render_streams(torch, tex, 3, 4, cam, seed=0, depth_noise=False, holes=False, pixel_noise=0)
    assert g.shape == (3, 4, 480, 640) and g.dtype == torch.uint8 and d.dtype == torch.int16
    z_all = d.numpy().view(np.uint16).astype(np.float64) / 5000.0
    gray = g.numpy().astype(np.float64)
    ys, xs = np.mgrid[0:480, 0:640]
    for s in range(3):                                   # stream 2 lives in room 0 on another path
        T = S3.relative_pose(Twc[s], 3, 0)
        z = z_all[s, 0]
        assert (z > 0.3).mean() > 0.99                   # a closed room: every ray hits something
        X = np.stack([(xs - cam["cx"]) / cam["fx"] * z, (ys - cam["cy"]) / cam["fy"] * z, z], -1)
        Y = X @ T[:3, :3].T + T[:3, 3]
        u = Y[..., 0] / Y[..., 2] * cam["fx"] + cam["cx"]; v = Y[..., 1] / Y[..., 2] * cam["fy"] + cam["cy"]
        ok = (u > 1) & (u < 638) & (v > 1) & (v < 478)
        ui, vi = np.rint(u[ok]).astype(int), np.rint(v[ok]).astype(int)
        moved = np.hypot(u[ok] - xs[ok], v[ok] - ys[ok])
        assert 1.0 < np.median(moved) < 30.0             # three frames of 1.2 cm / 0.25 deg: pixels, not sub-pixels, not a jump
        dz = np.abs(z_all[s, 3][vi, ui] - Y[..., 2][ok])
        dg = np.abs(gray[s, 3][vi, ui] - gray[s, 0][ok])
        assert np.median(dz) < 0.001 and np.percentile(dz, 90) < 0.02, (np.median(dz), np.percentile(dz, 90))      # occlusion edges are the tail
        assert np.median(dg) <= 3.0, np.median(dg)


def test_loop_index_and_noise_are_deterministic():
    assert [S3.frame_index(i, 4) for i in range(10)] == [0, 0, 1, 2, 3, 2, 1, 0, 1, 2]
    tex = torch.from_numpy(synth.gray_image(99, 736, 576)[None])
    a = S3.render_streams(torch, tex, 1, 2, synth.TUM3, seed=5)
    b = S3.render_streams(torch, tex, 1, 2, synth.TUM3, seed=5)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[1][0, 0] == 0).float().mean() > 0.005      # the holes are there
