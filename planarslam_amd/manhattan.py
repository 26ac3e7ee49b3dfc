"""Mirror of the reference's Manhattan-frame tracking (Tracking::TrackManhattanFrame, src/Tracking.cc:963) over the C ABI.

Test / bench plumbing, like the other modules of this package: the product is `planar_track_manhattan_frame(_dev)`."""
import numpy as np

from ._lib import Context, check, lib


class Tracking:
    """Only the Manhattan-frame part of the reference's Tracking class."""

    def __init__(self, ctx: Context | None = None):
        self.L = lib()
        self.ctx = ctx or Context(0)

    def TrackManhattanFrame(self, R_last, normals, n_normals, line_dirs, n_lines):
        """R_last (B,3,3) f32; normals (B,S,3) f32 with n_normals (B,); line_dirs (B,T,3) f64 with n_lines (B,).
        Returns dict(R (B,3,3), member_normals (B,S) u8, member_lines (B,T) u8, info (B,8) i32, density (B,3) f32)."""
        R_last = np.ascontiguousarray(R_last, np.float32).reshape(-1, 9)
        B = len(R_last)
        normals = np.ascontiguousarray(normals, np.float32).reshape(B, -1, 3)
        line_dirs = np.ascontiguousarray(line_dirs, np.float64).reshape(B, -1, 3)
        S, T = normals.shape[1], line_dirs.shape[1]
        if S == 0:
            normals, S = np.zeros((B, 1, 3), np.float32), 1
        if T == 0:
            line_dirs, T = np.zeros((B, 1, 3), np.float64), 1
        nn = np.ascontiguousarray(n_normals, np.int32); nl = np.ascontiguousarray(n_lines, np.int32)
        R = np.zeros((B, 9), np.float32); member = np.zeros((B, S + T), np.uint8)
        info = np.zeros((B, 8), np.int32); dens = np.zeros((B, 3), np.float32)
        check(self.L.planar_track_manhattan_frame(self.ctx.h, B, R_last.ctypes.data, normals.ctypes.data, nn.ctypes.data, S, line_dirs.ctypes.data,
                                                  nl.ctypes.data, T, R.ctypes.data, member.ctypes.data, info.ctypes.data, dens.ctypes.data))
        return dict(R=R.reshape(B, 3, 3), member_normals=member[:, :S], member_lines=member[:, S:], info=info, density=dens)
