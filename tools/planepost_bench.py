"""Standalone timing of the plane post-processing kernel (planepost.hip) with its in-kernel phase marks: PEAC on B synthetic frames once, then the
plane-cloud launch `reps` times, HIP events around it.  python tools/planepost_bench.py [B] [reps]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from planarslam_amd import Context, PlaneClouds, PlaneDetection  # noqa: E402
from planarslam_amd._lib import check, lib  # noqa: E402
from planarslam_amd.synth import depth_image  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W, H = 640, 480
src = np.stack([depth_image(700 + i) for i in range(8)])
depth = torch.from_numpy(src[np.arange(B) % 8].view(np.int16)).cuda()
st = torch.cuda.Stream()
ctx = Context(0, stream=st.cuda_stream)
det = PlaneDetection(W, H, max_batch=B, ctx=ctx)
pcz = PlaneClouds(W, H, max_batch=B, ctx=ctx)
L = lib()
PS, MP = pcz.pl_stride, pcz.max_points
z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
lab, pls, npl = z((B, H * W), torch.int32), z((B, PS, 8), torch.float64), z((B,), torch.int32)
out = dict(n=z((B,), torch.int32), coef=z((B, PS, 4), torch.float32), src=z((B, PS), torch.int32), off=z((B, PS + 1), torch.int32), pts=z((B, MP, 3), torch.float32),
           status=z((B,), torch.int32))
with torch.cuda.stream(st):
    det.segment_dev(depth.data_ptr(), lab.data_ptr(), pls.data_ptr(), npl.data_ptr(), B)
    torch.cuda.synchronize()
    check(L.planar_plane_clouds_set_timing(pcz.h, 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for r in range(reps):
        e0.record(st)
        pcz.compute_dev(depth.data_ptr(), lab.data_ptr(), pls.data_ptr(), npl.data_ptr(), B, out["n"].data_ptr(), out["coef"].data_ptr(), out["src"].data_ptr(),
                        out["off"].data_ptr(), out["pts"].data_ptr(), out["status"].data_ptr())
        e1.record(st)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    tm = np.zeros((B, 16), np.int64)
    check(L.planar_plane_clouds_read_timing(pcz.h, B, tm.ctypes.data))
ph = np.diff(tm[:, :6], axis=1) / 100.0          # us
print(f"B={B}: launch ms {['%.3f' % m for m in ms]}; status any={bool(out['status'].any())}; planes/frame {npl.float().mean().item():.2f} kept {out['n'].float().mean().item():.2f}; "
      f"voxels/frame {tm[:, 6].mean():.0f}")
for name, col in zip(("clear", "voxel sums", "sort+centroids", "refit", "compaction"), ph.T):
    print(f"  {name:15s} mean {col.mean():9.1f} us   max {col.max():9.1f} us")
print(f"  wave 0: row loops {tm[:, 8].mean() / 1e3:.0f} kcycles, of which parked-run insertions {tm[:, 11].mean() / 1e3:.0f} kcycles in {tm[:, 10].mean():.0f} executions; tile ends {tm[:, 9].mean() / 1e3:.0f} kcycles")
print(f"  whole workgroup mean {(tm[:, 5] - tm[:, 0]).mean() / 100.0:.1f} us")
