"""How often pcl::VoxelGrid's std::sort ends in libstdc++'s heap-sort fallback on the bench's frames (windows over the synthetic canvases), and how long the
fallback kernel takes alone.  python tools/heap_stats.py [frames]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from planarslam_amd import Context, PlaneClouds, PlaneDetection, synth  # noqa: E402
from planarslam_amd._lib import check, lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 640, 480
_, dcan = synth.stream_canvases(min(B, 64), 0, 736, 576, procs=16)
ox, oy = synth.pan_offset(3)
d = np.stack([dcan[i % len(dcan)][oy + (i // len(dcan)) * 2:oy + (i // len(dcan)) * 2 + H, ox:ox + W] for i in range(B)])
depth = torch.from_numpy(np.ascontiguousarray(d).view(np.int16)).cuda()
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
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        pcz.compute_dev(depth.data_ptr(), lab.data_ptr(), pls.data_ptr(), npl.data_ptr(), B, out["n"].data_ptr(), out["coef"].data_ptr(), out["src"].data_ptr(),
                        out["off"].data_ptr(), out["pts"].data_ptr(), out["status"].data_ptr())
        e1.record(st)
        torch.cuda.synchronize()
        print(f"plane clouds, B = {B}: {e0.elapsed_time(e1):.2f} ms")
stats = np.zeros((B, 4), np.int64)
check(L.planar_plane_clouds_sort_stats(pcz.h, B, stats.ctypes.data))
hit = stats[:, 0] > 0
print(f"frames with a heap-sort fallback: {hit.sum()} of {B}; jobs {stats[:, 0].sum()}, elements {stats[:, 1].sum()}, longest {stats[:, 2].max()}; status any {bool(out['status'].any())}")
print("per frame with a fallback (jobs, elements, longest, blocks):", stats[hit][:24].tolist())
print("LDS-tier blocks per frame: mean %.1f max %d" % (stats[:, 3].mean(), stats[:, 3].max()))
