"""Counter-pass workload: the three extraction stages (PEAC + surface normals, LSD/LBD + 3-D lines, ORB) of 1024 frames, three launches each, one stage after the other on
one stream.  Used for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (separate passes): the counter mode serialises every dispatch, and the full bench.py step
(thousands of small launches over eight streams) does not finish within the profiling time limit under it.  The frames are the bench's own panning windows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from planarslam_amd.synth import TUM3, pan_offset, stream_canvases  # noqa: E402

B, W, H, MARGIN, NC = 1024, 640, 480, 48, 64
canv_g, canv_d = stream_canvases(NC, 0, W + 2 * MARGIN, H + 2 * MARGIN, procs=16)
import torch  # noqa: E402

from planarslam_amd.track import TrackPipeline  # noqa: E402

dev = torch.device("cuda", 0)
tp = TrackPipeline(B, torch, 0, depth=0, cam=TUM3, W=W, H=H)
g = torch.from_numpy(canv_g).to(dev); d = torch.from_numpy(canv_d.view(np.int16)).to(dev)
frames = torch.zeros((B, H, W), dtype=torch.uint8, device=dev); depths = torch.zeros((B, H, W), dtype=torch.int16, device=dev)
for rep in range(3):
    ox, oy = pan_offset(rep, MARGIN)
    for a in range(0, B, NC):
        frames[a:a + NC].copy_(g[:, oy:oy + H, ox:ox + W]); depths[a:a + NC].copy_(d[:, oy:oy + H, ox:ox + W])
    with torch.cuda.stream(tp.stream):
        tp.step(rep, frames, depths)
        tp.drain()
    torch.cuda.synchronize()
tp.check()
print("done")
