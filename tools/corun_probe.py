"""Do the two side chains of the frame pipeline overlap on the device?  Runs the plane chain (PlaneDetection.segment_dev + PlaneClouds + SurfaceNormals) and the line
chain (planar_lsd_preprocess_dev + planar_lsd_detect_dev) for B frames, each on its own stream: each alone, then both together; wall-clock per round.
  PYTHONPATH=. python tools/corun_probe.py [B] [rounds]        (under rocprofv3 --kernel-trace the timeline of the last round shows which kernels overlap)"""
import sys
import time

import numpy as np
import torch

from planarslam_amd import Context, PlaneDetection
from planarslam_amd._lib import check, lib
from planarslam_amd.lines import LineSegment
from planarslam_amd.planes import PlaneClouds, SurfaceNormals
from planarslam_amd.synth import depth_image, gray_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = sys.argv[3] if len(sys.argv) > 3 else "all"
dev = torch.device("cuda:0")
sp, sl = torch.cuda.Stream(), torch.cuda.Stream()
cp, cl = Context(0, stream=sp.cuda_stream), Context(0, stream=sl.cuda_stream)
pd = PlaneDetection(640, 480, max_batch=B, ctx=cp); pc = PlaneClouds(640, 480, max_batch=B, ctx=cp); sn = SurfaceNormals(640, 480, max_batch=B, ctx=cp)
ls = LineSegment(640, 480, B, cl)
nsrc = min(B, 32)
dsrc = np.stack([depth_image(4321 + i) for i in range(nsrc)]); gsrc = np.stack([gray_image(4321 + i) for i in range(nsrc)])
idx = np.arange(B) % nsrc
d = torch.from_numpy(dsrc[idx].view(np.int16)).to(dev); g = torch.from_numpy(gsrc[idx]).to(dev)
lab = torch.zeros((B, 480 * 640), dtype=torch.int32, device=dev); pls = torch.zeros((B, pd.max_planes, 8), dtype=torch.float64, device=dev); n = torch.zeros(B, dtype=torch.int32, device=dev)
MP = 4096
PS = pd.max_planes
o = dict(n=torch.zeros(B, dtype=torch.int32, device=dev), coef=torch.zeros((B, PS, 4), device=dev), src=torch.zeros((B, PS), dtype=torch.int32, device=dev),
         off=torch.zeros((B, PS + 1), dtype=torch.int32, device=dev), pts=torch.zeros((B, MP, 3), device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev))
nrm = torch.zeros((B, sn.count, 3), device=dev)
from planarslam_amd._lib import KEYLINE_DTYPE
kl = torch.zeros((B, 40, KEYLINE_DTYPE.itemsize), dtype=torch.uint8, device=dev); ld = torch.zeros((B, 40, 32), dtype=torch.uint8, device=dev)
le = torch.zeros((B, 40, 3), dtype=torch.float64, device=dev); nl = torch.zeros(B, dtype=torch.int32, device=dev)
L = lib()


def planes():
    pd.segment_dev(d.data_ptr(), lab.data_ptr(), pls.data_ptr(), n.data_ptr(), B)
    pc.compute_dev(d.data_ptr(), lab.data_ptr(), pls.data_ptr(), n.data_ptr(), B, o["n"].data_ptr(), o["coef"].data_ptr(), o["src"].data_ptr(), o["off"].data_ptr(),
                   o["pts"].data_ptr(), o["status"].data_ptr())
    sn.compute_dev(d.data_ptr(), nrm.data_ptr(), B)


def lines():
    check(L.planar_lsd_preprocess_dev(ls.h, g.data_ptr(), B, 640, 640 * 480))
    check(L.planar_lsd_detect_dev(ls.h, B, 40, kl.data_ptr(), ld.data_ptr(), le.data_ptr(), nl.data_ptr()))


def timed(fs):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(R):
        for f in fs: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / R


planes(); lines(); torch.cuda.synchronize()
if which in ("all", "alone"):
    print(f"B={B}: plane chain alone %.1f ms   line chain alone %.1f ms" % (timed([planes]), timed([lines])))
if which in ("all", "both"):
    print(f"B={B}: both chains, two streams %.1f ms per round" % timed([planes, lines]))
