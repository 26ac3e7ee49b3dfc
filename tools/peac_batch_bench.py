"""Stand-alone timing of the plane extractor on a batch (default 1024 frames, 32 distinct noisy depth images tiled): per-launch milliseconds from the handle's own HIP events
(peac_blocks, clustering = peac_ahc3 + peac_ahc2, peac_order, peac_refine), the launches alone on the device.   PYTHONPATH=. python tools/peac_batch_bench.py [B]"""
import ctypes as C
import sys

import numpy as np
import torch

from planarslam_amd import Context, PlaneDetection
from planarslam_amd._lib import check, lib
from planarslam_amd.synth import depth_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
pd = PlaneDetection(640, 480, max_batch=B, ctx=ctx)
src = np.stack([depth_image(4321 + i) for i in range(min(B, 32))])
d = torch.from_numpy(src[np.arange(B) % len(src)].view(np.int16)).to(dev)
lab = torch.zeros((B, 480 * 640), dtype=torch.int32, device=dev); pls = torch.zeros((B, pd.max_planes, 8), dtype=torch.float64, device=dev); n = torch.zeros(B, dtype=torch.int32, device=dev)
L = lib()
for _ in range(2):
    pd.segment_dev(d.data_ptr(), lab.data_ptr(), pls.data_ptr(), n.data_ptr(), B)
torch.cuda.synchronize()
check(L.planar_peac_set_profiling(pd.h, 1))
K = 3
for _ in range(K):
    pd.segment_dev(d.data_ptr(), lab.data_ptr(), pls.data_ptr(), n.data_ptr(), B)
tot = np.zeros(4); nc = C.c_int64()
check(L.planar_peac_get_profile(pd.h, tot.ctypes.data, C.byref(nc)))
check(L.planar_peac_check(pd.h, B))
print(f"PEAC B={B}: blocks %.2f  clustering %.2f  order %.2f  refine %.2f ms per launch (mean of {K}); planes/frame %.2f" % (*(tot / K), float(n.float().mean().item())))
