import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from planarslam_amd.planes import SurfaceNormals
from planarslam_amd import synth
B=1024
d = np.stack([synth.depth_image(60+i) for i in range(16)]); d = np.concatenate([d]*(B//16))
sn = SurfaceNormals(640,480,B)
dd = torch.from_numpy(d.view(np.int16)).cuda(); out = torch.zeros((B, sn.count, 3), dtype=torch.float32, device="cuda")
for r in range(3):
    torch.cuda.synchronize(); t=time.perf_counter(); sn.compute_dev(dd.data_ptr(), out.data_ptr(), B); sn.ctx.sync() if hasattr(sn.ctx,"sync") else torch.cuda.synchronize(); torch.cuda.synchronize(); print("normals 1024 frames ms", round((time.perf_counter()-t)*1e3,2))
