"""Where peac_refine's time goes, per frame, on the SE3-rendered frames bench.py times (tools; run through gpurun):
    python tools/peac_refine_probe.py [B]
Phase marks of the refinement kernel (100 MHz wall clock): seeds / erosion, flood fill, final clustering + relabel; queue entries and flood-fill steps per frame;
the launch's duration alone on the device for this B (HIP events of planar_peac_set_profiling)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from planarslam_amd import Context, PlaneDetection, synth_se3  # noqa: E402
from planarslam_amd._lib import check  # noqa: E402
from planarslam_amd.synth import TUM3, gray_image  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
tex = torch.from_numpy(np.stack([gray_image(1234 + i, 736, 576) for i in range(min(16, B))])).to(dev)
_, loop_d, _ = synth_se3.render_streams(torch, tex, B, 2, TUM3, seed=0)
d = loop_d[:, 1].contiguous()
stream = torch.cuda.Stream()
ctx = Context(0, stream=stream.cuda_stream)
pd = PlaneDetection(640, 480, max_batch=B, ctx=ctx)
lab = torch.zeros((B, 480 * 640), dtype=torch.int32, device=dev); pl = torch.zeros((B, 128, 8), dtype=torch.float64, device=dev); npl = torch.zeros(B, dtype=torch.int32, device=dev)
with torch.cuda.stream(stream):
    for rep in range(3):
        if rep == 2:
            check(pd.L.planar_peac_set_profiling(pd.h, 1))
        pd.segment_dev(d.data_ptr(), lab.data_ptr(), pl.data_ptr(), npl.data_ptr(), B)
    torch.cuda.synchronize()
tot = np.zeros(4); nc = C.c_int64()
check(pd.L.planar_peac_get_profile(pd.h, tot.ctypes.data, C.byref(nc)))
print(f"B={B}: launches alone (ms): blocks {tot[0]:.2f}  ahc {tot[1]:.2f}  order {tot[2]:.2f}  refine {tot[3]:.2f}")
t = np.zeros((B, 48), np.int64)
check(pd.L.planar_peac_read_timing(pd.h, B, t.ctypes.data))
ms = lambda a: a / 1e5
seeds, flood, tail = ms(t[:, 4] - t[:, 3]), ms(t[:, 5] - t[:, 4]), ms(t[:, 6] - t[:, 5])
q = t[:, 8]
lab_h = lab.cpu().numpy()
black = (lab_h < 0).mean(1)
print("per frame (ms)      mean    p50    max")
for nm, v in (("seeds+erosion", seeds), ("flood fill", flood), ("cluster+relabel", tail), ("refine total", seeds + flood + tail), ("ahc (clustering kernel)", ms(t[:, 3]))):
    print(f"{nm:24s} {v.mean():6.2f} {np.median(v):6.2f} {v.max():6.2f}")
print(f"queue entries: mean {q.mean():.0f} p50 {np.median(q):.0f} max {q.max()}  -> flood-fill steps of 512 entries: mean {np.ceil(q / 512).mean():.0f}; us per step: {(flood * 1e3 / np.maximum(1, np.ceil(q / 512))).mean():.1f}")
st = t[:, 7]
print(f"flood-fill steps actually taken: mean {st.mean():.0f} p50 {np.median(st):.0f} max {st.max()}; entries per step {(q / np.maximum(st, 1)).mean():.0f}; us per step {(flood * 1e3 / np.maximum(st, 1)).mean():.2f}")
if t[:, 20:25].any():
    names = ("A1 entries+membership", "A2 geometry+chains", "B folds", "B fence+barrier", "C pushes")
    tot = t[:, 20:25].sum(1).astype(float)
    print("flood-fill phases (share of the loop, mean over frames; us per step): " + "  ".join(f"{n} {100 * (t[:, 20 + i] / np.maximum(tot, 1)).mean():.0f}% {(t[:, 20 + i] / 100.0 / np.maximum(st, 1)).mean():.1f}" for i, n in enumerate(names)))
print(f"planes per frame {npl.float().mean().item():.2f}; unlabelled pixels {black.mean() * 100:.1f} %")
