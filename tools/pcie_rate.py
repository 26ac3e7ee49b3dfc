"""PCIe-inclusive rate of the host-pointer entry points: B frames handed over as host buffers, results back on the host."""
import sys
import time

import numpy as np

from planarslam_amd import Context, ORBextractor, PlaneDetection
from planarslam_amd.lines import LineSegment
from planarslam_amd.synth import depth_image, gray_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = Context(0)
gray = gray_batch(B, seed=1234)
depth = np.stack([depth_image(4321 + (i % 8)) for i in range(B)])
ex = ORBextractor(1000, 1.2, 8, 20, 7, width=640, height=480, max_batch=B, ctx=ctx)
ls = LineSegment(640, 480, B, ctx)
pd = PlaneDetection(640, 480, max_batch=B, ctx=ctx)
for name, fn in (("ORB", lambda: ex(gray)), ("LSD+LBD", lambda: ls.ExtractLineSegment(gray)), ("PEAC", lambda: pd.run(depth))):
    fn()
    t = time.perf_counter()
    for _ in range(3):
        fn()
    dt = (time.perf_counter() - t) / 3
    print(f"{name}: host-pointer entry point, B={B}: {dt * 1e3:.1f} ms/batch = {B / dt:.0f} frames/s (H2D + kernels + D2H + sync)")
