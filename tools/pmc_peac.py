"""Counter-pass workload for the dominant stage only: PlaneDetection (peac_blocks + peac_ahc + peac_order + peac_refine) on B frames of the bench's panning windows,
three launches through the host-pointer entry point (one stream, no torch).  `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python tools/pmc_peac.py [B]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from planarslam_amd import PlaneDetection  # noqa: E402
from planarslam_amd.synth import pan_offset, stream_canvases  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H, MARGIN, NC = 640, 480, 48, 32
_, canv_d = stream_canvases(NC, 0, W + 2 * MARGIN, H + 2 * MARGIN, procs=8)
pd = PlaneDetection(W, H, max_batch=B)
for rep in range(3):
    ox, oy = pan_offset(rep, MARGIN)
    d = np.ascontiguousarray(np.concatenate([canv_d[:, oy:oy + H, ox:ox + W]] * (B // NC))[:B])
    res = pd.run(d)
    print("launch", rep, "planes per frame %.2f" % np.mean([len(r[0]) for r in res]), flush=True)
