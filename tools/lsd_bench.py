"""Time the line extractor (planar_lsd_extract_dev) on a synthetic batch; prints per-kernel share via hipEvents around the call."""
import ctypes as C
import sys
import time

import numpy as np
import torch

from planarslam_amd import synth
from planarslam_amd._lib import KEYLINE_DTYPE, Context, check, lib
from planarslam_amd.lines import LineSegment

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ls = LineSegment(640, 480, B, ctx, top_only=int(sys.argv[2]) if len(sys.argv) > 2 else 0)   # argv[2]: 1 = the top-lines mode (planar_lsd_set_top_only)
imgs = torch.from_numpy(synth.gray_batch(B, seed=1234)).to(dev)
kl = torch.zeros(B * 40 * KEYLINE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
desc = torch.zeros(B * 40 * 32, dtype=torch.uint8, device=dev)
eq = torch.zeros(B * 40 * 3, dtype=torch.float64, device=dev)
n = torch.zeros(B, dtype=torch.int32, device=dev)


def run():
    check(lib().planar_lsd_extract_dev(ls.h, imgs.data_ptr(), B, 640, 640 * 480, 40, kl.data_ptr(), desc.data_ptr(), eq.data_ptr(), n.data_ptr()))


for _ in range(2):
    run()
torch.cuda.synchronize()
t = time.time()
K = 3
for _ in range(K):
    run()
torch.cuda.synchronize()
dt = (time.time() - t) / K
regs = [int(ls.read_stage(b, 4)[0]) for b in range(min(B, 8))]
segs = [len(ls.read_stage(b, 3)) for b in range(min(B, 8))]
st = np.stack([ls.read_stage(b, 6) for b in range(min(B, 64))])
print("top-lines mode, first 64 frames: settled", int(st[:, 0].sum()), "redone", int(st[:, 1].sum()), "all regions from the start", int(st[:, 2].sum()))
print(f"LSD+LBD B={B}: {dt * 1e3:.1f} ms/batch = {B / dt:.0f} frames/s | regions/frame {regs} | raw segments/frame {segs} | kept {n[:8].tolist()}")
t = ls.read_stage(0, 5)
print("frame 0 detect cycles: total %d | grow %d | region2rect %d | refine %d | (rect_improve: separate kernel) %d | ordered px %d | grown px %d" % tuple(t[:7]))
print("frame 0 sort cycles: workgroup tier %d | LDS tier %d | radix passes %d" % tuple(t[7:10]))
