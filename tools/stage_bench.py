import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch, ctypes as C
from planarslam_amd import Context, PlaneDetection, Optimizer, ORBextractor
from planarslam_amd._lib import PoseBatch, lib, check
from planarslam_amd.synth import depth_image, pose_batch, TUM3, gray_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
stream = torch.cuda.Stream()
ctx = Context(0, stream=stream.cuda_stream)
# PEAC
pd = PlaneDetection(640, 480, max_batch=B, ctx=ctx)
base = np.stack([depth_image(4321 + i) for i in range(8)])
d = torch.from_numpy(np.concatenate([base] * (B // 8 + 1))[:B].astype(np.int16)).cuda()
lab = torch.zeros((B, 480 * 640), dtype=torch.int32, device="cuda"); pl = torch.zeros((B, 128, 8), dtype=torch.float64, device="cuda"); npl = torch.zeros(B, dtype=torch.int32, device="cuda")
def run_peac(): pd.segment_dev(d.data_ptr(), lab.data_ptr(), pl.data_ptr(), npl.data_ptr(), B)
# pose
pb_np = pose_batch(B=min(B, 32), seed=7)
def rep(a): return np.concatenate([a] * (B // len(a) + 1))[:B]
dev = {}
pb = PoseBatch(); pb.B, pb.max_points, pb.max_lines, pb.max_planes = B, 1000, 75, 4
for k in ("n_points","n_lines","n_planes","pt_valid","pt_xw","pt_obs","pt_inv_sigma2","ln_valid","ln_obs","ln_xw","pl_meas","pl_valid","pl_world"):
    dev[k] = torch.from_numpy(rep(pb_np[k])).cuda(); setattr(pb, k, dev[k].data_ptr())
dev["Tcw"] = torch.from_numpy(rep(pb_np["Tcw"])).cuda(); pb.Tcw_in = dev["Tcw"].data_ptr()
outs = dict(Tcw_out=torch.zeros((B,16), dtype=torch.float32, device="cuda"), pt_outlier=torch.zeros((B,1000), dtype=torch.uint8, device="cuda"), ln_outlier=torch.zeros((B,75), dtype=torch.uint8, device="cuda"), pl_outlier=torch.zeros((B,4,3), dtype=torch.uint8, device="cuda"), n_inliers=torch.zeros(B, dtype=torch.int32, device="cuda"), lm_iters=torch.zeros(B, dtype=torch.int32, device="cuda"))
for k,v in outs.items(): setattr(pb, k, v.data_ptr())
opt = Optimizer(TUM3, ctx=ctx)
def run_pose(): opt.enqueue_dev(pb, 0, 4, 10)
def run_pose1(): opt.enqueue_dev(pb, 0, 1, 10)
# match
L = lib()
cur = torch.randint(0, 256, (B, 1024, 32), dtype=torch.uint8, device="cuda"); last = torch.randint(0, 256, (B, 1024, 32), dtype=torch.uint8, device="cuda")
n1 = torch.full((B,), 1000, dtype=torch.int32, device="cuda"); has = torch.ones((B,1024), dtype=torch.uint8, device="cuda"); outl = torch.zeros((B,1024), dtype=torch.uint8, device="cuda")
cm = torch.full((B,1024), -1, dtype=torch.int32, device="cuda"); npair = torch.zeros(B, dtype=torch.int32, device="cuda")
def run_match(): check(L.planar_match_orb_points_dev(ctx.h, cur.data_ptr(), n1.data_ptr(), 1024, last.data_ptr(), n1.data_ptr(), 1024, has.data_ptr(), outl.data_ptr(), B, cm.data_ptr(), npair.data_ptr()))
with torch.cuda.stream(stream):
    for name, fn, reps in [("peac", run_peac, 3), ("pose4x10", run_pose, 5), ("pose1x10", run_pose1, 5), ("match_orb_points", run_match, 10)]:
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print(f"{name}: {dt*1e3:.3f} ms per batch of {B} -> {B/dt:.0f} /s")
print("planes", npl[:8].tolist(), "inliers", outs["n_inliers"][:4].tolist(), "iters", outs["lm_iters"][:4].tolist(), "npair", npair[:4].tolist())
