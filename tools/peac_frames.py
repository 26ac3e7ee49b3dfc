import sys; sys.path.insert(0,'.')
import numpy as np
from planarslam_amd import PlaneDetection
from planarslam_amd._lib import check
from planarslam_amd.synth import depth_image
B=32
if len(sys.argv) > 1 and sys.argv[1] == "se3":            # the frames bench.py times
    import torch
    from planarslam_amd import synth_se3
    from planarslam_amd.synth import TUM3, gray_image
    tex = torch.from_numpy(np.stack([gray_image(1234 + i, 736, 576) for i in range(16)])).cuda()
    d = synth_se3.render_streams(torch, tex, B, 2, TUM3, seed=0)[1][:, 1].cpu().numpy().view(np.uint16)
else:
    d = np.stack([depth_image(4321+i) for i in range(B)])
pd = PlaneDetection(640,480,max_batch=B)
res = pd.run(d)
t = np.zeros((B,48), np.int64)
check(pd.L.planar_peac_read_timing(pd.h, B, t.ctypes.data))
for b in range(B):
    print(b, "ahc ms %.1f" % (t[b,3]/1e5), "nodes", t[b,9], "phases %d evald %d hits %d big %d bigsolves %d" % (t[b,7]>>40, (t[b,7]>>20)&0xfffff, t[b,7]&0xfffff, t[b,10], t[b,11]), "planes", len(res[b][0]))
names = ["q-pop", "q-kill", "issue", "wait-rec", "partner", "chase", "bits", "rank+write", "big-union", "create", "push", "no-merge", "ev-select", "ev-roots", "ev-loads", "ev-eigen", "ev-fold", "ev-publish", "ev-big", "big-compact", "big-bounds"]
if t[:, 16:40].any():
    print("mean Mcyc over the batch: " + "  ".join(f"{n} {t[:, 16 + i].mean() / 1e6:.2f}" for i, n in enumerate(names)), "| sum %.1f" % (t[:, 16:40].sum(1).mean() / 1e6))
    for b in (3, 15, 25):
        print(f"frame {b} Mcyc: " + "  ".join(f"{n} {t[b, 16 + i] / 1e6:.2f}" for i, n in enumerate(names)), "| sum %.1f" % (t[b, 16:40].sum() / 1e6))
