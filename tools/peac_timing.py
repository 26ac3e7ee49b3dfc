import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from planarslam_amd import PlaneDetection
from planarslam_amd._lib import check
from planarslam_amd.synth import depth_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pd = PlaneDetection(640,480,max_batch=B)
d = np.stack([depth_image(4321+i) for i in range(B)])
res = pd.run(d)
t = np.zeros((B,48), np.int64)
check(pd.L.planar_peac_read_timing(pd.h, B, t.ctypes.data))
for b in range(min(B, 8)):
    ph = np.diff(t[b,:7])/100.0  # us (slot 7 carries the cooperative-ahCluster counters)
    print(b, "planes", len(res[b][0]), "us: edges %.0f heap %.0f ahc %.0f seeds %.0f flood %.0f tail %.0f | total %.0f ms"%(*ph, t[b,6]/1e5), "queue", t[b,8], "nodes", t[b,9], "| Mcyc pop %.1f cand %.1f new %.1f union %.1f eval-phases %.1f app %.1f"%tuple(t[b,10:16]/1e6),
          "| phases %d nodes-evaluated %d cache-hits %d" % (t[b,7] >> 40, (t[b,7] >> 20) & 0xfffff, t[b,7] & 0xfffff))
tot = t[:, 6] / 1e5
print("per-frame total ms: min %.1f mean %.1f max %.1f |" % (tot.min(), tot.mean(), tot.max()), " ".join("%.0f" % x for x in tot[:32]))
