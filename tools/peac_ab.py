"""A/B of the clustering kernels (debug aid): runs PlaneDetection with the exact-heap kernel only (planar_peac_set_variant(.., 1, ..); named "legacy" below for the
round-2 kernel this tool was written against, which no longer exists) and with the product's fast attempt + exact redo on the same depth
images, compares labels / planes with the oracle and - on a mismatch - the node records the two kernels left in the frame workspace (first node whose
moments / plane / N / rid differ = the first merge that went differently), plus per-frame timing of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from planarslam_amd import PlaneDetection
from planarslam_amd._lib import check
from planarslam_amd.synth import depth_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
SEED0 = int(os.environ.get("SEED0", "50"))   # SEED0 != 50: all frames noisy with holes (the bench's frames start at 4321)
depths = np.stack([depth_image(SEED0 + i, noise=(i % 2 == 0) or SEED0 != 50, holes=(i % 3 != 0) or SEED0 != 50) for i in range(B)])


def run(kind):
    pd = PlaneDetection(640, 480, max_batch=B)
    check(pd.L.planar_peac_set_variant(pd.h, 1 if kind == "legacy" else 0, -1))
    try:
        res = pd.run(depths); err = None
    except Exception as e:   # capacity errors etc.
        res = None; err = e
    lay = np.zeros(12, np.int64); check(pd.L.planar_peac_debug_layout(pd.h, lay.ctypes.data))
    t = np.zeros((B, 48), np.int64); check(pd.L.planar_peac_read_timing(pd.h, B, t.ctypes.data))
    def rd(frame, off, n, dt):
        a = np.zeros(n, dt); check(pd.L.planar_peac_debug_read(pd.h, frame, int(off), a.nbytes, a.ctypes.data)); return a
    NB2 = int(lay[2])
    st = []
    for b in range(B):
        hand = rd(b, lay[10], 4 + 128, np.int32)
        st.append(dict(stats=rd(b, lay[3], NB2 * 9, np.float64).reshape(NB2, 9), geo=rd(b, lay[4], NB2 * 7, np.float64).reshape(NB2, 7), N=rd(b, lay[5], NB2, np.int32),
                       rid=rd(b, lay[8], NB2, np.uint16), dsp=rd(b, lay[6], NB2 // 2, np.uint16), dss=rd(b, lay[7], NB2 // 2, np.uint16), hand=hand))
    return res, err, t, st


ra, ea, ta, sa = run("legacy")
rb, eb, tb, sb = run("new")
print("legacy err:", ea, "| new err:", eb)
import oracle_lib as ol
for b in range(B):
    op, olab = ol.peac_run(depths[b])
    oka = ra is not None and np.array_equal(ra[b][1], olab) and ra[b][0].shape == op.shape and np.array_equal(ra[b][0], op)
    okb = rb is not None and np.array_equal(rb[b][1], olab) and rb[b][0].shape == op.shape and np.array_equal(rb[b][0], op)
    na, nb = int(sa[b]["hand"][2]), int(sb[b]["hand"][2])
    msg = f"frame {b}: legacy {'OK' if oka else 'DIFF'} new {'OK' if okb else 'DIFF'} | nodes {na}/{nb} ext {sa[b]['hand'][0]}/{sb[b]['hand'][0]} err {sa[b]['hand'][1]}/{sb[b]['hand'][1]}"
    n = min(na, nb)
    bad = [k for k in ("stats", "geo", "N", "rid") if not np.array_equal(sa[b][k][:n], sb[b][k][:n])]
    if bad or na != nb:
        first = n
        for k in bad:
            d = sa[b][k][:n] != sb[b][k][:n]
            d = d.reshape(n, -1).any(1)
            first = min(first, int(np.argmax(d)))
        msg += f" | node records differ in {bad}, first node {first} (NB = {int(sb[b]['N'].shape[0] // 2)})"
        if first < n:
            msg += f"\n    legacy N {sa[b]['N'][first]} rid {sa[b]['rid'][first]} mse {sa[b]['geo'][first, 6]!r} c {sa[b]['geo'][first, :3]}\n    new    N {sb[b]['N'][first]} rid {sb[b]['rid'][first]} mse {sb[b]['geo'][first, 6]!r} c {sb[b]['geo'][first, :3]}"
    else:
        msg += " | node records identical"
    for k in ("dsp", "dss"):
        if not np.array_equal(sa[b][k], sb[b][k]): msg += f" | {k} differs"
    print(msg)
for name, t in (("legacy", ta), ("new", tb)):
    tot = t[:, 3] / 1e5
    print(f"{name}: clustering kernel per frame ms: " + " ".join("%.1f" % x for x in tot[:16]), "| graph %.2f heap %.2f" % (t[0, 1] / 1e5, (t[0, 2] - t[0, 1]) / 1e5),
          "| phases %d nodes %d hits %d big %d" % (t[0, 7] >> 40, (t[0, 7] >> 20) & 0xfffff, t[0, 7] & 0xfffff, t[0, 10] if name == "new" else 0))
    if name == "new":
        names = ["q-pop", "q-kill/siftup", "issue+dead", "wait-rec", "partner", "chase", "bits+prefix", "rank+write", "big-union", "create", "push", "no-merge",
                 "ev-select", "ev-roots", "ev-loads", "ev-eigen", "ev-fold", "ev-publish", "ev-big"]
        for b in range(min(B, 2)):
            print(f"  frame {b} Mcyc: " + "  ".join(f"{n} {t[b, 16 + i] / 1e6:.2f}" for i, n in enumerate(names)), "| sum %.1f" % (t[b, 16:40].sum() / 1e6))
    else:
        print("  Mcyc " + " ".join("%.1f" % (x / 1e6) for x in t[0, 10:16]))
