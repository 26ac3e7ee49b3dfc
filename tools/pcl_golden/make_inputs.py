#!/usr/bin/env python3
"""Inputs for tools/pcl_golden/dump_pcl_golden.cpp and the packer of its output (see that file's header; for a maintainer who has PCL).

    python tools/pcl_golden/make_inputs.py pcl_in.bin
    python tools/pcl_golden/make_inputs.py --pack pcl_out.bin tests/golden/pcl_golden.npz

Cases (all regenerated from seeds, so the .npz stores only PCL's outputs): VOX = voxel grid of seeded clouds and of the member points of synthetic PEAC
planes; SAC = tests/planepost_cases.refit_cases() (clouds that pass / fail / starve the RANSAC); NRM = the 214 x 160 organised cloud Frame::ComputePlanes
builds from a synthetic depth frame."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases():
    import planepost_cases as pc
    from planarslam_amd.synth import depth_image
    out = []
    rng = np.random.default_rng(1)
    out.append(("vox/uniform", 0, np.concatenate([rng.uniform(-2, 2, size=(20000, 3)), rng.normal(scale=0.03, size=(5000, 3)) + [0.5, -0.5, 1.5]]).astype(np.float32), 0, 0, 0.0))
    for c in pc.refit_cases():
        out.append(("vox/" + c["name"], 0, c["pts"], 0, 0, 0.0))
        out.append(("sac/" + c["name"], 1, c["pts"], 0, 0, float(c["th"])))
    for seed in (50, 52):
        d = depth_image(seed)
        z = d[::3, ::3].astype(np.float32) * np.float32(1.0 / 5000.0)
        h, w = z.shape
        n, m = np.meshgrid(np.arange(0, 640, 3, dtype=np.float32), np.arange(0, 480, 3, dtype=np.float32))
        x = (n - np.float32(320.1)) * z / np.float32(535.4); y = (m - np.float32(247.6)) * z / np.float32(539.2)
        out.append((f"nrm/{seed}", 2, np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32), w, h, 0.0))
    return out


def main():
    if sys.argv[1] == "--pack":
        buf = open(sys.argv[2], "rb").read()
        cs = cases()
        off = 4
        assert struct.unpack_from("<i", buf, 0)[0] == len(cs)
        res = {}
        for name, kind, pts, w, h, th in cs:
            k = struct.unpack_from("<i", buf, off)[0]; off += 4
            assert k == kind
            if kind == 1:
                ni, nc = struct.unpack_from("<ii", buf, off); off += 8
                res[name + "/n_inliers"] = np.int32(ni)
                res[name + "/coef"] = np.frombuffer(buf, "<f4", nc, off).copy(); off += 4 * nc
            else:
                m = struct.unpack_from("<i", buf, off)[0]; off += 4
                res[name] = np.frombuffer(buf, "<f4", m * 3, off).reshape(m, 3).copy(); off += 12 * m
        assert off == len(buf)
        np.savez_compressed(sys.argv[3], **res)
        print("wrote", sys.argv[3], len(res), "arrays")
        return
    cs = cases()
    with open(sys.argv[1], "wb") as f:
        f.write(struct.pack("<i", len(cs)))
        for name, kind, pts, w, h, th in cs:
            pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
            f.write(struct.pack("<iiiid", kind, len(pts), w, h, th)); f.write(pts.tobytes())
    print("wrote", sys.argv[1], len(cs), "cases")


if __name__ == "__main__":
    main()
