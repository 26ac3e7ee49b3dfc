"""Oracle vs PCL itself - ONLY when a maintainer has produced tests/golden/pcl_golden.npz (tools/pcl_golden/: a C++ dumper that links PCL + the input
generator / packer) on a machine that has PCL 1.7-1.9; neither this container nor the GPU box has it.  Absent file = skipped, and
oracle/planepost_oracle.cpp / oracle/normals_oracle.cpp keep saying PARITY UNPINNED."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol

PATH = os.path.join(os.path.dirname(__file__), "golden", "pcl_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/pcl_golden.npz not generated (needs PCL; tools/pcl_golden/)")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pcl_golden"))


def _cases():
    import make_inputs
    return make_inputs.cases()


def test_voxel_grid_and_refit_and_normals():
    gold = np.load(PATH)
    for name, kind, pts, w, h, th in _cases():
        if kind == 0:
            np.testing.assert_array_equal(ol.voxel_grid(pts), gold[name], err_msg=name)        # same libstdc++ => same summation order
        elif kind == 1:
            # pcl::SACSegmentation alone (no distance gate): compare through the oracle's refit with a plane every point passes
            c = gold[name + "/coef"]
            L = ol.lib()
            import ctypes as C
            out = np.zeros(4, np.float32); inf = np.zeros(12, np.int32)
            L.orc_sac_plane.restype = C.c_int
            ok = L.orc_sac_plane(C.c_void_p(np.ascontiguousarray(pts, np.float32).ctypes.data), len(pts), C.c_double(th), C.c_void_p(out.ctypes.data), C.c_void_p(inf.ctypes.data))
            assert int(gold[name + "/n_inliers"]) == (int(inf[6]) if ok else 0), name
            if len(c) == 4 and ok:
                assert np.abs(out - c).max() < 1e-6, (name, out, c)
        else:
            from planarslam_amd.synth import depth_image
            seed = int(name.split("/")[1])
            nrm, _, = ol.surface_normals(depth_image(seed))
            g = gold[name].reshape(h, w, 3)[1::2, 1::2].reshape(-1, 3)[:len(nrm)]
            assert np.array_equal(np.isnan(nrm), np.isnan(g)), name
            np.testing.assert_array_equal(nrm[~np.isnan(nrm)], g[~np.isnan(nrm)], err_msg=name)
