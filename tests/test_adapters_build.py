"""CPU: the adapter harnesses of tests/test_adapters_gpu.py (include/planar_adapters.hpp + the shared stand-ins, built with -DSTANDINS_NO_REFERENCE as on the GPU
box) still COMPILE and link against the in-tree library.  They only run on a GPU; a stand-in or harness edit that breaks their build would otherwise show up
at round end only."""
import os

import pytest

import test_adapters_gpu as A

LIB = os.path.join(A.ROOT, "planarslam_amd", "libplanar_hip.so")


@pytest.mark.skipif(not os.path.exists(LIB), reason="libplanar_hip.so not built")
def test_adapter_harnesses_build(tmp_path):
    d = str(tmp_path)
    A._build(d, "adapter_match", [os.path.join(A.ROOT, "oracle", "ref_match_main.cpp")])
    A._build(d, "adapter_pose", [os.path.join(A.SHIM, "adapter_pose_main.cpp")])
    A._build(d, "adapter_ba", [os.path.join(A.SHIM, "adapter_ba_main.cpp")], standins="opt")
