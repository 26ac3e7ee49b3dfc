"""Build sanity of the device code (no GPU needed: hipcc cross-compiles).

hipcc of ROCm 7.2 can materialise a wave-uniform FP64 constant whose low half is zero (+inf, 2^k ...) as `s_mov_b64 sN, <64-bit literal>`.  gfx950 has
no 64-bit literals on SALU moves: the object file carries the low 32 bits and the register ends up 0 - silently (found in round 3: a "best so far =
+inf" that started at 0 in the PEAC clustering kernel; the host emulator cannot see such a thing).  The kernels avoid such constants (DBL_MAX instead of
+inf for uniform values); this test scans the ISA of the kernels that use FP64 sentinels for the pattern."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
BAD = re.compile(r"s_mov_b64\s+s\[[0-9:]+\],\s*(0x[0-9a-fA-F]{9,}|-?[0-9]{11,})")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["peac.hip", "lsd.hip", "pose.hip", "ba.hip", "planepost.hip", "line3d.hip", "manhattan.hip"])   # the FP64 kernels
def test_no_64bit_literal_scalar_moves(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "planarslam_amd", "csrc", src), "-o", str(out)], stderr=subprocess.DEVNULL)
    bad = [ln.strip() for ln in open(out) if BAD.search(ln)]
    assert not bad, f"{src}: scalar moves with a 64-bit literal (not encodable on gfx950, the register would hold the low half only): {bad[:4]}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_pose_opt_kernel_has_no_scratch(tmp_path):
    """pose_opt_kernel at two workgroups per CU (256 VGPRs): the register allocator spills nothing and no array lives in private memory (round 3: 92 VGPRs)."""
    out = tmp_path / "pose.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "planarslam_amd", "csrc", "pose.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    blocks = [b for b in meta.split("  - .agpr_count:") if "pose_opt_kernel" in b]
    assert len(blocks) == 1
    get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blocks[0]).group(1))
    assert get("vgpr_spill_count") == 0 and get("private_segment_fixed_size") == 0, blocks[0]
    assert get("vgpr_count") <= 256
    assert "scratch_" not in text[:text.index("amdhsa.kernels:")]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_peac_refine_fits_four_frames_per_cu(tmp_path):
    """peac_refine runs one workgroup per frame and a batch of 1024 frames puts four on every CU: its LDS must stay within a quarter of a CU's 160 KB (round 5:
    39.8 KB with the work-item arrays of the flood fill), and the flood-fill loop itself must not touch scratch (the spills of the final clustering are outside it)."""
    out = tmp_path / "peac.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "planarslam_amd", "csrc", "peac.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    blocks = [b for b in meta.split("  - .agpr_count:") if "peac_refineE" in b]
    assert len(blocks) == 1
    lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blocks[0]).group(1))
    assert lds <= 160 * 1024 // 4, lds
