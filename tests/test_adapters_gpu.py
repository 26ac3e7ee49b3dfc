"""The C++ drop-in boundary, executed: include/planar_adapters.hpp compiled into small executables ON THE GPU BOX and run against
the same inputs the real reference processed.

  adapter_match   = oracle/ref_match_main.cpp (the harness that drives the REAL src/ORBmatcher.cc / LSDmatcher.cpp / PlaneMatcher.cpp in
                    oracle/_ref/ref_match) rebuilt with tests/adapter_shim/*.h in place of the reference headers, so ORBmatcher::SearchByProjection,
                    SearchByBoW, MatchORBPoints, Fuse, LSDmatcher::SearchByProjection / SearchByDescriptor / Fuse and PlaneMatcher::SearchMapByCoefficients
                    are the adapter definitions (gather Frame fields -> C ABI -> scatter MapPoint* back).  Expected: tests/golden/guided_ref.npz
                    (outputs of the real reference on the same seeds): every index identical.
  adapter_pose    = Optimizer::PoseOptimization / TranslationOptimization(Frame*) adapters on stand-in Frames; expected tests/golden/opt_ref.npz
                    (real src/Optimizer.cc + g2o): pose within 1e-5, flags and return values identical.
  adapter_ba      = Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*) adapter (graph collection from covisibility / observation maps -> planar_local_ba ->
                    erase lists + write-back); expected tests/golden/opt_ref.npz (the real LocalBundleAdjustment): poses 1e-5, erase flags identical.
  adapter_extract = ORBextractor / LineSegment / PlaneDetection used as src/Frame.cc:90-97 does (three threads per frame, by-value copies,
                    LineSegment through a null pointer); expected: the oracle's output on the same images.
"""
import os
import subprocess

import numpy as np
import pytest

import opt_cases as cases
import oracle_lib as O
from planarslam_amd import synth
from planarslam_amd.synth import TUM3
from test_oracle_opt_ref import golden_pose

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "adapter_shim")


def _build(tmp, name, sources, standins="match"):
    lib = os.path.join(ROOT, "planarslam_amd", "libplanar_hip.so")
    exe = os.path.join(tmp, name)
    cmd = ["g++", "-O1", "-std=c++14", "-w", "-pthread", "-DCVSHIM_ALGEBRA", "-I" + SHIM, "-I" + os.path.join(ROOT, "include")]
    if standins == "opt":        # the optimiser stand-ins use the mini-Eigen (must precede oracle/shim, whose <Eigen/Core> is the 3-type stub)
        cmd += ["-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROOT, "oracle", "shim", "minieigen")]
    cmd += ["-I" + os.path.join(ROOT, "oracle", "shim")]
    if standins:
        cmd += ["-DSTANDINS_NO_REFERENCE", "-include", os.path.join(ROOT, "oracle", "shim", standins + "_standins.hpp")]
    cmd += ["-o", exe] + sources + [os.path.join(ROOT, "oracle", "cvprim.cpp"), lib, "-Wl,-rpath," + os.path.dirname(lib), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


@pytest.fixture(scope="module")
def bins(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("adapters"))
    return dict(match=_build(d, "adapter_match", [os.path.join(ROOT, "oracle", "ref_match_main.cpp")]),
                pose=_build(d, "adapter_pose", [os.path.join(SHIM, "adapter_pose_main.cpp")]),
                extract=_build(d, "adapter_extract", [os.path.join(SHIM, "adapter_extract_main.cpp")], standins=None),
                ba=_build(d, "adapter_ba", [os.path.join(SHIM, "adapter_ba_main.cpp")], standins="opt"), dir=d)


@pytest.fixture()
def as_reference(bins):
    """Route oracle_lib's ref_match / ref_opt runners (file formats shared with the real-reference binaries) to the adapter executables."""
    O.BINARY_OVERRIDE.update(ref_match=bins["match"], ref_opt=bins["pose"])      # ref_opt: `pose` mode here, the `ba` test swaps in its own binary
    yield
    O.BINARY_OVERRIDE.clear()


def test_matcher_adapters_equal_real_reference_fixtures(as_reference, golden_dir):
    g = np.load(os.path.join(golden_dir, "guided_ref.npz"))
    seed = int(g["seed"])
    fr = synth.guided_frame(B=2, N=800, seed=seed)
    cur, last = synth.guided_last_frame(fr, seed=seed + 1, dup=0.3)
    for b in range(2):
        m, n = O.ref_search_by_projection_frame(cur, last, b, 15.0)
        np.testing.assert_array_equal(m, g["proj_frame_match"][b, :len(m)]); assert n == g["proj_frame_n"][b]
    fr2, pr = synth.guided_map_probes(fr, seed=seed + 2, n_probes=2000)
    for b in range(2):
        m, n = O.ref_search_by_projection_map(fr2, pr, b, 3.0, 0.8)
        np.testing.assert_array_equal(m, g["proj_map_match"][b, :len(m)]); assert n == g["proj_map_n"][b]
    kf, f = synth.guided_bow(B=2, N=800, seed=seed + 3)
    for b in range(2):
        m, n = O.ref_search_by_bow(kf, f, b, 0.7)
        np.testing.assert_array_equal(m, g["bow_match"][b, :len(m)]); assert n == g["bow_n"][b]
    frp, mp = synth.guided_planes(B=4, seed=seed + 4)
    for b in range(4):
        a, v, p, n = O.ref_plane_search(frp, mp, b)
        for k, arr in enumerate((a, v, p)):
            np.testing.assert_array_equal(arr, g["plane_avp"][k, b, :len(arr)])
        assert n == g["plane_n"][b]
    lines, ml = synth.guided_lines(B=3, n_lines=150, n_ml=400, seed=seed + 5)
    for b in range(3):
        m, n = O.ref_lsd_search_by_projection(lines, ml, b, synth.scale_factors(), 3.0, 0.6)
        np.testing.assert_array_equal(m, g["lsd_proj_match"][b, :len(m)]); assert n == g["lsd_proj_n"][b]


def test_fuse_adapters_equal_real_reference_fixtures(as_reference, golden_dir):
    """ORBmatcher::Fuse / LSDmatcher::Fuse adapters (search on the device, the reference's map edits on the stand-in objects) against the pairings the REAL
    functions made on the same inputs (tests/golden/fuse_points_ref.npz).  The harness reads a pairing off the edit the adapter makes (AddObservation / Replace),
    so the key frame's occupied slots hold no bad map points here (those make no edit); the search result does not depend on the slots' state."""
    import fuse_cases as fc
    g = np.load(os.path.join(golden_dir, "fuse_points_ref.npz"))
    kf, mp = fc.fuse_case(seed=131)
    lsf, nlev = fc.scale()
    state, kobs = fc.kf_map_points(kf)
    state = np.minimum(state, 1)
    for th in (3.0, 1.5):
        for b in range(kf["keys_un"].shape[0]):
            idx, nf = O.ref_fuse(kf, mp, b, th, lsf, nlev, kf_state=state[b], kf_obs=kobs[b])
            np.testing.assert_array_equal(idx, g[f"fuse_idx_th{th}"][b, :len(idx)]); assert nf == g[f"n_fused_th{th}"][b]
    kfl, lines, ml = fc.fuse_lines_case()
    rng = np.random.default_rng(5)
    lstate = np.minimum(rng.choice([0, 1, 2], lines["keylines"].shape, p=[0.5, 0.4, 0.1]), 1).astype(np.uint8); lobs = rng.integers(1, 9, lstate.shape).astype(np.int32)
    for th in (3.0, 6.0):
        for b in range(kfl["B"]):
            idx, nf = O.ref_lsd_fuse(kfl, lines, ml, b, th, lsf, nlev, kf_state=lstate[b], kf_obs=lobs[b])
            np.testing.assert_array_equal(idx, g[f"lsd_fuse_idx_th{th}"][b, :len(idx)]); assert nf == g[f"lsd_n_fused_th{th}"][b]


def test_descriptor_matcher_adapters_equal_oracle(as_reference):
    """MatchORBPoints and LSDmatcher::SearchByDescriptor (cv::BFMatcher based; pinned to the real reference in tests/test_oracle_guided_ref.py)."""
    rng = np.random.default_rng(5)
    last = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    cur = last[rng.permutation(500)[:400]].copy()
    cur[rng.random(cur.shape) < 0.02] ^= 0x10
    has = (rng.random(500) < 0.7).astype(np.uint8); outl = (rng.random(500) < 0.1).astype(np.uint8)
    m, n = O.ref_match_orb_points(cur, last, has, outl)
    want, wn = O.match_orb_points(cur, last, has, outl, np.full(400, -1, np.int32))
    np.testing.assert_array_equal(m, want); assert n == wn and n > 100
    kfd = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    cd = kfd[rng.permutation(60)[:45]].copy()
    cd[rng.random(cd.shape) < 0.03] ^= 0x04
    hasl = (rng.random(60) < 0.8).astype(np.uint8)
    m, n = O.ref_lsd_search_by_descriptor(kfd, cd, hasl)
    wm, wn = O.lsd_search_by_descriptor(kfd, cd, hasl)
    np.testing.assert_array_equal(m, wm); assert n == wn and n > 10


@pytest.mark.parametrize("name", ["c4_b8", "ragged", "few", "planes_partly_missing"])
def test_pose_adapters_equal_real_reference_fixtures(as_reference, golden_dir, name):
    golden = np.load(os.path.join(golden_dir, "opt_ref.npz"))
    build, modes = cases.POSE_CASES[name]
    b = build()
    for mode in modes:
        got = O.run_ref_pose(b, TUM3, mode)
        want = golden_pose(golden, name, mode, b)
        assert np.abs(got["Tcw"] - want["Tcw"]).max() <= 1e-5
        assert np.array_equal(got["n_inliers"], want["n_inliers"])
        for k in ("pt_outlier", "ln_outlier", "pl_outlier"):
            assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("name", list(cases.BA_CASES))
def test_local_ba_adapter_equals_real_reference_fixtures(bins, golden_dir, name):
    """Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*) as defined by the adapter header, on KeyFrame / MapPoint / MapLine / MapPlane stand-ins built by the same
    harness that drives the real src/Optimizer.cc (oracle/ref_opt_harness.hpp): poses, landmarks and erased observations vs tests/golden/opt_ref.npz."""
    from test_oracle_opt_ref import check_ba, golden_ba
    golden = np.load(os.path.join(golden_dir, "opt_ref.npz"))
    build, cur = cases.BA_CASES[name]
    pr = build()
    O.BINARY_OVERRIDE["ref_opt"] = bins["ba"]
    try:
        got = O.run_ref_local_ba(pr, TUM3, cur)
    finally:
        O.BINARY_OVERRIDE.clear()
    check_ba(golden_ba(golden, name, pr), got, pr)


def test_extractor_adapters_three_threads_per_frame(bins):
    from planarslam_amd._lib import KEYLINE_DTYPE, KP_DTYPE
    W, H = 640, 480
    frames = [(synth.gray_image(11 + i), synth.depth_image(31 + i)) for i in range(3)]
    fin, fout = os.path.join(bins["dir"], "ex_in.bin"), os.path.join(bins["dir"], "ex_out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([W, H, len(frames)], np.int32).tobytes())
        for g, d in frames:
            f.write(np.ascontiguousarray(g, np.uint8).tobytes()); f.write(np.ascontiguousarray(d, np.uint16).tobytes())
    subprocess.check_call([bins["extract"], fin, fout])
    buf = open(fout, "rb").read()
    off = 0
    orb = O.OrbOracle()
    for g, d in frames:
        n = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
        kps = np.frombuffer(buf, KP_DTYPE, n, off); off += n * KP_DTYPE.itemsize
        desc = np.frombuffer(buf, np.uint8, n * 32, off).reshape(n, 32); off += n * 32
        rk, rd = orb.extract(g)
        assert n == len(rk) and n > 500
        for fld in rk.dtype.names:
            np.testing.assert_array_equal(kps[fld], rk[fld], err_msg=fld)
        np.testing.assert_array_equal(desc, rd)
        n = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
        kl = np.frombuffer(buf, KEYLINE_DTYPE, n, off); off += n * KEYLINE_DTYPE.itemsize
        ld = np.frombuffer(buf, np.uint8, n * 32, off).reshape(n, 32); off += n * 32
        eq = np.frombuffer(buf, "<f8", n * 3, off).reshape(n, 3); off += n * 24
        wk, wd, we, _, _ = O.extract_line_segment(g, tie_order=0)
        assert n == len(wk) and n > 5
        for fld in wk.dtype.names:
            np.testing.assert_array_equal(kl[fld], wk[fld], err_msg=fld)
        np.testing.assert_array_equal(ld, wd); np.testing.assert_array_equal(eq, we)
        n = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
        planes, labels = O.peac_run(d)
        assert n == len(planes) and n >= 1
        for i in range(n):
            npix = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
            nc = np.frombuffer(buf, "<f8", 6, off); off += 48
            assert npix == int((labels == i).sum())
            np.testing.assert_array_equal(nc, planes[i, 1:7])
        lab = np.frombuffer(buf, "<i4", W * H, off).reshape(H, W); off += 4 * W * H
        np.testing.assert_array_equal(lab, labels)
        # the Frame::ComputePlanes loop (voxel clouds + refit): same planes kept, the clouds within the float-summation error, the refit the oracle's on them
        want = O.plane_clouds(d, labels, planes)
        k = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
        assert k == want["n"]
        for q in range(k):
            coef = np.frombuffer(buf, "<f4", 4, off); off += 16
            npts = int(np.frombuffer(buf, "<i4", 1, off)[0]); off += 4
            cloud = np.frombuffer(buf, "<f4", npts * 3, off).reshape(npts, 3); off += npts * 12
            assert npts == want["pt_off"][q + 1] - want["pt_off"][q] and np.abs(cloud - want["points"][want["pt_off"][q]:want["pt_off"][q + 1]]).max() < 2e-5
            P = planes[want["src"][q]]
            c0 = np.array([P[1], P[2], P[3], -(P[1] * P[4] + P[2] * P[5] + P[3] * P[6])]).astype(np.float32)
            st, pl, _ = O.plane_refit(c0, cloud, 0.05)
            assert st == 0 and np.abs(pl - coef).max() < 1e-6
    assert off == len(buf)
