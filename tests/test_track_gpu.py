"""The bench's pipelined schedule (planarslam_amd/track.py: five streams, buffer sets reused across steps, the tracking chain of step i - depth
enqueued behind the extraction of step i) checked against the CPU oracle, stage by stage and frame by frame.

Run twice: on the panned canvases of rounds 1-3 and on the SE3-rendered streams bench.py times by default (same renderer, same pipeline depth, the LSD in its top-lines
mode: what the driver's bench line is made of is what is compared here).  B = 64 camera streams, five (pan) / seven (se3) steps; the last two steps (both with a full tracking chain, overlapping the extraction launches of later steps) have
every stage's inputs and outputs cloned on the stream (TrackPipeline.capture_steps).  Each oracle stage is fed the DEVICE's inputs of that stage
(so a 1e-5 difference of an optimiser cannot snowball into different matches downstream) and must reproduce the device's outputs: bit-exact for
extraction, stereo, all matchers, frustum, assembly; 1e-5 on poses / rotation with identical outlier flags for the optimisers and the Manhattan tracker.
This is where a race between streams or a buffer reused too early would show."""
import numpy as np
import pytest

import oracle_lib as ol
from planarslam_amd._lib import KEYLINE_DTYPE
from planarslam_amd.synth import TUM3, pan_offset, scale_factors

pytestmark = pytest.mark.gpu
B, W, H, MARGIN = 64, 640, 480, 48
# "pan": rounds 1-3's input (a window panning over unrelated gray / depth canvases).  "se3": THE INPUT bench.py TIMES BY DEFAULT (--streams se3): cameras moving through
# textured box rooms, gray and depth of a frame ray-cast from one pose (planarslam_amd/synth_se3.py), with bench.py's pipeline depth and its 12-frame loop.
KINDS = {"pan": dict(depth=2, steps=5, tag=""), "se3": dict(depth=3, steps=7, tag="se3/", loop=12)}


def run_pipeline(kind="pan"):
    import torch
    from planarslam_amd.synth import stream_canvases
    from planarslam_amd.track import TrackPipeline, build_map
    cfg = KINDS[kind]
    STEPS, DEPTH = cfg["steps"], cfg["depth"]
    cg, cd = stream_canvases(16, 3, W + 2 * MARGIN, H + 2 * MARGIN, procs=8)
    dev = torch.device("cuda", 0)
    if kind == "se3":
        # as bench.py: the canvases are the rooms' textures, stream s lives in room s mod 16 on its own SE3 path; every stream has its own first frame, hence its own map
        from planarslam_amd import synth_se3
        loop_g, loop_d, _ = synth_se3.render_streams(torch, torch.from_numpy(cg).to(dev), B, cfg["loop"], TUM3, seed=3, W=W, H=H)
        loop_g, loop_d = loop_g.cpu().numpy(), loop_d.cpu().numpy().view(np.uint16)

        def window_np(i):
            fi = synth_se3.frame_index(i, cfg["loop"])
            return np.ascontiguousarray(loop_g[:, fi]), np.ascontiguousarray(loop_d[:, fi])
    else:
        goff = [(24 * (g & 1), 24 * ((g >> 1) & 1)) for g in range(4)]

        def window_np(i):
            ox, oy = pan_offset(i, MARGIN)
            g = np.zeros((B, H, W), np.uint8); d = np.zeros((B, H, W), np.uint16)
            for q in range(4):
                x0, y0 = ox + goff[q][0], oy + goff[q][1]
                g[q * 16:(q + 1) * 16] = cg[:, y0:y0 + H, x0:x0 + W]; d[q * 16:(q + 1) * 16] = cd[:, y0:y0 + H, x0:x0 + W]
            return g, d
    tp = TrackPipeline(B, torch, 0, depth=DEPTH)
    g0, d0 = window_np(0)
    nmap = B if kind == "se3" else 16
    kf, mp, sn = build_map(g0[:nmap], d0[:nmap], TUM3, seed=1)
    rep = lambda a: np.concatenate([a] * (B // nmap))[:B]
    maps = ({k: rep(v) for k, v in kf.items()}, {k: rep(v) for k, v in mp.items()}, {k: rep(v) for k, v in sn.items()})
    tp.set_map(*maps)
    tp.capture_steps = {STEPS - 2, STEPS - 1}
    frames = [torch.zeros((B, H, W), dtype=torch.uint8, device=dev) for _ in range(tp.NB)]
    depths = [torch.zeros((B, H, W), dtype=torch.int16, device=dev) for _ in range(tp.NB)]
    inputs = {}
    with torch.cuda.stream(tp.stream):
        for i in range(STEPS):
            k = i % tp.NB
            tp.stream.wait_event(tp.done[k])
            g, d = window_np(i)
            inputs[i] = (g, d)
            frames[k].copy_(torch.from_numpy(g), non_blocking=False); depths[k].copy_(torch.from_numpy(d.view(np.int16)), non_blocking=False)
            tp.step(i, frames[k], depths[k])
        tp.drain()
    torch.cuda.synchronize()
    tp.check()
    cap = {j: {k: ({kk: vv.cpu().numpy() for kk, vv in v.items()} if isinstance(v, dict) else v.cpu().numpy()) for k, v in c.items()} for j, c in tp.captured.items()}
    return dict(tp=tp, cap=cap, inputs=inputs, maps=maps, S=tp.S, PS=tp.PS, sf=np.asarray(tp.sf, np.float32), lsf=tp.lsf, kind=kind, steps=STEPS, tag=cfg["tag"])


@pytest.fixture(scope="module", params=["pan", "se3"])
def run(request):
    return run_pipeline(request.param)


def oracle_chain_coefficients(c, d):
    """Frame::mvPlaneCoefficients of the oracle's own plane chain for every frame of a captured step"""
    coef_o = np.zeros_like(c["pl_coef"])
    for b in range(B):
        npl = int(c["npl"][b])
        want = ol.plane_clouds(d[b], c["lab"][b].reshape(H, W), c["pls"][b, :npl])
        coef_o[b, :want["n"]] = want["coef"]
    return coef_o


KEYS_P = ("n_points", "n_lines", "n_planes", "pt_valid", "pt_xw", "pt_obs", "pt_inv_sigma2", "ln_valid", "ln_obs", "ln_xw", "pl_meas", "pl_valid", "pl_world")
GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "track_pose_ref.npz")


def _real_optimiser(pb, mode, tag):
    """The captured problems through the REAL Optimizer::PoseOptimization / TranslationOptimization (oracle/_ref/ref_opt: src/Optimizer.cc:550-1275,
    2995-3738 + the vendored g2o, compiled where they lie).  Where the binary did not travel, the committed fixture of its outputs on these very problems
    (tools/gen_golden_track_pose.py; the pipeline is deterministic, the problems are checked by a digest)."""
    import hashlib
    import os
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(pb[k]).tobytes() for k in KEYS_P + ("Tcw",))).hexdigest()
    if os.path.exists(ol.ref_opt_path()):
        r = ol.run_ref_pose(pb, TUM3, mode)
        return dict(r, digest=digest, source="ref_opt")
    z = np.load(GOLDEN)
    assert str(z[tag + "/digest"]) == digest, "the captured problem is not the one the fixture was made from"
    MP, ML, MM = pb["pt_valid"].shape[1], pb["ln_valid"].shape[1], pb["pl_valid"].shape[1]
    nb = len(pb["n_points"])
    return dict(Tcw=z[tag + "/Tcw"], n_inliers=z[tag + "/n_inliers"], pt_outlier=np.unpackbits(z[tag + "/pt_outlier"])[:nb * MP].reshape(nb, MP),
                ln_outlier=np.unpackbits(z[tag + "/ln_outlier"])[:nb * ML].reshape(nb, ML), pl_outlier=np.unpackbits(z[tag + "/pl_outlier"])[:nb * MM * 3].reshape(nb, MM, 3),
                digest=digest, source="fixture")


def _frame_dict(c, Tcw, blocked=None):
    d = dict(n=c["n"], keys_un=c["kps"].view(ol.KP_DTYPE).reshape(B, -1), u_right=c["ur"], desc=c["desc"], Tcw=Tcw, min_x=0.0, max_x=float(W), min_y=0.0, max_y=float(H),
             fx=TUM3["fx"], fy=TUM3["fy"], cx=TUM3["cx"], cy=TUM3["cy"], bf=TUM3["bf"], b=TUM3["bf"] / TUM3["fx"], scale_factors=scale_factors())
    if blocked is not None:
        d["blocked"] = blocked
    return d


@pytest.mark.parametrize("which", [0, 1])
def test_extraction_of_a_pipelined_step(run, which):
    j = run["steps"] - 2 + which
    c, (g, d) = run["cap"][j], run["inputs"][j]
    o = ol.OrbOracle()
    kl = c["kls"].view(KEYLINE_DTYPE).reshape(B, 40)
    for b in range(0, B, 5):                     # every fifth stream: the oracle extractors cost ~60 ms per frame
        kp, de = o.extract(g[b])
        n = int(c["n"][b])
        assert n == len(kp) and c["kps"][b, :n].tobytes() == kp.tobytes() and np.array_equal(c["desc"][b, :n], de)
        rk, rd, re, _, _ = ol.extract_line_segment(g[b], tie_order=0)
        nl = int(c["nl"][b])
        assert nl == len(rk) and kl[b, :nl].tobytes() == rk.tobytes() and np.array_equal(c["ldesc"][b, :nl], rd) and np.array_equal(c["leq"][b, :nl], re)
        planes, labels = ol.peac_run(d[b])
        assert int(c["npl"][b]) == len(planes) and np.array_equal(c["lab"][b].reshape(H, W), labels)
        assert np.array_equal(c["pls"][b, :len(planes)], planes)


@pytest.mark.parametrize("which", [0, 1])
def test_tracking_chain_of_a_pipelined_step(run, which):
    j = run["steps"] - 2 + which
    c, (g, d) = run["cap"][j], run["inputs"][j]
    S, PS, sf, lsf = run["S"], run["PS"], run["sf"], run["lsf"]
    kf, mp, sn = run["maps"]
    keys = c["kps"].view(ol.KP_DTYPE).reshape(B, S)
    # the previous chain's result is this chain's start: pose / rotation hand-over between pipelined steps
    if which == 1:
        prev = run["cap"][j - 1]
        assert np.array_equal(c["pose_in"], prev["pose_out"]) and np.array_equal(c["last_xw"], prev["new_xw"]) and np.array_equal(c["Rcm_in"], prev["Rcm_new"])
    # ---- TrackManhattanFrame ----
    for b in range(0, B, 7):
        nrm, _ = ol.surface_normals(d[b])                  # Frame::ComputePlanes' normals of THIS frame's depth, every bit (NaN pattern included)
        assert np.array_equal(np.isnan(nrm), np.isnan(c["snrm"][b])) and np.array_equal(nrm[~np.isnan(nrm)], c["snrm"][b][~np.isnan(nrm)])
        nlb = int(c["nl"][b])                              # Frame::isLineGood on THIS frame's key lines and depth, with the seed the pipeline used
        l3 = ol.is_line_good(c["kls"].view(KEYLINE_DTYPE).reshape(B, 40)[b, :nlb], d[b], int(np.uint32(c["l3_seeds"][b])))
        gl = l3["good"] > 0
        assert int(c["l3_n_good"][b]) == gl.sum() and np.array_equal(c["l3_lines3d"][b, :nlb], l3["lines3d"]) and np.array_equal(c["l3_depth_line"][b, :nlb], l3["depth_line"])
        assert np.abs(c["l3_packed"][b, :gl.sum()] - l3["direction"][gl]).max(initial=0) <= 1e-9
        w = ol.track_manhattan_frame(c["Rcm_in"][b].reshape(3, 3), nrm, c["l3_packed"][b, :gl.sum()])
        assert np.abs(w["R"].ravel() - c["Rcm_new"][b]).max() <= 1e-5
    # ---- SearchByProjection(Cur, Last) ----
    cur = _frame_dict(c, c["pose_in"])
    last = dict(n=c["last_n"], Tcw=c["pose_in"], usable=c["last_valid"], xw=c["last_xw"], octave=c["last_oct"], angle=c["last_ang"], mp_desc=c["last_desc"],
                mp_observed=np.ones((B, S), np.uint8))
    m, nm = ol.search_by_projection_frame(cur, last, 15.0)
    assert np.array_equal(m, c["pm0"]) and np.array_equal(nm, c["nm"]) and nm.mean() > 300
    # ---- LSDmatcher::SearchByDescriptor, MatchORBPoints ----
    for b in range(B):
        nk, nc = int(kf["n"][b]), int(c["nl"][b])
        wm, wn = ol.lsd_search_by_descriptor(kf["ldesc"][b, :nk], c["ldesc"][b, :nc], np.ones(nk, np.uint8))
        assert np.array_equal(wm, c["lm0"][b, :nc]) and wn == c["nlm0"][b]
    for b in range(0, B, 9):
        n, nlst = int(c["n"][b]), int(c["last_n"][b])
        wm, wn = ol.match_orb_points(c["desc"][b, :n], c["last_desc"][b, :nlst], c["last_valid"][b, :nlst], np.zeros(nlst, np.uint8), np.full(n, -1, np.int32))
        assert np.array_equal(wm, c["cm2"][b, :n]) and wn == c["npair"][b]
    # ---- Frame::ComputePlanes' voxel clouds + refit (mvPlaneCoefficients, mnPlaneNum), then PlaneMatcher on exactly those ----
    assert not c["pl_status"].any()
    for b in range(0, B, 5):
        npl = int(c["npl"][b])
        want = ol.plane_clouds(d[b], c["lab"][b].reshape(H, W), c["pls"][b, :npl])
        k = want["n"]
        assert int(c["pl_n"][b]) == k and np.array_equal(c["pl_src"][b, :k], want["src"]) and np.array_equal(c["pl_off"][b, :k + 1], want["pt_off"])
        assert np.array_equal(c["pl_pts"][b, :want["pt_off"][k]], want["points"])      # PCL's float sums in std::sort's order: every bit
        for q in range(k):                      # the refit equals the oracle's on the (identical) cloud
            P = c["pls"][b, want["src"][q]]
            c0 = np.array([P[1], P[2], P[3], -(P[1] * P[4] + P[2] * P[5] + P[3] * P[6])]).astype(np.float32)
            st, pl, _ = ol.plane_refit(c0, c["pl_pts"][b, want["pt_off"][q]:want["pt_off"][q + 1]], 0.05)
            assert st == 0 and np.abs(pl - c["pl_coef"][b, q]).max() < 1e-6
        assert np.abs(c["pl_coef"][b, :k] - want["coef"]).max(initial=0) <= 1e-6
    coef = c["pl_coef"]
    a, v, p, npm = ol.plane_search_by_coefficients(dict(n=c["pl_n"], coef=coef, Tcw=c["pose_in"]), dict(n=mp["n"], valid=mp["valid"], coef=mp["coef"], npts=mp["npts"], pts=mp["pts"]))
    assert np.array_equal(a, c["plm"][0]) and np.array_equal(p, c["plm"][1]) and np.array_equal(v, c["plm"][2]) and np.array_equal(npm, c["nplm"])
    # ---- assembled translation problem + TranslationOptimization ----
    T = c["pbT"]
    # src/Tracking.cc:1778: the last pose with the Manhattan rotation (Rotation_cm * MF_can^T)^T of THIS frame in its rotation block (oracle pinned to the real statements)
    assert np.array_equal(T["Tcw_in"], ol.manhattan_pose(c["Rcm0"], c["Rcm_new"], c["pose_in"]))
    # the streams pan without rotating: Rotation_cm (fixed by the first tracked frame) and this frame's MF_can stay close, so the Manhattan rotation stays next to
    # the tracked one - within TrackManhattanFrame's own frame-to-frame scatter on these scenes (median ~0.6 deg, a few streams several degrees: measured by
    # tools/manhattan_probe.py), which costs TranslationOptimization inliers on those streams exactly as it would cost the reference
    rot_gap = np.abs(T["Tcw_in"].reshape(B, 4, 4)[:, :3, :3] - c["pose_in"].reshape(B, 4, 4)[:, :3, :3]).max((1, 2))
    # (se3 rooms: the walls ARE a Manhattan frame that turns against the camera, 0.25 deg per frame; the bound is a sanity check of the scene, not a parity statement)
    print(f"[{run['kind']}] Manhattan rotation vs tracked rotation: median {np.median(rot_gap):.4f}, max {rot_gap.max():.4f}; TranslationOptimization inliers {T['n_inliers'].mean():.0f}")
    assert np.median(rot_gap) < 0.03 and rot_gap.max() < (0.25 if run["kind"] == "pan" else 0.5), (float(np.median(rot_gap)), float(rot_gap.max()))
    assert T["n_inliers"].mean() > 300, float(T["n_inliers"].mean())
    for b in range(0, B, 11):
        n = int(c["n"][b])
        ok = c["pm0"][b, :n] >= 0
        ok &= c["last_valid"][b][np.clip(c["pm0"][b, :n], 0, S - 1)] > 0
        assert np.array_equal(T["pt_valid"][b, :n] > 0, ok)
        assert np.array_equal(T["pt_xw"][b, :n][ok], c["last_xw"][b][c["pm0"][b, :n][ok]])
        assert np.array_equal(T["pt_obs"][b, :n, 0], keys["x"][b, :n]) and np.array_equal(T["pt_obs"][b, :n, 2], c["ur"][b, :n])
    pbT = {k: T[k] for k in KEYS_P}
    pbT["Tcw"] = T["Tcw_in"]
    # the REAL TranslationOptimization on the captured problems: 1e-5 on EVERY frame, identical inlier counts and outlier flags
    r = _real_optimiser(pbT, 1, f"{run['tag']}step{which}/T")
    dT = np.abs(r["Tcw"] - T["Tcw_out"]).max(1)
    assert dT.max() <= 1e-5, (r["source"], float(dT.max()), np.nonzero(dT > 1e-5)[0].tolist())
    assert np.array_equal(r["n_inliers"], T["n_inliers"])
    # ... and the restating oracle: on these problems it is within 2e-6 of the real optimiser on every frame too (tools/gen_golden_track_pose.py prints it)
    w = ol.pose_optimize(pbT, TUM3, 1, 4, 10)
    dTo = np.abs(w["Tcw"] - T["Tcw_out"]).max(1)
    assert dTo.max() <= 1e-5 and np.array_equal(w["n_inliers"], T["n_inliers"])
    vT = T["pt_valid"] > 0
    assert np.array_equal(r["pt_outlier"][vT] > 0, w["pt_outlier"][vT] > 0)
    # (the device cleared the flags of the matches it then dropped; compare what the optimiser wrote through the matches that survived)
    dropped = (c["pm0"] >= 0) & (c["pm1"] < 0)
    assert np.array_equal(dropped, (w["pt_outlier"] > 0) & (c["pm0"] >= 0))
    assert np.array_equal(c["kept"], ((c["pm0"] >= 0) & ~dropped).sum(1))
    # ---- TrackLocalMap: isInFrustum, SearchByProjection(map), line search ----
    T1 = T["Tcw_out"]
    fr2 = _frame_dict(c, T1, blocked=(c["pm1"] >= 0).astype(np.uint8))
    old = dict(n=c["old_n"], valid=c["old_valid"], xw=c["old_xw"], normal=c["old_normal"], min_dist=c["old_mind"], max_dist=c["old_maxd"])
    pr = ol.is_in_frustum_points(fr2, old, lsf, len(sf), 0.5)
    inrange = np.arange(S)[None, :] < c["old_n"][:, None]          # rows beyond n[b] are never written (stale values of earlier steps on the device)
    iv = (pr["in_view"] > 0) & inrange
    assert np.array_equal(pr["in_view"][inrange], c["pr"]["in_view"][inrange]) and iv.mean() > 0.05
    for k in ("proj_x", "proj_y", "proj_xr", "level", "view_cos"):
        assert np.array_equal(pr[k][iv], c["pr"][k][iv]), k
    probes = dict({k: np.where(inrange, c["pr"][k], 0).astype(c["pr"][k].dtype) for k in c["pr"]}, n=c["old_n"], desc=c["old_desc"], observed=np.ones((B, S), np.uint8))
    mm, nmm = ol.search_by_projection_map(fr2, probes, 3.0, 0.8)
    assert np.array_equal(mm, c["mm"]) and np.array_equal(nmm, c["nmm"])
    ml = dict(n=kf["n"], valid=np.ones((B, 40), np.uint8), xw6=kf["xw6"], normal=kf["normal"], min_dist=kf["min_dist"], max_dist=kf["max_dist"])
    lp = ol.is_in_frustum_lines(fr2, ml, lsf, 0.5)
    lrange = np.arange(40)[None, :] < kf["n"][:, None]
    il = (lp["in_view"] > 0) & lrange
    assert np.array_equal(lp["in_view"][lrange], c["lpr"]["in_view"][lrange]) and np.array_equal(lp["proj"][il], c["lpr"]["proj"][il])
    lines = dict(n=c["nl"], keylines=c["kls"].view(KEYLINE_DTYPE).reshape(B, 40), ldesc=c["ldesc"], blocked=(c["lm1"] >= 0).astype(np.uint8))
    maplines = dict(n=kf["n"], in_view=lp["in_view"], proj=lp["proj"], level=lp["level"], view_cos=lp["view_cos"], desc=kf["ldesc"], observed=np.ones((B, 40), np.uint8))
    lm2, nlm2 = ol.lsd_search_by_projection(lines, maplines, sf, 3.0, 0.6, match=c["lm1"])
    assert np.array_equal(lm2, c["lm2"]) and np.array_equal(nlm2, c["nlm2"])
    # ---- merged matches, assembled pose problem, PoseOptimization, new map points ----
    want_all = np.where(c["pm1"] >= 0, c["pm1"], np.where(c["mm"] >= 0, c["mm"] + S, -1))
    assert np.array_equal(want_all, c["pm_all"])
    Pp = c["pbP"]
    pbP = {k: Pp[k] for k in KEYS_P}
    pbP["Tcw"] = Pp["Tcw_in"]
    assert np.array_equal(Pp["Tcw_in"], T1)
    # the REAL PoseOptimization on the captured problems: 1e-5 on EVERY frame, identical inlier counts and outlier flags (src/Optimizer.cc:550-1275)
    r = _real_optimiser(pbP, 0, f"{run['tag']}step{which}/P")
    dT = np.abs(r["Tcw"] - Pp["Tcw_out"]).max(1)
    assert dT.max() <= 1e-5, (r["source"], float(dT.max()), np.nonzero(dT > 1e-5)[0].tolist())
    assert np.array_equal(r["n_inliers"], Pp["n_inliers"])
    v = Pp["pt_valid"] > 0
    assert np.array_equal(r["pt_outlier"][v] > 0, Pp["pt_outlier"][v] > 0)          # rows without a map point are never written
    # the restating oracle: within 2e-6 of the real optimiser on every frame of these steps (at a flat minimum the accept / reject decisions of the LM trials are
    # knife edges - rho > 0 on chi2 differences in the 9th digit - and a restatement could part ways with g2o there; it does not on these 128 problems)
    w = ol.pose_optimize(pbP, TUM3, 0, 4, 10)
    dTo = np.abs(w["Tcw"] - Pp["Tcw_out"]).max(1)
    assert dTo.max() <= 1e-5
    assert np.array_equal(w["n_inliers"], Pp["n_inliers"])
    assert np.array_equal(w["pt_outlier"][v], Pp["pt_outlier"][v])
    assert np.array_equal(c["pose_out"], Pp["Tcw_out"]) and Pp["n_inliers"].mean() > 300
    for b in range(0, B, 13):
        n = int(c["n"][b])
        s = ol.stereo_from_rgbd(keys[b, :n], d[b], c["pose_out"][b], TUM3)
        assert np.array_equal(s["xw"], c["new_xw"][b, :n]) and np.array_equal(s["valid"], c["new_valid"][b, :n]) and np.array_equal(s["u_right"], c["new_ur"][b, :n])
        # MapPoint::UpdateNormalAndDepth of the new "map points" (one observation each: this frame)
        ow = ol.keyframe_center(c["pose_out"][b])
        for i in range(0, n, 37):
            if c["new_valid"][b, i]:
                wn, wmn, wmx = ol.update_normal_and_depth(c["new_xw"][b, i], ow[None], ow, keys["octave"][b, i], sf)
                assert np.array_equal(wn, c["new_normal"][b, i]) and wmn == c["new_mind"][b, i] and wmx == c["new_maxd"][b, i]


@pytest.mark.parametrize("which", [0, 1])
def test_pose_with_the_oracles_own_plane_chain(run, which):
    """SURVEY §8 f4, end to end.  PCL (and the oracle) add a voxel's points up as floats in the order std::sort leaves them; since round 4 the device
    reproduces that order (isort.h, libstdc++'s heap-sort fallback included), so its centroids are the oracle's bit for bit and its refitted
    mvPlaneCoefficients the oracle's to 1e-6.  With the ORACLE's own plane chain in place of the device's, PlaneMatcher makes the same associations on every
    frame and the REAL optimisers (oracle/_ref/ref_opt) return the device's pose within 1e-5 on EVERY frame, for both TranslationOptimization and
    PoseOptimization, with identical inlier counts and outlier flags."""
    j = run["steps"] - 2 + which
    c, (g, d) = run["cap"][j], run["inputs"][j]
    kf, mp, sn = run["maps"]
    coef_o = np.zeros_like(c["pl_coef"])
    worst = 0.0
    for b in range(B):
        npl = int(c["npl"][b])
        want = ol.plane_clouds(d[b], c["lab"][b].reshape(H, W), c["pls"][b, :npl])
        k = want["n"]
        assert int(c["pl_n"][b]) == k and np.array_equal(c["pl_src"][b, :k], want["src"])
        assert np.array_equal(c["pl_pts"][b, :want["pt_off"][k]], want["points"]), b
        coef_o[b, :k] = want["coef"]
        worst = max(worst, float(np.abs(c["pl_coef"][b, :k] - want["coef"]).max(initial=0)))
    assert worst <= 1e-6, worst
    # PlaneMatcher::SearchMapByCoefficients on the oracle's coefficients: the device's associations (plane, parallel, vertical), frame by frame
    a, v, p, npm = ol.plane_search_by_coefficients(dict(n=c["pl_n"], coef=coef_o, Tcw=c["pose_in"]), dict(n=mp["n"], valid=mp["valid"], coef=mp["coef"], npts=mp["npts"], pts=mp["pts"]))
    assert np.array_equal(a, c["plm"][0]) and np.array_equal(p, c["plm"][1]) and np.array_equal(v, c["plm"][2]) and np.array_equal(npm, c["nplm"])
    MM = c["pbT"]["pl_meas"].shape[1]
    for name, mode in (("pbT", 1), ("pbP", 0)):
        Q = c[name]
        pb = {k: Q[k] for k in KEYS_P}
        pb["pl_meas"] = np.ascontiguousarray(coef_o[:, :MM]).astype(np.float32)      # Frame::mvPlaneCoefficients of the oracle's chain; associations unchanged (checked above)
        assert np.array_equal((pb["pl_meas"] != 0).any(2), (Q["pl_meas"] != 0).any(2))
        pb["Tcw"] = Q["Tcw_in"]
        r = _real_optimiser(pb, mode, f"{run['tag']}step{which}/{name}_oracle_chain")
        dT = np.abs(r["Tcw"] - Q["Tcw_out"]).max(1)
        assert dT.max() <= 1e-5, (name, r["source"], float(dT.max()), np.nonzero(dT > 1e-5)[0].tolist())
        assert np.array_equal(r["n_inliers"], Q["n_inliers"]), name
        vmask = Q["pt_valid"] > 0
        assert np.array_equal(r["pt_outlier"][vmask] > 0, Q["pt_outlier"][vmask] > 0), name


def test_manhattan_rotation_into_the_translation_pose():
    """planar_manhattan_pose_dev = mRotation_wc = (Rotation_cm * MF_can^T)^T copied into mTcw's rotation block (src/Tracking.cc:251-253, 1778), against the oracle
    that tests/test_oracle_frame_ref.py pins to the real statements (cv::gemm's small-matrix path: float products, float sums)."""
    import torch
    import frame_cases
    from planarslam_amd import Context
    from planarslam_amd._lib import check, lib
    R0, Rn, T = frame_cases.manhattan_pose_case()
    n = len(T)
    dev = torch.device("cuda", 0)
    dRn, dR0, dT = (torch.from_numpy(a).to(dev) for a in (Rn, R0, T))
    out = torch.zeros_like(dT)
    ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    check(lib().planar_manhattan_pose_dev(ctx.h, n, dRn.data_ptr(), dR0.data_ptr(), dT.data_ptr(), out.data_ptr()))
    assert np.array_equal(out.cpu().numpy(), ol.manhattan_pose(R0, Rn, T))


def test_pipeline_with_a_distorting_camera():
    """Camera.k1 != 0 (TUM1's yaml): the pipeline's mvKeysUn = Frame::UndistortKeyPoints(mvKeys), mvuRight / mvDepth read the depth at mvKeys and project from
    mvKeysUn (Frame.cc:600-623), and the chain behind them (matchers, pose) runs on the undistorted keys - poses stay finite and tracked."""
    import torch
    import frame_cases
    from planarslam_amd._lib import KP_DTYPE
    from planarslam_amd.synth import stream_canvases
    from planarslam_amd.track import TrackPipeline, build_map
    K, D = frame_cases.DIST["TUM1"]
    cam = dict(TUM3, fx=K[0], fy=K[1], cx=K[2], cy=K[3])
    cam["bf"] = 40.0
    Bs = 16
    cg, cd = stream_canvases(Bs, 3, W + 2 * MARGIN, H + 2 * MARGIN, procs=8)
    dev = torch.device("cuda", 0)
    tp = TrackPipeline(Bs, torch, 0, depth=1, cam=cam, dist_coef=D)
    win = lambda i: tuple(a[:, pan_offset(i, MARGIN)[1]:pan_offset(i, MARGIN)[1] + H, pan_offset(i, MARGIN)[0]:pan_offset(i, MARGIN)[0] + W].copy() for a in (cg, cd))
    g0, d0 = win(0)
    tp.set_map(*build_map(g0, d0, cam, seed=1))
    n_steps = 5
    tp.capture_steps = {n_steps - 1}
    frames = [torch.zeros((Bs, H, W), dtype=torch.uint8, device=dev) for _ in range(tp.NB)]
    depths = [torch.zeros((Bs, H, W), dtype=torch.int16, device=dev) for _ in range(tp.NB)]
    with torch.cuda.stream(tp.stream):
        for i in range(n_steps):
            k = i % tp.NB
            tp.stream.wait_event(tp.done[k])
            g, d = win(i)
            frames[k].copy_(torch.from_numpy(g)); depths[k].copy_(torch.from_numpy(d.view(np.int16)))
            tp.step(i, frames[k], depths[k])
        tp.drain()
    torch.cuda.synchronize()
    tp.check()
    c = tp.captured[n_steps - 1]
    kps, kpu, n, ur, zd = (c[x].cpu().numpy() for x in ("kps", "kpu", "n", "ur", "zd"))
    moved = 0.0
    for b in range(0, Bs, 3):
        nb = int(n[b])
        k = kps[b, :nb].copy().view(KP_DTYPE).reshape(nb)
        want = ol.undistort_keypoints(k, cam, D)
        assert kpu[b, :nb].tobytes() == want.tobytes()
        st = ol.stereo_from_rgbd(k, d[b], np.eye(4, dtype=np.float32), cam, keys_un=want)
        assert np.array_equal(ur[b, :nb], st["u_right"]) and np.array_equal(zd[b, :nb], st["depth"])
        moved = max(moved, float(np.abs(want["x"] - k["x"]).max()))
    assert moved > 1.0                                                          # the lens model does move key points by pixels
    # Frame::ComputeImageBounds (Frame.cc:575-598): the four undistorted image corners, as the oracle undistorts them
    corners = np.zeros(4, KP_DTYPE); corners["x"] = [0.0, W, 0.0, W]; corners["y"] = [0.0, 0.0, H, H]
    cu = ol.undistort_keypoints(corners, cam, D)
    want_bounds = (float(min(cu["x"][0], cu["x"][2])), float(max(cu["x"][1], cu["x"][3])), float(min(cu["y"][0], cu["y"][1])), float(max(cu["y"][2], cu["y"][3])))
    assert tp.bounds == want_bounds and want_bounds != (0.0, float(W), 0.0, float(H)) and np.isfinite(want_bounds).all()
    assert np.isfinite(c["pose_out"].cpu().numpy()).all()


def test_pipeline_tracks_se3_rendered_streams():
    """Streams rendered from SE3 camera motion in a textured box room (planarslam_amd/synth_se3.py: gray and depth of a frame come from the same pose): the
    pipeline's pose of every stream follows the TRUE camera motion - the map is the stream's first frame at the identity pose, so the estimate is compared with
    inv(Twc[j]) Twc[0].  This is the property the panned canvases of the other tests cannot have (their gray and depth do not belong to one camera)."""
    import torch
    from planarslam_amd import synth_se3
    from planarslam_amd.synth import gray_image
    from planarslam_amd.track import TrackPipeline, build_map
    Bs, K, n_steps = 16, 8, 12
    dev = torch.device("cuda", 0)
    tex = torch.from_numpy(np.stack([gray_image(1234 + i, W + 2 * MARGIN, H + 2 * MARGIN) for i in range(Bs)])).to(dev)
    loop_g, loop_d, Twc = synth_se3.render_streams(torch, tex, Bs, K, TUM3, seed=3)
    tp = TrackPipeline(Bs, torch, 0, depth=1)
    tp.set_map(*build_map(loop_g[:, 0].cpu().numpy(), loop_d[:, 0].cpu().numpy().view(np.uint16), TUM3, seed=1))
    tp.capture_steps = {n_steps - 1}
    frames = [torch.zeros((Bs, H, W), dtype=torch.uint8, device=dev) for _ in range(tp.NB)]
    depths = [torch.zeros((Bs, H, W), dtype=torch.int16, device=dev) for _ in range(tp.NB)]
    with torch.cuda.stream(tp.stream):
        for i in range(n_steps):
            k = i % tp.NB
            tp.stream.wait_event(tp.done[k])
            fi = synth_se3.frame_index(i, K)
            frames[k].copy_(loop_g[:, fi]); depths[k].copy_(loop_d[:, fi])
            tp.step(i, frames[k], depths[k])
        tp.drain()
    torch.cuda.synchronize()
    tp.check()
    c = tp.captured[n_steps - 1]
    T = c["pose_out"].cpu().numpy().reshape(Bs, 4, 4).astype(np.float64)
    fi = synth_se3.frame_index(n_steps - 1, K)
    assert fi != 0
    dt, dr, moved = [], [], []
    for b in range(Bs):
        want = synth_se3.relative_pose(Twc[b], fi, 0)
        dR = T[b, :3, :3] @ want[:3, :3].T
        dr.append(np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))))
        dt.append(np.linalg.norm(T[b, :3, 3] - want[:3, 3]))
        moved.append(np.linalg.norm(want[:3, 3]))
    dt, dr = np.array(dt), np.array(dr)
    print("se3 streams: translation error (m)", np.round(dt, 4), "rotation error (deg)", np.round(dr, 3), "true motion (m)", np.round(moved, 3))
    assert np.median(moved) > 0.02                                          # the cameras did move
    assert np.median(dt) < 0.01 and np.median(dr) < 0.2, (dt, dr)          # and the tracker knows where to (1 cm, 0.2 deg; depth noise is 0.0012 z^2)
    assert (dt < 0.05).mean() >= 0.9 and (dr < 1.0).mean() >= 0.9, (dt, dr)


def test_cu_partition_and_extractor_sets_do_not_change_results():
    """Round 6: the clustering kernel (and, with seq_which = 3, LSD's region growing) on a CU-masked side stream (planar_cu_stream_create / planar_ctx_set_seq_stream: fork
    and join by events) and the extractors' workspaces per extraction in flight instead of per buffer set are scheduling only: every output of a pipelined run - the
    extractors' and the tracker's - is bit-identical to the unpartitioned run's with one extractor set per buffer set (the stochastic part, the 3-D line RANSAC, is
    seeded per stream and step).  The masked run must also have actually created its side stream."""
    import torch
    from planarslam_amd import synth_se3
    from planarslam_amd.synth import gray_image
    from planarslam_amd.track import TrackPipeline, build_map
    Bs, K, n_steps = 24, 6, 9
    dev = torch.device("cuda", 0)
    tex = torch.from_numpy(np.stack([gray_image(4321 + i, W + 2 * MARGIN, H + 2 * MARGIN) for i in range(4)])).to(dev)
    loop_g, loop_d, _ = synth_se3.render_streams(torch, tex, Bs, K, TUM3, seed=11)
    maps = build_map(loop_g[:, 0].cpu().numpy(), loop_d[:, 0].cpu().numpy().view(np.uint16), TUM3, seed=1)
    keys = ("kps", "desc", "n", "kls", "ldesc", "nl", "lab", "pls", "npl", "pl_coef", "pl_pts", "pl_n", "snrm", "l3_lines3d", "pm0", "lm2", "mm", "pose_out", "Rcm_new")

    def run(**kw):
        tp = TrackPipeline(Bs, torch, 0, depth=2, **kw)
        tp.set_map(*maps)
        tp.capture_steps = {n_steps - 1, n_steps - 2}
        frames = [torch.zeros((Bs, H, W), dtype=torch.uint8, device=dev) for _ in range(tp.NB)]
        depths = [torch.zeros((Bs, H, W), dtype=torch.int16, device=dev) for _ in range(tp.NB)]
        with torch.cuda.stream(tp.stream):
            for i in range(n_steps):
                k = i % tp.NB
                tp.stream.wait_event(tp.done[k])
                fi = synth_se3.frame_index(i, K)
                frames[k].copy_(loop_g[:, fi]); depths[k].copy_(loop_d[:, fi])
                tp.step(i, frames[k], depths[k])
            tp.drain()
        torch.cuda.synchronize()
        tp.check()
        out = {(j, name): tp.captured[j][name].cpu().numpy() for j in tp.capture_steps for name in keys}
        info = (tp.NW, len(tp.seq_streams))
        tp.close()
        return out, info

    base, (nw0, ns0) = run(work_sets=4, seq_cus=0)
    part, (nw1, ns1) = run(work_sets=2, seq_cus=64, seq_which=3)
    assert (nw0, ns0) == (4, 0) and (nw1, ns1) == (2, 1)
    for key in base:
        assert np.array_equal(base[key], part[key], equal_nan=True), key


def test_pipeline_tracks_a_1280x720_stream():
    """Round 6: the whole per-frame path at 1280 x 720 (the sizes the plane and line extractors refused before): SE3-rendered HD streams, the tracker's pose against the
    true camera motion, and the extractors' outputs of the last step against the oracle for one stream."""
    import torch
    from planarslam_amd import synth_se3
    from planarslam_amd._lib import KP_DTYPE
    from planarslam_amd.synth import gray_image
    from planarslam_amd.track import TrackPipeline, build_map
    Wh, Hh = 1280, 720
    cam = dict(TUM3, fx=2 * TUM3["fx"], fy=2 * TUM3["fy"], cx=2 * TUM3["cx"], cy=2 * TUM3["cy"], bf=2 * TUM3["bf"])
    Bs, K, n_steps = 4, 6, 8
    dev = torch.device("cuda", 0)
    tex = torch.from_numpy(np.stack([gray_image(77 + i, 736, 576) for i in range(2)])).to(dev)
    loop_g, loop_d, Twc = synth_se3.render_streams(torch, tex, Bs, K, cam, seed=5, W=Wh, H=Hh)
    tp = TrackPipeline(Bs, torch, 0, depth=1, cam=cam, W=Wh, H=Hh)
    tp.set_map(*build_map(loop_g[:, 0].cpu().numpy(), loop_d[:, 0].cpu().numpy().view(np.uint16), cam, seed=1))
    tp.capture_steps = {n_steps - 1}
    frames = [torch.zeros((Bs, Hh, Wh), dtype=torch.uint8, device=dev) for _ in range(tp.NB)]
    depths = [torch.zeros((Bs, Hh, Wh), dtype=torch.int16, device=dev) for _ in range(tp.NB)]
    with torch.cuda.stream(tp.stream):
        for i in range(n_steps):
            k = i % tp.NB
            tp.stream.wait_event(tp.done[k])
            fi = synth_se3.frame_index(i, K)
            frames[k].copy_(loop_g[:, fi]); depths[k].copy_(loop_d[:, fi])
            tp.step(i, frames[k], depths[k])
        tp.drain()
    torch.cuda.synchronize()
    tp.check()
    c = tp.captured[n_steps - 1]
    fi = synth_se3.frame_index(n_steps - 1, K)
    T = c["pose_out"].cpu().numpy().reshape(Bs, 4, 4).astype(np.float64)
    dt = [np.linalg.norm(T[b, :3, 3] - synth_se3.relative_pose(Twc[b], fi, 0)[:3, 3]) for b in range(Bs)]
    # (the reference's search radii are in pixels and the streams move twice as many pixels per frame as at 640x480: some streams lose track with the zero-velocity
    #  motion model - a property of the tracker's parameters, the same in the oracle -, so the pose check asks for the streams that hold on, the stage checks below for parity)
    assert np.isfinite(T).all() and sorted(dt)[len(dt) // 2 - 1] < 0.02, dt
    # stream 0 of the last step against the oracle: key points, key lines, plane labels
    b = 0
    g = loop_g[b, fi].cpu().numpy(); d = loop_d[b, fi].cpu().numpy().view(np.uint16)
    n = int(c["n"][b]); kps = c["kps"][b, :n].cpu().numpy().copy().view(KP_DTYPE).reshape(n)
    okps, odesc = ol.OrbOracle().extract(g)
    assert kps.tobytes() == okps.tobytes() and np.array_equal(c["desc"][b, :n].cpu().numpy(), odesc)
    rk, rd, re, _, nd = ol.extract_line_segment(g, tie_order=0)
    nl = int(c["nl"][b])
    assert nl == len(rk) and np.array_equal(c["ldesc"][b, :nl].cpu().numpy(), rd)
    op, olab = ol.peac_run(d, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    assert np.array_equal(c["lab"][b].cpu().numpy().reshape(Hh, Wh), olab) and int(c["npl"][b]) == len(op) >= 2
