#!/usr/bin/env python3
"""bench.py — throughput of the MI355X hot path (BASELINE.json metric: RGB-D frames/sec @640x480).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of `--batch` synthetic 640x480 RGB-D frames that are already
resident in HBM: ORB extraction, LSD+LBD line extraction and PEAC plane segmentation (extract; separate HIP streams,
mirroring the reference's three extraction threads, src/Frame.cc:90-95), SearchByProjection(Cur, Last) and
MatchORBPoints against the previous batch (match) and the 4x10 PoseOptimization protocol on a config-4-shaped
problem per frame (pose-opt), which waits for all three extractors of its own step.
Steps are software-pipelined `--depth` deep (default 2; frames are independent): the line / plane launches of step i
overlap the tails of the previous steps', and PoseOptimization of step i-depth is enqueued behind the point stages of step i.  Every one of the K
timed steps is complete - including its PoseOptimization - before the closing barrier.
Frames are independent, so ranks shard them with no data-path collective ("scaling": "weak": every rank processes
its own `--batch` frames per step).  Rank 0 prints ONE JSON line: whole-job frames/s, per-stage times, the roofline
of the dominant kernel (HIP events on the stream the kernels run on, inside the timed region) and a CPU baseline
(the oracle restatement of the same stages, timed on this box's host cores on a bounded sample).

--workload orb restricts the step to BASELINE config[1] (ORB only).  Stages of the metric that are not built yet are
listed in config.not_yet_in_workload; the number is NOT the complete extract+match+pose-opt rate until that is empty.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
W, H = 640, 480
NPTS, NLINES, NPLANES = 1000, 75, 4          # BASELINE config 4: 1000 point + 150 line-endpoint + 12 plane edges


def orb_algorithmic_bytes(ex, avg_kp):
    """Algorithmic HBM bytes per frame per kernel launch."""
    px = [w * h for w, h in (ex.level_size(l) for l in range(ex.nlevels))]
    return {
        "orb_copy_level0": 2 * px[0],
        "orb_resize": sum(px[l - 1] + px[l] for l in range(1, ex.nlevels)) / max(1, ex.nlevels - 1),
        "orb_fast_cells": sum(px),
        "orb_sort": 0, "orb_octree": 0,
        "orb_blur": 2 * sum(px),
        "orb_describe": avg_kp * (709 + 512 + 60),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step (one workspace set per step in flight: ~25 GB at 1024)")
    ap.add_argument("--workload", choices=["full", "orb"], default="full")
    ap.add_argument("--depth", type=int, default=2, help="software-pipeline depth: PoseOptimization of step i is enqueued during step i+depth")
    ap.add_argument("--prio", default="-1,0,0", help="stream priorities: ORB/match/pose stream, LSD streams, PEAC streams (lower = higher priority)")
    ap.add_argument("--cpu-seconds", type=float, default=18.0, help="budget of the cpu_baseline leg, split over its three variants (0 = skip)")
    ap.add_argument("--latency-reps", type=int, default=15, help="repetitions of the B = 1 latency block (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks (one process per GPU, RCCL rendezvous on 127.0.0.1) and pass the line through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import numpy as np
    import torch

    from planarslam_amd.dist import Ranks, whole_job_fps
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = Ranks(backend="nccl", device=dev)          # RCCL: barrier + max-over-ranks only (frames shard, no data collective)
    rank, world = ranks.rank, ranks.world

    from planarslam_amd import Context, ORBextractor, Optimizer, PlaneDetection
    from planarslam_amd._lib import PoseBatch, check, lib
    from planarslam_amd.synth import TUM3, depth_image, gray_image, pose_batch

    B = args.batch
    full = args.workload == "full"
    prio = [int(x) for x in args.prio.split(',')]
    stream = torch.cuda.Stream(device=local_rank, priority=prio[0])
    ctx = Context(local_rank, stream=stream.cuda_stream)
    # the reference extracts ORB / lines / planes on three threads (src/Frame.cc:90-95): three HIP streams here
    # depth+1 PEAC streams and LSD streams (used round-robin): consecutive launches of the sequential extractors overlap (see step())
    NBUF = args.depth + 1                                     # steps in flight on the line / plane streams
    s_peacs = [torch.cuda.Stream(device=local_rank, priority=prio[2]) for _ in range(NBUF)]
    s_lsds = [torch.cuda.Stream(device=local_rank, priority=prio[1]) for _ in range(NBUF)]
    ctx_peacs = [Context(local_rank, stream=q.cuda_stream) for q in s_peacs]
    ctx_lsds = [Context(local_rank, stream=q.cuda_stream) for q in s_lsds]
    L = lib()

    # ---- inputs resident in HBM (synthetic; distinct per rank, 16 distinct frames tiled over the batch) ----
    nsrc = min(B, 16)
    rep = lambda a: np.concatenate([a] * ((B + len(a) - 1) // len(a)))[:B]
    gray_src = np.stack([gray_image(1234 + 16 * rank + i) for i in range(nsrc)])
    frames = torch.from_numpy(rep(gray_src)).to(dev)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, width=W, height=H, max_batch=B, ctx=ctx)
    d_kps = torch.zeros((B, ex.kp_cap, 7), dtype=torch.float32, device=dev)
    d_desc = [torch.zeros((B, ex.kp_cap, 32), dtype=torch.uint8, device=dev) for _ in range(2)]   # current / previous batch
    d_n = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
    if full:
        depth_src = np.stack([depth_image(4321 + 16 * rank + i) for i in range(nsrc)])
        depth = torch.from_numpy(rep(depth_src).view(np.int16)).to(dev)
        pds = [PlaneDetection(W, H, max_batch=B, ctx=c) for c in ctx_peacs]                     # one workspace per step in flight
        pd = pds[0]
        d_labs = [torch.zeros((B, H * W), dtype=torch.int32, device=dev) for _ in range(NBUF)]
        d_pls = [torch.zeros((B, pd.max_planes, 8), dtype=torch.float64, device=dev) for _ in range(NBUF)]
        d_npls = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NBUF)]
        d_lab, d_pl, d_npl = d_labs[0], d_pls[0], d_npls[0]
        # matcher state
        has_mp = torch.ones((B, ex.kp_cap), dtype=torch.uint8, device=dev)
        outl = torch.zeros((B, ex.kp_cap), dtype=torch.uint8, device=dev)
        cur_match = torch.full((B, ex.kp_cap), -1, dtype=torch.int32, device=dev)
        npair = torch.zeros(B, dtype=torch.int32, device=dev)
        # pose problems (config 4 shape)
        pbn = pose_batch(B=nsrc, n_points=NPTS, n_lines=NLINES, n_planes=NPLANES, seed=7 + 100 * rank)
        keep = {}
        pb = PoseBatch()
        pb.B, pb.max_points, pb.max_lines, pb.max_planes = B, NPTS, NLINES, NPLANES
        for k in ("n_points", "n_lines", "n_planes", "pt_valid", "pt_xw", "pt_obs", "pt_inv_sigma2", "ln_valid", "ln_obs", "ln_xw",
                  "pl_meas", "pl_valid", "pl_world"):
            keep[k] = torch.from_numpy(rep(pbn[k])).to(dev)
            setattr(pb, k, keep[k].data_ptr())
        keep["Tcw"] = torch.from_numpy(rep(pbn["Tcw"])).to(dev)
        pb.Tcw_in = keep["Tcw"].data_ptr()
        outs = dict(Tcw_out=torch.zeros((B, 16), dtype=torch.float32, device=dev), pt_outlier=torch.zeros((B, NPTS), dtype=torch.uint8, device=dev),
                    ln_outlier=torch.zeros((B, NLINES), dtype=torch.uint8, device=dev), pl_outlier=torch.zeros((B, NPLANES, 3), dtype=torch.uint8, device=dev),
                    n_inliers=torch.zeros(B, dtype=torch.int32, device=dev), lm_iters=torch.zeros(B, dtype=torch.int32, device=dev))
        for k, v in outs.items():
            setattr(pb, k, v.data_ptr())
        opt = Optimizer(TUM3, ctx=ctx)
        # line extractor (LSD + LBD), lsdNFeatures = 40
        from planarslam_amd._lib import KEYLINE_DTYPE, FrameView, LastFrameView
        from planarslam_amd.lines import LineSegment
        from planarslam_amd.synth import scale_factors
        lss = [LineSegment(W, H, B, c) for c in ctx_lsds]
        ls = lss[0]
        d_kls = [torch.zeros(B * 40 * KEYLINE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(NBUF)]
        d_ldescs = [torch.zeros((B, 40, 32), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
        d_leqs = [torch.zeros((B, 40, 3), dtype=torch.float64, device=dev) for _ in range(NBUF)]
        d_nls = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NBUF)]
        d_kl, d_ldesc, d_leq, d_nl = d_kls[0], d_ldescs[0], d_leqs[0], d_nls[0]
        # SearchByProjection(Cur, Last): the last frame's map points are the back-projections of the keypoints ORB finds on
        # the same images (identity motion), so every probe has a realistic window of candidates and a true match.
        with torch.cuda.stream(stream):
            ex.extract_dev(frames.data_ptr(), d_kps.data_ptr(), d_desc[0].data_ptr(), d_n[0].data_ptr(), B)
            ex.extract_dev(frames.data_ptr(), d_kps.data_ptr(), d_desc[1].data_ptr(), d_n[1].data_ptr(), B)
        torch.cuda.synchronize()
        h_kps = d_kps.cpu().numpy(); h_n = d_n[0].cpu().numpy()
        rng = np.random.default_rng(11 + rank)
        z = rng.uniform(0.8, 5.0, (B, ex.kp_cap)).astype(np.float32)
        xw = np.stack([(h_kps[..., 0] - TUM3["cx"]) * z / TUM3["fx"], (h_kps[..., 1] - TUM3["cy"]) * z / TUM3["fy"], z], -1).astype(np.float32)
        eye = np.tile(np.eye(4, dtype=np.float32).ravel(), (B, 1))
        pj = dict(u_right=torch.from_numpy((h_kps[..., 0] - np.float32(TUM3["bf"]) / z).astype(np.float32)).to(dev),
                  Tcw=torch.from_numpy(eye).to(dev), usable=torch.ones((B, ex.kp_cap), dtype=torch.uint8, device=dev),
                  xw=torch.from_numpy(xw).to(dev), octave=torch.from_numpy(np.ascontiguousarray(h_kps[..., 5]).view(np.int32).copy()).to(dev),
                  angle=torch.from_numpy(np.ascontiguousarray(h_kps[..., 3])).to(dev),
                  observed=torch.ones((B, ex.kp_cap), dtype=torch.uint8, device=dev),
                  match=torch.full((B, ex.kp_cap), -1, dtype=torch.int32, device=dev), nm=torch.zeros(B, dtype=torch.int32, device=dev))
        fv = FrameView()
        fv.B, fv.stride = B, ex.kp_cap
        fv.keys_un, fv.u_right, fv.Tcw = d_kps.data_ptr(), pj["u_right"].data_ptr(), pj["Tcw"].data_ptr()
        fv.min_x, fv.max_x, fv.min_y, fv.max_y = 0.0, float(W), 0.0, float(H)
        fv.grid_w_inv, fv.grid_h_inv = 64.0 / W, 48.0 / H
        fv.fx, fv.fy, fv.cx, fv.cy, fv.bf, fv.b = TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["bf"], TUM3["bf"] / TUM3["fx"]
        for li, sfv in enumerate(scale_factors()):
            fv.scale_factors[li] = float(sfv)
        lv = LastFrameView()
        lv.stride = ex.kp_cap
        lv.Tcw, lv.usable, lv.xw, lv.octave, lv.angle, lv.mp_observed = (pj["Tcw"].data_ptr(), pj["usable"].data_ptr(), pj["xw"].data_ptr(),
                                                                        pj["octave"].data_ptr(), pj["angle"].data_ptr(), pj["observed"].data_ptr())

    stage_names = ["orb_extract"] + (["search_by_projection", "match_orb_points", "wait_lines_planes", "pose_opt_4x10"] if full else [])
    nst = len(stage_names)
    join_p, join_l, pose_done = ([torch.cuda.Event() for _ in range(NBUF)] for _ in range(3))
    pending = []          # steps whose PoseOptimization is still to be enqueued (software pipeline, FIFO of length --depth)

    def pose(k, evs=None):
        if evs: evs[6].record(stream)
        stream.wait_event(join_p[k]); stream.wait_event(join_l[k])      # PoseOptimization consumes points, lines and planes
        if evs: evs[4].record(stream)
        opt.enqueue_dev(pb, 0, 4, 10)
        if evs: evs[5].record(stream)
        pose_done[k].record(stream)

    def step(i, evs=None, side=None):
        # Five streams.  The reference runs its three extractors as three threads per frame; here the two sequential extractors (PEAC: one
        # workgroup per frame, 149 KB LDS; LSD: one wavefront per frame, 8 KB) of step i run on their own streams beside the ORB stream,
        # and - frames being independent - beside the tail of step i-1's launches (alternating streams, one workspace per step in flight):
        # a PEAC launch ends with its slowest frame (1.5x the mean), and the next launch fills the CUs the finished frames left.
        # PoseOptimization of step i-depth is enqueued after the point stages of step i, when its lines and planes have had `depth` steps to finish.
        cur, prev, k = i & 1, (i & 1) ^ 1, i % NBUF
        sp, sl = s_peacs[k], s_lsds[k]
        if evs: evs[0].record(stream)
        if full:
            sl.wait_event(pose_done[k]); sp.wait_event(pose_done[k])      # their outputs of step i-NBUF have been consumed
            if side: side[2].record(sl)
            check(L.planar_lsd_preprocess_dev(lss[k].h, frames.data_ptr(), B, W, W * H))
            if side: side[0].record(sp)
            pds[k].segment_dev(depth.data_ptr(), d_labs[k].data_ptr(), d_pls[k].data_ptr(), d_npls[k].data_ptr(), B)
            check(L.planar_lsd_detect_dev(lss[k].h, B, 40, d_kls[k].data_ptr(), d_ldescs[k].data_ptr(), d_leqs[k].data_ptr(), d_nls[k].data_ptr()))
            if side: side[1].record(sp); side[3].record(sl)
            join_p[k].record(sp); join_l[k].record(sl)
        ex.extract_dev(frames.data_ptr(), d_kps.data_ptr(), d_desc[cur].data_ptr(), d_n[cur].data_ptr(), B)
        if evs: evs[1].record(stream)
        if full:
            fv.n, fv.desc = d_n[cur].data_ptr(), d_desc[cur].data_ptr()
            lv.n, lv.mp_desc = d_n[prev].data_ptr(), d_desc[prev].data_ptr()
            check(L.planar_search_by_projection_frame_dev(ctx.h, C.byref(fv), C.byref(lv), 15.0, 0, 1, pj["match"].data_ptr(), pj["nm"].data_ptr()))
            if evs: evs[2].record(stream)
            check(L.planar_match_orb_points_dev(ctx.h, d_desc[cur].data_ptr(), d_n[cur].data_ptr(), ex.kp_cap, d_desc[prev].data_ptr(),
                                                d_n[prev].data_ptr(), ex.kp_cap, has_mp.data_ptr(), outl.data_ptr(), B, cur_match.data_ptr(),
                                                npair.data_ptr()))
            if evs: evs[3].record(stream)
            pending.append((k, evs))
            if len(pending) > args.depth:
                pose(*pending.pop(0))

    def drain():
        while pending:
            pose(*pending.pop(0))

    def barrier():
        torch.cuda.synchronize()
        ranks.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            step(i)
        if full: drain()
        standalone = {}
        if full:   # calibration: each sequential extractor alone on the device (not part of the timed region)
            for name, fn in (("peac_segment_alone_ms", lambda: pd.segment_dev(depth.data_ptr(), d_lab.data_ptr(), d_pl.data_ptr(), d_npl.data_ptr(), B)),
                             ("lsd_lbd_alone_ms", lambda: check(L.planar_lsd_extract_dev(ls.h, frames.data_ptr(), B, W, W * H, 40, d_kl.data_ptr(),
                                                                                          d_ldesc.data_ptr(), d_leq.data_ptr(), d_nl.data_ptr())))):
                torch.cuda.synchronize()
                t1 = time.perf_counter(); fn(); torch.cuda.synchronize()
                standalone[name] = round((time.perf_counter() - t1) * 1e3, 3)
        ex.set_profiling(True)
        evsets = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 2)] for _ in range(args.steps)]
        sides = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, evsets[i], sides[i])
        if full: drain()               # the last step's PoseOptimization: all K steps are complete inside the timed region
        barrier()
        elapsed = time.perf_counter() - t0
        prof, calls = ex.get_profile()
        ex.set_profiling(False)
    if full:
        for q in pds:
            q.L.planar_peac_check(q.h, B)

    stage_ms = {n: sum(e[k].elapsed_time(e[k + 1]) for e in evsets) / args.steps for k, n in enumerate(stage_names)}
    if full:   # the pose stage of step i is enqueued during step i+1: its wait starts at event 6, not at the end of step i's matchers
        stage_ms["wait_lines_planes"] = sum(e[6].elapsed_time(e[4]) for e in evsets) / args.steps
    if full:   # the two side streams run concurrently with the ORB stream
        stage_ms["peac_extract(stream 2)"] = sum(e[0].elapsed_time(e[1]) for e in sides) / args.steps
        stage_ms["lsd_lbd_extract(stream 3)"] = sum(e[2].elapsed_time(e[3]) for e in sides) / args.steps
    elapsed = ranks.max_over_ranks(elapsed)
    n_kp = int(d_n[(args.warmup + args.steps - 1) & 1].sum().item())
    if rank != 0:
        ranks.close()
        return

    fps = whole_job_fps(world, B, args.steps, elapsed)
    avg_kp = n_kp / B
    # ---- kernels: per-launch HIP-event times (ORB kernels individually; the other stages are one or two launches each) ----
    kernels = {k: {"ms_per_step": round(v[0] / max(1, calls), 4), "launches_per_step": v[1] // max(1, calls)} for k, v in prof.items()}
    alg = orb_algorithmic_bytes(ex, avg_kp)
    cand = {k: (v[0] / max(1, v[1]), alg.get(k, 0) * B) for k, v in prof.items()}          # (avg launch ms, algorithmic bytes per launch)
    if full:
        avg_planes = float(d_npl.float().mean().item())
        # PEAC: read u16 depth + write int32 labels (1 843 200 B/frame); peac_blocks + peac_segment timed together
        cand["peac_blocks+peac_segment"] = (stage_ms["peac_extract(stream 2)"], 1843200 * B)
        kernels["peac_blocks+peac_segment"] = {"ms_per_step": round(stage_ms["peac_extract(stream 2)"], 4), "launches_per_step": 2,
                                               "note": "concurrent with the LSD stream"}
        # LSD+LBD: read gray + 40 x (32 B descriptor + 68 B KeyLine + 24 B equation) (~311 680 B/frame)
        # the LSD stream's last two small kernels queue behind peac_segment's LDS, so its co-run stage time is not a kernel time:
        # use the standalone time of the 8 launches for the dominance test
        cand["lsd_detect(+7 small kernels)"] = (standalone["lsd_lbd_alone_ms"], (307200 + 40 * 124) * B)
        kernels["lsd_detect(+7 small kernels)"] = {"ms_per_step": round(stage_ms["lsd_lbd_extract(stream 3)"], 4), "launches_per_step": 8,
                                                   "alone_ms": standalone["lsd_lbd_alone_ms"], "note": "concurrent with the PEAC stream"}
        kernels["peac_blocks+peac_segment"]["alone_ms"] = standalone["peac_segment_alone_ms"]
        cand["projection_kernel"] = (stage_ms["search_by_projection"], (1000 * (28 + 4 + 32) + 1000 * (12 + 4 + 4 + 32 + 2)) * B)
        kernels["projection_kernel"] = {"ms_per_step": round(stage_ms["search_by_projection"], 4), "launches_per_step": 1}
        cand["hamming_knn+match_orb_points"] = (stage_ms["match_orb_points"], (64000 + 8000) * B)
        kernels["hamming_knn+match_orb_points"] = {"ms_per_step": round(stage_ms["match_orb_points"], 4), "launches_per_step": 2}
        # pose LM: 65 130 B per frame per LM evaluation; evaluations = LM iterations + trial steps (>= 2 per iteration)
        lm = float(outs["lm_iters"].float().mean().item())
        cand["pose_opt_kernel"] = (stage_ms["pose_opt_4x10"], 65130 * B * 2 * lm)
        kernels["pose_opt_kernel"] = {"ms_per_step": round(stage_ms["pose_opt_4x10"], 4), "launches_per_step": 1, "avg_lm_iterations": round(lm, 2)}
    dom = max(cand, key=lambda k: cand[k][0] * (kernels[k]["launches_per_step"] if k in prof else 1))
    dom_ms, dom_bytes = cand[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    per_frame = 1961064 + (1843200 + 312160 if full else 0) + (72000 + 118000 + 65130 * 2 * 20 if full else 0)
    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r01m_pmc_*.csv: separate --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE runs of this same command at B=1024, KB per launch); scaled to this run's batch.  Raw counter sums: the gfx950
    # "FETCH_SIZE reports half of a wide streaming read" correction is NOT applied because these kernels gather narrow records.
    traffic, traffic_note = None, None
    pmc_csv = os.path.join(ROOT, "profiles", "r01m_pmc_fetch_write_kb_per_launch.csv")
    pmc_key = {"peac_blocks+peac_segment": "planar::peac::peac_segment", "lsd_detect(+7 small kernels)": "planar::lsd::lsd_detect"}.get(dom, "planar::orb::" + dom)
    if os.path.exists(pmc_csv):
        for line in open(pmc_csv).read().splitlines()[1:]:
            k, _, f_kb, w_kb = line.rsplit(",", 3)
            if k == pmc_key:
                traffic = int((float(f_kb) + float(w_kb)) * 1024 * B / 1024)
                traffic_note = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, B=1024): {float(f_kb) / 1024:.0f} MB read + {float(w_kb) / 1024:.0f} MB written per launch"
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note, "avg_launch_ms": round(dom_ms, 4),
                "algorithmic_bytes_per_launch": int(dom_bytes), "pipeline_algorithmic_GBps": round(per_frame * fps / 1e9, 2),
                "note": "latency/occupancy-bound sequential stage (one workgroup per frame); see DESIGN.md" if dom.startswith(("peac", "lsd")) else None,
                "kernels": kernels}

    # ---- CPU baseline: the oracle restatement of the same stages on this box's host cores (tools/cpu_baseline.py) ----
    # value = all host cores (frames shard over processes, as they do over GPUs); next to it the one-thread figure and the reference's own
    # arrangement: three extraction threads per frame (src/Frame.cc:90-95), matching and LM on the calling thread.
    cpu = None
    if args.cpu_seconds > 0 and world == 1:        # rank 0 at N = 1 only
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cpu_baseline as cb
        t_each = args.cpu_seconds / 3.0
        r1 = cb.run(t_each, seed=rank, full=full)
        r3 = cb.run(t_each, seed=rank, threads3=True, full=full) if full else None
        ncores = os.cpu_count()
        fps_all, nworkers, _ = cb.all_cores(t_each, workers=ncores, full=full)
        cpu = {"value": round(fps_all, 2), "unit": "frames/s", "cores": nworkers, "kind": "port",
               "sample": f"{nworkers} worker processes (one per host core), each looping the oracle/ restatements of the same stages over its own synthetic frames for {t_each:.1f} s",
               "one_thread": {"value": round(r1["frames"] / r1["seconds"], 2), "cores": 1, "frames": r1["frames"], "ms_per_frame": cb.summarize(r1)},
               "threads3": None if r3 is None else {"value": round(r3["frames"] / r3["seconds"], 2), "cores": 3, "frames": r3["frames"], "ms_per_frame": cb.summarize(r3),
                                                     "note": "extraction on 3 threads per frame as src/Frame.cc:90-95; matching + LM single-threaded"},
               "host_cores": ncores}

    # ---- single-frame latency (B = 1, host-pointer entry points: H2D + kernels + D2H + sync; the reference is a live B = 1 tracker) ----
    latency = None
    if world == 1 and full and args.latency_reps > 0:
        from planarslam_amd.lines import LineSegment as LS1
        c1 = Context(local_rank)
        ex1 = ORBextractor(1000, 1.2, 8, 20, 7, width=W, height=H, max_batch=1, ctx=c1)
        ls1 = LS1(W, H, 1, c1)
        pd1 = PlaneDetection(W, H, max_batch=1, ctx=c1)
        opt1 = Optimizer(TUM3, ctx=c1)
        g1, dp1 = gray_src[0], depth_src[0]
        pb1 = {k: (v[:1].copy() if isinstance(v, np.ndarray) and v.shape[:1] == (nsrc,) else v) for k, v in pbn.items()}
        kp1, de1 = ex1(g1)
        from planarslam_amd.matcher import ORBmatcher as OM1
        om1 = OM1(ctx=c1)

        def med(fn):
            fn(); fn()
            ts = []
            for _ in range(args.latency_reps):
                t1 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t1) * 1e3)
            return round(float(np.median(ts)), 3)
        latency = {"orb_extract": med(lambda: ex1(g1)), "lsd_lbd_extract": med(lambda: ls1.ExtractLineSegment(g1)),
                   "peac_segment": med(lambda: pd1.run(dp1[None])), "pose_opt_4x10": med(lambda: opt1.PoseOptimization(pb1, 4, 10)),
                   "match_orb_points": med(lambda: om1.MatchORBPoints(de1[None], np.array([len(de1)], np.int32), de1[None], np.array([len(de1)], np.int32),
                                                                       np.ones((1, len(de1)), np.uint8), np.zeros((1, len(de1)), np.uint8))),
                   "reps": args.latency_reps, "note": "median wall ms per call, one frame, host buffers in and out"}
        latency["extract_3_streams_serial_sum"] = round(latency["orb_extract"] + latency["lsd_lbd_extract"] + latency["peac_segment"], 3)

    workload = ("configs[2]+[3]: full extract (ORB + LSD/LBD lines + PEAC planes on separate streams, pose of step i-depth pipelined behind step i) + SearchByProjection + MatchORBPoints + PoseOptimization 4x10 "
                "(1000 pt + 150 line-endpoint + 12 plane edges)"
                if full else "configs[1]: ORB only, 640x480 gray, 8-level pyramid, 1000 keypoints + 256-bit rBRIEF")
    out = {
        "metric": "RGB-D frames/sec (extract+match+pose-opt) @640x480; 1->8-GPU batch scaling",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8/f64" if full else "u8", "data": "synthetic",
        "config": {"workload": workload, "frames_per_gpu_per_step": B, "avg_keypoints_per_frame": round(avg_kp, 1),
                   "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
                   "not_yet_in_workload": ([] if full else
                                           ["LSD/LBD lines", "PEAC planes", "matching", "pose optimisation"]),
                   "parallelism": f"frame-sharded x{world}, no collective"},
        "roofline": roofline, "cpu_baseline": cpu, "latency_b1_ms": latency,
    }
    if full:
        out["config"]["avg_planes_per_frame"] = round(avg_planes, 2)
        out["config"]["avg_lines_per_frame"] = round(float(d_nl.float().mean().item()), 2)
        out["config"]["avg_projection_matches_per_frame"] = round(float(pj["nm"].float().mean().item()), 1)
    print(json.dumps(out))
    ranks.close()


if __name__ == "__main__":
    main()
