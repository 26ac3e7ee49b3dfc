#!/usr/bin/env python3
"""bench.py — throughput of the MI355X hot path (BASELINE.json metric: RGB-D frames/sec, extract + match + pose-opt @640x480).

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N ranks, or is launched by torch.distributed.run)

A "step" = one frame of every one of `--batch` independent camera streams through the reference's per-frame tracking sequence
(planarslam_amd/track.py; src/Tracking.cc):

    extract   ORBextractor + LineSegment::ExtractLineSegment + PlaneDetection on three HIP streams (src/Frame.cc:90-95), ComputeStereoFromRGBD
    track     TrackManhattanFrame - SearchByProjection(Cur, Last) - LSDmatcher::SearchByDescriptor - MatchORBPoints - PlaneMatcher -
              TranslationOptimization 4x10 (src/Tracking.cc:1739-1790) - isInFrustum + SearchByProjection(map points) +
              LSDmatcher::SearchByProjection - PoseOptimization 4x10 (:1954-2040) - UnprojectStereo
    The pose problems are assembled on the device from the matchers' outputs (planar_pose_assemble); nothing is a canned array.

Inputs: per rank 256 distinct synthetic RGB-D canvases (736 x 576), resident in HBM; stream s looks at canvas s mod 256 through a 640x480 window
that pans a few pixels per step (consecutive frames of a stream overlap like video, every step shows every stream a NEW frame).  The window
copy (a strided device-to-device copy) is inside the timed region.  Steps are software-pipelined `--depth` deep: the tracking chain of step
i - depth runs behind the extraction launches of step i.  Every one of the K timed steps is complete before the closing barrier.
Frames are independent, so ranks shard them with no data-path collective ("scaling": "weak").  Rank 0 prints ONE JSON line: whole-job frames/s,
per-stage times, the roofline of the dominant stage (HIP events on the stream it runs on, inside the timed region), a CPU baseline (the
oracle restatement of the same stages on all host cores, one thread and the reference's three extraction threads), the single-frame latency and the
PCIe-inclusive rate.  --workload orb restricts the step to BASELINE config[1] (ORB only)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The step runs on ten HIP streams (points, tracking chain, four line and four plane streams).  The ROCm runtime maps streams onto 4 hardware queues by default and
# launches that share a queue serialise; with 8 queues the step takes 105.7 instead of 115.0 ms (12 / 16 measure the same).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ENV_OVERRIDES = []
PARITY_SAMPLE = True    # after the timed region: a few frames of the last step against the oracle (tools/stage_skip.py turns it off: nothing to compare there)
W, H = 640, 480
MARGIN = 48
NCANVAS = 256
SEQ_CUS = 128           # default CU partition: PEAC's clustering kernel (one wavefront per frame, 352 registers, 38 KB of LDS) confined to half of the CUs by a CU-masked
SEQ_WHICH = 1           # side stream, so that the other half always has room for the wide kernels' workgroups (round-6 sweep, DESIGN.md §6: +2-4 % at 2048 frames per step)


def orb_algorithmic_bytes(ex, avg_kp):
    """Algorithmic HBM bytes per frame per kernel launch."""
    px = [w * h for w, h in (ex.level_size(l) for l in range(ex.nlevels))]
    return {"orb_copy_level0": 2 * px[0], "orb_resize": sum(px[l - 1] + px[l] for l in range(1, ex.nlevels)) / max(1, ex.nlevels - 1),
            "orb_fast_cells": sum(px), "orb_sort": 0, "orb_octree": 0, "orb_blur": 2 * sum(px), "orb_describe": avg_kp * (709 + 512 + 60)}


def pick_device(args, torch):
    """LOCAL_RANK's GPU.  RCCL needs one device per rank; with --backend gloo the ranks wrap around the visible devices."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if args.backend == "nccl":
            raise SystemExit(f"rank {local_rank} has no GPU ({ndev} visible): RCCL needs one device per rank (use --backend gloo to share devices)")
        local_rank %= ndev
    return local_rank


def parity_sample(tp, frames, depths, k, B, cam, streams=None):
    """AFTER the timed region: a few frames of the LAST timed step against the CPU oracle (the checker; never the thing measured) - the buffers of set k still hold that
    step's inputs and every stage's outputs: ORB key points / descriptors, key lines / LBD / line equations (the LSD's top-lines mode), PEAC labels and planes, the
    planes' voxel clouds (PCL's float sums in std::sort's order) bit for bit; the refitted coefficients to 1e-6; TranslationOptimization and PoseOptimization of the
    last tracking chain within 1e-5 of the oracle's LM on the problems the device assembled, inlier counts equal.  -> {"frames": n, "ok": bool, ...}"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol                                      # test infrastructure: used here only as the checker, outside the timed region
    from planarslam_amd._lib import KEYLINE_DTYPE
    streams = streams or sorted({0, B // 3, (2 * B) // 3, B - 1})
    H_, W_ = frames[k].shape[1:]
    o = ol.OrbOracle()
    fails, worst_pose, worst_coef = [], 0.0, 0.0
    KEYS = ("n_points", "n_lines", "n_planes", "pt_valid", "pt_xw", "pt_obs", "pt_inv_sigma2", "ln_valid", "ln_obs", "ln_xw", "pl_meas", "pl_valid", "pl_world")
    for b in streams:
        g = frames[k][b].cpu().numpy(); d = depths[k][b].cpu().numpy().view(np.uint16)
        n = int(tp.n[k][b]); kp, de = o.extract(g)
        if not (n == len(kp) and tp.kps[k][b, :n].cpu().numpy().tobytes() == kp.tobytes() and np.array_equal(tp.desc[k][b, :n].cpu().numpy(), de)):
            fails.append(f"stream {b}: ORB")
        rk, rd, re, _, _ = ol.extract_line_segment(g, tie_order=0)
        nl = int(tp.nl[k][b]); kl = tp.kls[k].cpu().numpy().view(KEYLINE_DTYPE).reshape(B, 40)
        if not (nl == len(rk) and kl[b, :nl].tobytes() == rk.tobytes() and np.array_equal(tp.ldesc[k][b, :nl].cpu().numpy(), rd) and np.array_equal(tp.leq[k][b, :nl].cpu().numpy(), re)):
            fails.append(f"stream {b}: LSD/LBD")
        planes, labels = ol.peac_run(d)
        npl = int(tp.npl[k][b]); lab = tp.lab[k][b].cpu().numpy().reshape(H_, W_)
        if not (npl == len(planes) and np.array_equal(lab, labels) and np.array_equal(tp.pls[k][b, :npl].cpu().numpy(), planes)):
            fails.append(f"stream {b}: PEAC")
        else:
            want = ol.plane_clouds(d, labels, planes)
            pc = tp.pc[k]; q = want["n"]
            off = pc["off"][b].cpu().numpy()
            if not (int(pc["n"][b]) == q and np.array_equal(pc["src"][b, :q].cpu().numpy(), want["src"]) and np.array_equal(off[:q + 1], want["pt_off"])
                    and np.array_equal(pc["pts"][b, :off[q]].cpu().numpy(), want["points"])):
                fails.append(f"stream {b}: plane clouds")
            else:
                worst_coef = max(worst_coef, float(np.abs(pc["coef"][b, :q].cpu().numpy() - want["coef"]).max(initial=0)))
        for which, mode in ((0, 1), (1, 0)):                     # the last chain's TranslationOptimization / PoseOptimization problems, as the device assembled them
            A = tp.pb_arrays[which]
            pb = {kk: A[kk][b:b + 1].cpu().numpy() for kk in KEYS}
            pb["Tcw"] = A["Tcw_in"][b:b + 1].cpu().numpy()
            w = ol.pose_optimize(pb, cam, mode, 4, 10)
            dT = float(np.abs(w["Tcw"] - A["Tcw_out"][b:b + 1].cpu().numpy()).max())
            worst_pose = max(worst_pose, dT)
            if not (dT <= 1e-5 and int(w["n_inliers"][0]) == int(A["n_inliers"][b])):
                fails.append(f"stream {b}: {'TranslationOptimization' if mode else 'PoseOptimization'} ({dT:.2e})")
    if worst_coef > 1e-6:
        fails.append(f"plane refit {worst_coef:.2e}")
    for f in fails:
        print(f"bench.py: PARITY SAMPLE MISMATCH: {f}", file=sys.stderr)
    return {"frames": len(streams), "streams": streams, "ok": not fails, "mismatches": fails, "max_pose_diff_vs_oracle": worst_pose, "max_refit_coef_diff": worst_coef,
            "checked": "last timed step: ORB keypoints+descriptors, key lines+LBD+equations, PEAC labels+planes, voxel centroids bit-exact; refit <= 1e-6; both optimisers <= 1e-5 "
                       "and inlier counts (oracle = checker, run after the timed region; the full comparison is tests/test_track_gpu.py on these se3 streams)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="camera streams per GPU = frames per GPU per step (0: 2048 for the full workload - two full rounds of the clustering "
                                                         "kernel's 1024 frames; it fits since the extractors' workspaces exist per extraction in flight, not per buffer set -, "
                                                         "1024 for --workload orb, 256 for --workload pose)")
    ap.add_argument("--workload", choices=["full", "orb", "ba", "pose"], default="full",
                    help="full: the per-frame path (BASELINE metric); orb: config[1] only; ba: config[4], ONE local bundle adjustment partitioned over the ranks")
    ap.add_argument("--depth", type=int, default=3, help="software-pipeline depth: the tracking chain of step i runs during step i + depth")
    ap.add_argument("--prio", default="-1,0,0", help="stream priorities: point stream, LSD streams, PEAC streams[, tracking stream] (lower = higher priority)")
    ap.add_argument("--work-sets", type=int, default=0, help="extractor sets (line / plane stream + the extractors' workspaces): 0 = one per extraction in flight (= --depth)")
    ap.add_argument("--seq-cus", type=int, default=SEQ_CUS, help="compute units the one-wavefront-per-frame kernels (PEAC clustering, LSD region growing) are confined to by a "
                                                                "CU-masked side stream (planar_ctx_set_seq_stream); 0 = no partition")
    ap.add_argument("--seq-which", type=int, default=SEQ_WHICH, help="which of them: 1 PEAC clustering, 2 LSD region growing (on the same stream when shared), 3 both, 4 LSD region growing on a masked stream per line context, 5 = 1 + 4")
    ap.add_argument("--seq-per-ctx", action="store_true", help="one masked stream per context instead of one shared by all sets")
    ap.add_argument("--cpu-seconds", type=float, default=18.0, help="budget of the cpu_baseline leg, split over its three variants (0 = skip)")
    ap.add_argument("--latency-reps", type=int, default=15, help="repetitions of the B = 1 pose-optimisation call (0 = skip the whole latency block)")
    ap.add_argument("--latency-frames", type=int, default=200, help="distinct frames of the B = 1 latency block (4/5 se3 frames, 1/5 panned canvas windows)")
    ap.add_argument("--sub-steps", type=int, default=20, help="steps of the BASELINE configs[1] / [3] / [4] sub-runs folded into the default line as sub_benchmarks (0 = skip)")
    ap.add_argument("--pcie-steps", type=int, default=4, help="steps of the PCIe-inclusive loop (0 = skip)")
    ap.add_argument("--canvases", type=int, default=NCANVAS, help="distinct synthetic canvases (pan) / room scenes with their textures (se3) per GPU (the streams tile over them)")
    ap.add_argument("--streams", choices=["se3", "pan"], default="se3",
                    help="se3: every stream is a camera moving through a textured box room - gray and depth of every frame ray-cast from the same SE3 pose "
                         "(planarslam_amd/synth_se3.py; rendered on the GPU before the timed region, --loop frames per stream resident in HBM, played forwards and "
                         "backwards); pan: rounds 1-3's streams - a 640x480 window panning over unrelated gray / depth canvases")
    ap.add_argument("--loop", type=int, default=12, help="frames per stream of the se3 loop (resident: B x loop x 0.92 MB)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the barrier / max-over-ranks (nccl = RCCL, one rank per GPU).  gloo: ranks may share a GPU "
                         "(rank r uses device r mod device_count; the BA workload then exchanges through the hosted transport) - how the N > 1 "
                         "launch, rendezvous, shard and print paths are exercised on a 1-GPU box")
    ap.add_argument("--gen-procs", type=int, default=0, help="worker processes that synthesise the canvases (0 = up to 32; 1 = in this process: rocprofv3 --pmc hangs in "
                                                             "forked children, profiles/README.md)")
    args = ap.parse_args()
    # no timing / variant switch reaches the timed region through the environment: the product reads none (PLANAR_HIP_LIB, a developer build of the same library, is
    # reported in the line; stage skipping lives in tools/stage_skip.py, which relabels its output)
    bad = sorted(k for k in os.environ if k.startswith("PLANAR_") and k not in ("PLANAR_HIP_LIB", "PLANAR_ORACLE_LIB"))
    if bad:
        raise SystemExit(f"bench.py: refusing to run with {bad} set: the bench line is the product's default path only")
    global ENV_OVERRIDES
    ENV_OVERRIDES = [f"{k}={os.environ[k]}" for k in ("PLANAR_HIP_LIB",) if os.environ.get(k)]

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
    rank_env = int(os.environ.get("RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    if args.workload == "pose":
        return main_pose(args)
    if args.workload == "ba":
        return main_ba(args)
    full = args.workload == "full"
    B = args.batch or (2048 if full else 1024)
    # ---- inputs: generated on worker processes BEFORE the GPU runtime is initialised (fork) ----
    from planarslam_amd.synth import TUM3, pan_offset, stream_canvases
    ncanv = min(args.canvases, B)
    t_gen = time.perf_counter()
    canv_g, canv_d = stream_canvases(ncanv, rank_env, W + 2 * MARGIN, H + 2 * MARGIN, procs=args.gen_procs or max(1, min(32, (os.cpu_count() or 1) // max(1, world_env))))
    t_gen = time.perf_counter() - t_gen
    canv_g0, canv_d0 = canv_g, canv_d            # (the panned windows of the B = 1 latency block)

    import torch

    from planarslam_amd.dist import Ranks, whole_job_fps
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local_rank = pick_device(args, torch)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = Ranks(backend=args.backend, device=dev)    # RCCL (or gloo): barrier + max-over-ranks only (frames shard, no data collective)
    rank, world = ranks.rank, ranks.world

    from planarslam_amd import Context, ORBextractor, Optimizer, PlaneDetection
    from planarslam_amd._lib import check, lib
    from planarslam_amd.track import TrackPipeline, build_map

    prio = [int(x) for x in args.prio.split(',')]
    d_canv_g = torch.from_numpy(canv_g).to(dev)
    se3 = args.streams == "se3"
    if se3:
        # B camera streams in ncanv textured rooms (stream s: room s mod ncanv, its own SE3 path), args.loop frames each, rendered here and resident in HBM
        from planarslam_amd import synth_se3
        t_r = time.perf_counter()
        loop_g, loop_d, Twc_true = synth_se3.render_streams(torch, d_canv_g, B, args.loop, TUM3, seed=rank_env, W=W, H=H)
        torch.cuda.synchronize()
        t_gen += time.perf_counter() - t_r
        canv_g = loop_g[:1, 0].cpu().numpy(); canv_d = loop_d[:1, 0].cpu().numpy().view(np.uint16)     # (the B = 1 latency block and the CPU leg take stream 0's first frame)
        canv_g = np.pad(canv_g, ((0, 0), (MARGIN, MARGIN), (MARGIN, MARGIN))); canv_d = np.pad(canv_d, ((0, 0), (MARGIN, MARGIN), (MARGIN, MARGIN)))

        def window(i, out_g, out_d):
            """The frames of step i: a strided device copy out of the resident loops (part of the step)."""
            fi = synth_se3.frame_index(i, args.loop)
            out_g.copy_(loop_g[:, fi])
            if out_d is not None:
                out_d.copy_(loop_d[:, fi])
    else:
        d_canv_d = torch.from_numpy(canv_d.view(np.int16)).to(dev)
        ngroups = (B + ncanv - 1) // ncanv
        goff = [(24 * (g & 1), 24 * ((g >> 1) & 1)) for g in range(ngroups)]   # streams that share a canvas look through windows 24 px apart

        def window(i, out_g, out_d):
            """The frames of step i: a strided device copy out of the resident canvases (part of the step)."""
            ox, oy = pan_offset(i, MARGIN)
            for g in range(ngroups):
                a, b = g * ncanv, min(B, (g + 1) * ncanv)
                x0, y0 = ox + goff[g][0], oy + goff[g][1]
                out_g[a:b].copy_(d_canv_g[:b - a, y0:y0 + H, x0:x0 + W])
                if out_d is not None:
                    out_d[a:b].copy_(d_canv_d[:b - a, y0:y0 + H, x0:x0 + W])

    L = lib()
    if full:
        torch.cuda.synchronize()
        free0, talloc0 = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_allocated(dev)
        tp = TrackPipeline(B, torch, local_rank, depth=args.depth, prio=prio, cam=TUM3, W=W, H=H, work_sets=args.work_sets or None, seq_cus=args.seq_cus,
                           seq_which=args.seq_which, seq_shared=not args.seq_per_ctx)
        torch.cuda.synchronize()
        # device memory the pipeline holds per frame of the batch: the extractors' workspaces (hipMalloc inside the library: ORB's one set + NW sets of LSD / PEAC / plane
        # clouds / normals) = what left the device's free pool minus what torch's allocator took for the NB sets of outputs, the per-stream state and the optimiser buffers
        mem_total = free0 - torch.cuda.mem_get_info(dev)[0]
        mem_torch = torch.cuda.memory_allocated(dev) - talloc0
        MEM = {"workspace_bytes_per_frame": int(max(0, mem_total - mem_torch) // B), "output_and_state_bytes_per_frame": int(mem_torch // B),
               "pipeline_bytes_total": int(mem_total), "extractor_sets": tp.NW, "buffer_sets": tp.NB,
               "note": "workspace = hipMalloc'ed inside libplanar_hip.so (free-memory delta minus torch's allocations; torch's caching allocator may round the latter up)"}
        NB = tp.NB
        stream, ex = tp.stream, tp.ex
        frames = [torch.zeros((B, H, W), dtype=torch.uint8, device=dev) for _ in range(NB)]
        depths = [torch.zeros((B, H, W), dtype=torch.int16, device=dev) for _ in range(NB)]
        # the per-stream map stand-ins (reference key frame's lines, map planes) from the streams' first frames, through the product extractors
        window(0, frames[0], depths[0])
        torch.cuda.synchronize()
        nmap = B if se3 else min(B, ncanv)      # (se3: every stream has its own first frame, hence its own map)
        kf_lines, map_planes, normals = build_map(frames[0][:nmap].cpu().numpy(), depths[0][:nmap].cpu().numpy().view(np.uint16), TUM3, seed=rank)
        rep = lambda a: np.concatenate([a] * ((B + len(a) - 1) // len(a)))[:B]
        tp.set_map({k: rep(v) for k, v in kf_lines.items()}, {k: rep(v) for k, v in map_planes.items()}, {k: rep(v) for k, v in normals.items()})
    else:
        stream = torch.cuda.Stream(device=local_rank, priority=prio[0])
        ctx = Context(local_rank, stream=stream.cuda_stream)
        ex = ORBextractor(1000, 1.2, 8, 20, 7, width=W, height=H, max_batch=B, ctx=ctx)
        NB = 2
        frames = [torch.zeros((B, H, W), dtype=torch.uint8, device=dev) for _ in range(NB)]
        o_kps = torch.zeros((B, ex.kp_cap, 7), dtype=torch.float32, device=dev); o_desc = torch.zeros((B, ex.kp_cap, 32), dtype=torch.uint8, device=dev)
        o_n = torch.zeros(B, dtype=torch.int32, device=dev)

    PC_SLOTS = ("plane_voxels", "plane_items", "plane_sort_global", "plane_sort_lds", "plane_sort_heap", "plane_tail")
    EV = ("start", "orb", "stereo", "wait0", "wait1", "manhattan", "proj", "bf", "planes", "transl", "local", "pose", "state")

    def step(i, evs=None, side=None):
        k = i % NB
        if full:
            stream.wait_event(tp.done[k])      # the window buffers of step i - NB are free
            window(i, frames[k], depths[k])
            tp.step(i, frames[k], depths[k], evs, side)
        else:
            window(i, frames[k], None)
            if evs: evs["start"].record(stream)
            ex.extract_dev(frames[k].data_ptr(), o_kps.data_ptr(), o_desc.data_ptr(), o_n.data_ptr(), B)
            if evs: evs["orb"].record(stream)

    def barrier():
        torch.cuda.synchronize()
        ranks.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            step(i)
        if full: tp.drain()
        standalone = {}
        if full:   # calibration: each sequential extractor alone on the device (not part of the timed region)
            pd0, ls0 = tp.pds[0], tp.lss[0]
            for name, fn in (("peac_alone_ms", lambda: pd0.segment_dev(depths[0].data_ptr(), tp.lab[0].data_ptr(), tp.pls[0].data_ptr(), tp.npl[0].data_ptr(), B)),
                             ("lsd_lbd_alone_ms", lambda: check(L.planar_lsd_extract_dev(ls0.h, frames[0].data_ptr(), B, W, W * H, 40, tp.kls[0].data_ptr(),
                                                                                          tp.ldesc[0].data_ptr(), tp.leq[0].data_ptr(), tp.nl[0].data_ptr())))):
                torch.cuda.synchronize()
                fn(); torch.cuda.synchronize()
                check(L.planar_peac_set_profiling(pd0.h, 1)); check(L.planar_lsd_set_profiling(ls0.h, 1))
                t1 = time.perf_counter(); fn(); torch.cuda.synchronize()
                standalone[name] = round((time.perf_counter() - t1) * 1e3, 3)
                import ctypes as C0
                tot = np.zeros(4); nc = C0.c_int64()
                if name.startswith("peac"):
                    check(L.planar_peac_get_profile(pd0.h, tot.ctypes.data, C0.byref(nc)))
                    standalone["peac_kernels_alone_ms"] = dict(zip(("peac_blocks", "peac_ahc", "peac_order", "peac_refine"), (round(float(x), 3) for x in tot)))
                else:
                    check(L.planar_lsd_get_profile(ls0.h, tot.ctypes.data, C0.byref(nc)))
                    standalone["lsd_kernels_alone_ms"] = dict(zip(("preprocess", "lsd_sort", "lsd_detect", "improve+accept+keylines+lbd"), (round(float(x), 3) for x in tot)))
                check(L.planar_peac_set_profiling(pd0.h, 0)); check(L.planar_lsd_set_profiling(ls0.h, 0))
            # the plane clouds (Frame::ComputePlanes head: voxel grid in PCL's std::sort order + refit) alone, on the labels the calibration launch above left
            pc0 = tp.pcs[0]
            run_pc = lambda: pc0.compute_dev(depths[0].data_ptr(), tp.lab[0].data_ptr(), tp.pls[0].data_ptr(), tp.npl[0].data_ptr(), B, tp.pc[0]["n"].data_ptr(), tp.pc[0]["coef"].data_ptr(),
                                             tp.pc[0]["src"].data_ptr(), tp.pc[0]["off"].data_ptr(), tp.pc[0]["pts"].data_ptr(), tp.pc[0]["status"].data_ptr())
            run_pc(); torch.cuda.synchronize()
            check(L.planar_plane_clouds_set_profiling(pc0.h, 1))
            t1 = time.perf_counter(); run_pc(); torch.cuda.synchronize()
            standalone["plane_clouds_alone_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
            tot6 = np.zeros(6); nc = C0.c_int64()
            check(L.planar_plane_clouds_get_profile(pc0.h, tot6.ctypes.data, C0.byref(nc)))
            check(L.planar_plane_clouds_set_profiling(pc0.h, 0))
            standalone["plane_clouds_kernels_alone_ms"] = dict(zip(PC_SLOTS, (round(float(x), 3) for x in tot6)))
            st4 = np.zeros((B, 4), np.int64)
            check(L.planar_plane_clouds_sort_stats(pc0.h, B, st4.ctypes.data))
            standalone["plane_sort_stats"] = {"frames_with_heap_sort_fallback": int((st4[:, 0] > 0).sum()), "fallback_ranges": int(st4[:, 0].sum()), "fallback_elements": int(st4[:, 1].sum()),
                                              "longest_fallback_range": int(st4[:, 2].max()), "lds_blocks_per_frame": round(float(st4[:, 3].mean()), 1),
                                              "note": "std::sort of pcl::VoxelGrid::applyFilter per plane: ranges whose introsort depth budget ran out go through libstdc++'s heap sort (isort.h)"}
        ex.set_profiling(True)
        if full:
            for q in tp.pds: check(L.planar_peac_set_profiling(q.h, 1))
            for q in tp.lss: check(L.planar_lsd_set_profiling(q.h, 1))
            for q in tp.pcs: check(L.planar_plane_clouds_set_profiling(q.h, 1))
        evsets = [{n: torch.cuda.Event(enable_timing=True) for n in EV} for _ in range(args.steps)]
        sides = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, evsets[i], sides[i])
        if full: tp.drain()            # the last steps' tracking chains: all K steps are complete inside the timed region
        barrier()
        elapsed = time.perf_counter() - t0
        prof, calls = ex.get_profile()
        ex.set_profiling(False)
        peac_ms, peac_calls = np.zeros(4), 0
        if full:
            import ctypes as C
            for q in tp.pds:                       # one plan per buffer set: HIP events right before / after each of the four launches, on the stream they run on
                tot = np.zeros(4); nc = C.c_int64()
                check(L.planar_peac_get_profile(q.h, tot.ctypes.data, C.byref(nc)))
                check(L.planar_peac_set_profiling(q.h, 0))
                peac_ms += tot; peac_calls += nc.value
            lsd_ms, lsd_calls = np.zeros(4), 0
            for q in tp.lss:
                tot = np.zeros(4); nc = C.c_int64()
                check(L.planar_lsd_get_profile(q.h, tot.ctypes.data, C.byref(nc)))
                check(L.planar_lsd_set_profiling(q.h, 0))
                lsd_ms += tot; lsd_calls += nc.value
            pc_ms, pc_calls = np.zeros(6), 0
            for q in tp.pcs:
                tot = np.zeros(6); nc = C.c_int64()
                check(L.planar_plane_clouds_get_profile(q.h, tot.ctypes.data, C.byref(nc)))
                check(L.planar_plane_clouds_set_profiling(q.h, 0))
                pc_ms += tot; pc_calls += nc.value
    parity = None
    if full:
        tp.check()
        if PARITY_SAMPLE and rank_env == 0:
            parity = parity_sample(tp, frames, depths, (args.warmup + args.steps - 1) % NB, B, TUM3)

    seg = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in evsets) / args.steps
    stage_ms = {"orb_extract": seg("start", "orb")}
    if full:
        stage_ms.update({"compute_stereo_from_rgbd": seg("orb", "stereo"), "wait_lines_planes": seg("wait0", "wait1"), "track_manhattan_frame": seg("wait1", "manhattan"),
                         "search_by_projection_last": seg("manhattan", "proj"), "lsd_search_by_descriptor+match_orb_points": seg("proj", "bf"),
                         "plane_matcher": seg("bf", "planes"), "assemble+translation_opt_4x10+discard": seg("planes", "transl"),
                         "is_in_frustum+search_local_points_lines": seg("transl", "local"), "assemble+pose_opt_4x10": seg("local", "pose"),
                         "unproject+map_state": seg("pose", "state")})
        stage_ms["peac(stream 2)"] = sum(e[0].elapsed_time(e[1]) for e in sides) / args.steps
        stage_ms["lsd_lbd(stream 3)"] = sum(e[2].elapsed_time(e[3]) for e in sides) / args.steps
    elapsed = ranks.max_over_ranks(elapsed)
    n_last = tp.n[(args.warmup + args.steps - 1) % NB] if full else o_n
    n_kp = int(n_last.sum().item())
    if rank != 0:
        ranks.close()
        return

    fps = whole_job_fps(world, B, args.steps, elapsed)
    avg_kp = n_kp / B
    kernels = {k: {"ms_per_step": round(v[0] / max(1, calls), 4), "launches_per_step": v[1] // max(1, calls)} for k, v in prof.items()}
    alg = orb_algorithmic_bytes(ex, avg_kp)
    quality = None
    t_pc_pts = lambda q: torch.gather(q.pc[0]["off"], 1, q.pc[0]["n"].long().unsqueeze(1)).float().mean().item()     # pt_off[b][n[b]] = voxel centroids kept
    if full:
        # PEAC: read u16 depth + write int32 labels (1 843 200 B/frame); peac_blocks + peac_ahc + peac_order + peac_refine bracketed together
        pk = "peac_blocks+peac_ahc+peac_refine"
        pav = peac_ms / max(1, peac_calls)          # average duration of each launch, HIP events tight around it (co-run: other streams share the CUs)
        kernels[pk] = {"ms_per_step": round(float(pav.sum()), 4), "launches_per_step": 4, "alone_ms": standalone["peac_alone_ms"],
                       "avg_launch_ms": {"peac_blocks": round(float(pav[0]), 3), "peac_ahc": round(float(pav[1]), 3), "peac_order": round(float(pav[2]), 3),
                                         "peac_refine": round(float(pav[3]), 3)},
                       "stream_bracket_ms": round(stage_ms["peac(stream 2)"], 3),
                       "note": "avg_launch_ms: HIP events right before / after each launch on the PEAC stream, inside the timed region (what rocprofv3's AverageNs measures); "
                               "stream_bracket_ms also holds the surface-normal kernel and the queueing between launches with up to depth+2 launch sets in flight"}
        # plane clouds (Frame::ComputePlanes head): read labels + depth (1 843 200 B / frame), write <= 4096 centroids; its launches bracketed one by one
        ck = "plane_clouds(voxels+items+sort+tail)"
        cav = pc_ms / max(1, pc_calls)
        kernels[ck] = {"ms_per_step": round(float(cav.sum()), 4), "launches_per_step": 8, "alone_ms": standalone["plane_clouds_alone_ms"],
                       "avg_launch_ms": dict(zip(PC_SLOTS, (round(float(x), 3) for x in cav))), "alone_launch_ms": standalone["plane_clouds_kernels_alone_ms"],
                       "sort_stats_of_the_calibration_batch": standalone["plane_sort_stats"]}
        lk = "lsd_detect(+7 small kernels)"
        kernels[lk] = {"ms_per_step": round(stage_ms["lsd_lbd(stream 3)"], 4), "launches_per_step": 8, "alone_ms": standalone["lsd_lbd_alone_ms"]}
        kernels["projection_kernel"] = {"ms_per_step": round(stage_ms["search_by_projection_last"], 4), "launches_per_step": 1}
        A1 = tp.pb_arrays[1]
        lm_it = float(A1["lm_iters"].float().mean().item())
        kernels["pose_opt_kernel"] = {"ms_per_step": round(stage_ms["assemble+pose_opt_4x10"], 4), "launches_per_step": 2, "avg_lm_iterations": round(lm_it, 2)}
        quality = {"avg_planes_per_frame": round(float(tp.npl[0].float().mean().item()), 2), "avg_planes_kept_per_frame": round(float(tp.pc[0]["n"].float().mean().item()), 2),
                   "avg_plane_cloud_points_per_frame": round(float(t_pc_pts(tp)), 1), "avg_lines_per_frame": round(float(tp.nl[0].float().mean().item()), 2),
                   "avg_projection_matches_per_frame": round(float(tp.nm.float().mean().item()), 1),
                   "avg_local_map_matches_per_frame": round(float(tp.nmm.float().mean().item()), 1),
                   "avg_line_matches_per_frame": round(float((tp.lm >= 0).float().sum(1).mean().item()), 1),
                   "avg_plane_matches_per_frame": round(float(tp.nplm.float().mean().item()), 2),
                   "avg_translation_opt_inliers": round(float(tp.pb_arrays[0]["n_inliers"].float().mean().item()), 1),
                   "avg_pose_opt_inliers": round(float(A1["n_inliers"].float().mean().item()), 1)}
    ms_per_step = elapsed / args.steps * 1e3
    # ---- roofline: ONE KERNEL, the one with the most device time per step.  Durations: HIP events right before / after each launch on the stream the kernel runs on,
    #      inside the timed region (co-run with the other streams: what rocprofv3's AverageNs of the same command measures), and the same launch alone on the device
    #      (calibration launch before the timed region: rocprofv3's Min).  Bytes: SURVEY §8d's algorithmic figure of the kernel's stage x frames per launch. ----
    per = {}

    def add_kernel(name, rocprof, corun, alone, nbytes, launches=1):
        e = {"rocprof_name": rocprof, "alg_bytes_per_launch": int(nbytes), "corun_avg_ms": round(float(corun), 3), "alone_ms": None if alone is None else round(float(alone), 3),
             "launches_per_step": launches}
        if corun > 0: e["frac_corun"] = round(nbytes / (corun * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
        if alone: e["frac_alone"] = round(nbytes / (alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
        per[name] = e
    for k, v in prof.items():                                  # the ORB kernels (per-launch events of the extractor): average over their launches
        add_kernel(k, "planar::orb::" + k, v[0] / max(1, v[1]), None, alg.get(k, 0) * B, v[1] // max(1, calls))
    if full:
        lav = lsd_ms / max(1, lsd_calls)
        ka, la, ca = standalone.get("peac_kernels_alone_ms", {}), standalone.get("lsd_kernels_alone_ms", {}), standalone["plane_clouds_kernels_alone_ms"]
        add_kernel("peac_blocks", "planar::peac::peac_blocks", pav[0], ka.get("peac_blocks"), 614400 * B)
        add_kernel("peac_ahc3", "planar::peac::peac_ahc3", pav[1], ka.get("peac_ahc"), 1843200 * B)
        add_kernel("peac_refine", "planar::peac::peac_refine", pav[3], ka.get("peac_refine"), 1843200 * B)
        add_kernel("lsd_sort(4 launches)", "planar::lsd::lsd_sort", lav[1], la.get("lsd_sort"), 312160 * B, 4)
        add_kernel("lsd_detect", "planar::lsd::lsd_detect", lav[2], la.get("lsd_detect"), 312160 * B)
        for i, nm in enumerate(PC_SLOTS):
            add_kernel(nm, "planar::planepost::" + nm + ("" if nm.startswith("plane_sort") else "_kernel"), cav[i], ca.get(nm), 1843200 * B, 2 if nm == "plane_sort_heap" else 1)
    # "most device time": the launch's duration ALONE on the device where the calibration measured it (the co-run brackets of the wide kernels mostly measure how long a
    # launch queued for CUs behind the other streams - lsd_sort's four launches: 4.7 ms alone, 100 ms of brackets); the ORB-only workload has only the live averages
    dev_time = lambda k: (per[k]["alone_ms"] if per[k]["alone_ms"] else 0.0) if full else per[k]["corun_avg_ms"] * per[k]["launches_per_step"]
    dom = max(per, key=dev_time)
    dom_ms, dom_bytes = per[dom]["corun_avg_ms"], per[dom]["alg_bytes_per_launch"]
    if dom_ms > ms_per_step:
        print(f"bench.py: WARNING: the HIP-event bracket of {dom} ({dom_ms:.1f} ms) exceeds the step ({ms_per_step:.1f} ms): launches of several steps overlap on its stream", file=sys.stderr)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # algorithmic bytes of the whole step per frame: ORB; + PEAC (depth in, labels out) + normals; + plane clouds (labels + depth in); + matchers + the two pose problems
    # (65 130 B per LM evaluation x the MEASURED evaluations of both optimisers)
    lm_both = (float(tp.pb_arrays[0]["lm_iters"].float().mean().item()) + float(tp.pb_arrays[1]["lm_iters"].float().mean().item())) if full else 0.0
    per_frame = 1961064 + ((1843200 + 312160) + 1843200 + 72000 + 118000 + 65130 * lm_both if full else 0)
    # HBM-side traffic of that kernel from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this command, KB per launch;
    # raw counter sums); a kernel without a row is reported as null with the reason, never as zero
    traffic = None
    pmc_csv = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_fetch_write_kb_per_launch.csv") for r in (6, 5, 4, 3, 2)) if os.path.exists(q)), "")
    traffic_note = "null: no profiles/r0N_pmc_fetch_write_kb_per_launch.csv (tools/pmc_counters.py) found"
    if os.path.exists(pmc_csv):
        f_tot = w_tot = 0.0
        for ln in open(pmc_csv).read().splitlines()[1:]:
            k, _, f_kb, w_kb = ln.rsplit(",", 3)
            if k.strip('"').replace("void ", "").startswith(per[dom]["rocprof_name"]):
                f_tot += float(f_kb); w_tot += float(w_kb)
        if f_tot + w_tot > 0:
            traffic = int((f_tot + w_tot) * 1024 * B / 1024)                      # (the counter passes run B = 1024: scaled to this run's frames per launch)
            traffic_note = (f"NOT measured in this run: read from the committed {os.path.relpath(pmc_csv, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of "
                            f"`bench.py --canvases 16 --gen-procs 1`, B=1024; counter collection hangs in forked children): {f_tot / 1024:.0f} MB read + {w_tot / 1024:.0f} MB written per launch")
        else:
            traffic_note = f"null: {os.path.relpath(pmc_csv, ROOT)} has no row for {per[dom]['rocprof_name']} (re-collect with tools/collect_profiles.sh)"
            print(f"bench.py: WARNING: roofline.traffic not reported: {traffic_note}", file=sys.stderr)
    # the lens that fits this path (no dense contraction, L2-resident working sets): instruction issue.  issue_frac = (4 x VALU + SALU wavefront-instructions) / (launch
    # duration x 1024 SIMDs x 2.4 GHz), active / wait = share of the wavefronts' resident time with an instruction in flight / waiting for one; from the committed SQ
    # counter passes (tools/pmc_counters.py: B = 1024, every launch alone on the device, no CU partition), like `traffic`
    issue = None
    sq_csv = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_sq_per_launch.csv") for r in (6, 5)) if os.path.exists(q)), "")
    if os.path.exists(sq_csv):
        import csv as _csv
        rows = list(_csv.DictReader(open(sq_csv)))
        def _issue(rn):
            for r_ in rows:
                if r_["kernel"].replace("void ", "").startswith(rn) and r_.get("issue_frac") not in (None, "", "nan"):
                    return {"issue_frac": float(r_["issue_frac"]), "active": float(r_["active"]), "wait": float(r_["wait"]), "valu_winst_per_launch": int(float(r_["SQ_INSTS_VALU"])),
                            "salu_winst_per_launch": int(float(r_["SQ_INSTS_SALU"])), "alone_us_per_launch_b1024": float(r_["avg_duration_us"])}
            return None
        issue = _issue(per[dom]["rocprof_name"])
        if issue:
            issue["source"] = os.path.relpath(sq_csv, ROOT)
            issue["note"] = ("NOT measured in this run: committed rocprofv3 --pmc passes (SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY, "
                             "launches of 1024 frames alone on the device); issue_frac = (4 VALU + SALU) / (duration x 1024 SIMDs x 2.4 GHz)")
        for k_, v_ in per.items():
            q_ = _issue(v_["rocprof_name"])
            if q_: v_["issue_frac"] = q_["issue_frac"]; v_["active"] = q_["active"]
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic, "traffic_source": os.path.relpath(pmc_csv, ROOT) if traffic is not None else None, "traffic_note": traffic_note,
                "issue": issue,
                "avg_launch_ms": round(dom_ms, 4), "alone_ms": per[dom]["alone_ms"], "frac_alone": per[dom].get("frac_alone"), "algorithmic_bytes_per_launch": int(dom_bytes),
                "pipeline_algorithmic_GBps": round(per_frame * fps / 1e9, 2), "algorithmic_bytes_per_frame_of_the_step": int(per_frame),
                "note": ("one kernel (the one with the most device time per step), not a stage: avg_launch_ms = HIP events right before / after each of its launches on its own stream "
                         "inside the timed region (co-run with the other streams; rocprofv3's AverageNs of the same command), alone_ms = the same launch alone on the device "
                         "(rocprofv3's Min).  peac_ahc3 / lsd_detect: one sequential wavefront per frame, latency-bound (DESIGN.md §4)"),
                "per_kernel": per, "stages": kernels}
    if full and args.seq_cus and (args.seq_which & 1) and dom == "peac_ahc3":
        roofline["cu_partition"] = {"kernel_confined_to_cus": args.seq_cus, "of": 256,
                                    "note": ("this kernel runs on a CU-masked side stream (planar_ctx_set_seq_stream): its launches take 256 / %d times longer than on the whole device - by "
                                             "design, the other CUs run the wide kernels meanwhile - so `frac` (bytes / duration / 8 TB/s of the WHOLE device) is that much lower than "
                                             "with --seq-cus 0; alone_ms is measured on the same mask") % args.seq_cus,
                                    "frac_relative_to_its_share_of_hbm": round(achieved / HBM_PEAK_GBS * 256.0 / args.seq_cus, 6)}
    if full:
        roofline["per_kernel_note"] = ("co-run >> alone for the wide kernels: up to depth + 2 steps are in flight and every kernel waits for CUs behind the other streams' launches; "
                                       "the single-wavefront kernels of the two side chains stretch when other wavefronts share their SIMDs (tools/corun_probe.py; DESIGN.md)")

    # ---- PCIe-inclusive rate: the same step fed from pinned host memory and drained to it (H2D / D2H on copy streams, overlapped) ----
    pcie = None
    if full and world == 1 and args.pcie_steps > 0:
        h_g = [torch.empty((B, H, W), dtype=torch.uint8).pin_memory() for _ in range(NB)]
        h_d = [torch.empty((B, H, W), dtype=torch.int16).pin_memory() for _ in range(NB)]
        for k in range(NB):
            window(100 + k, frames[k], depths[k]); torch.cuda.synchronize()
            h_g[k].copy_(frames[k]); h_d[k].copy_(depths[k])
        S = tp.S
        h_out = [dict(kps=torch.empty((B, S, 7), dtype=torch.float32).pin_memory(), desc=torch.empty((B, S, 32), dtype=torch.uint8).pin_memory(),
                      kls=torch.empty(tp.kls[0].shape, dtype=torch.uint8).pin_memory(), ldesc=torch.empty((B, 40, 32), dtype=torch.uint8).pin_memory(),
                      lab=torch.empty((B, H * W), dtype=torch.int32).pin_memory(), pls=torch.empty(tp.pls[0].shape, dtype=torch.float64).pin_memory(),
                      pose=torch.empty((B, 16), dtype=torch.float32).pin_memory(), plc=torch.empty((B, tp.PS, 4), dtype=torch.float32).pin_memory(),
                      plp=torch.empty((B, 1024, 3), dtype=torch.float32).pin_memory()) for _ in range(2)]
        s_in, s_out = torch.cuda.Stream(device=local_rank), torch.cuda.Stream(device=local_rank)
        ev_up = [torch.cuda.Event() for _ in range(NB)]
        ev_dn = [torch.cuda.Event() for _ in range(NB)]
        ev_out = [torch.cuda.Event() for _ in range(NB)]
        nsteps = args.pcie_steps + args.depth
        torch.cuda.synchronize()
        base = ((args.warmup + args.steps + 8 + NB - 1) // NB) * NB
        with torch.cuda.stream(stream):
            t1 = None
            for q in range(nsteps):
                i = base + q
                k = i % NB
                if q == args.depth: torch.cuda.synchronize(); t1 = time.perf_counter()     # the pipeline is primed
                s_in.wait_event(tp.done[k]); s_in.wait_event(ev_out[k])       # chain and download of step i - NB have finished with these buffers
                with torch.cuda.stream(s_in):
                    frames[k].copy_(h_g[k], non_blocking=True); depths[k].copy_(h_d[k], non_blocking=True)
                    ev_up[k].record(s_in)
                stream.wait_event(tp.done[k]); stream.wait_event(ev_up[k]); stream.wait_event(ev_out[k])
                tp.step(i, frames[k], depths[k])
                j = i - args.depth                   # its tracking chain has just been enqueued: drain its outputs
                if q >= args.depth:
                    kj = j % NB
                    ev_dn[kj].record(tp.s_track)
                    with torch.cuda.stream(s_out):
                        s_out.wait_event(ev_dn[kj])
                        o = h_out[q & 1]
                        o["kps"].copy_(tp.kps[kj], non_blocking=True); o["desc"].copy_(tp.desc[kj], non_blocking=True); o["kls"].copy_(tp.kls[kj], non_blocking=True)
                        o["ldesc"].copy_(tp.ldesc[kj], non_blocking=True); o["lab"].copy_(tp.lab[kj], non_blocking=True); o["pls"].copy_(tp.pls[kj], non_blocking=True)
                        o["pose"].copy_(tp.pose, non_blocking=True); o["plc"].copy_(tp.pc[kj]["coef"], non_blocking=True); o["plp"].copy_(tp.pc[kj]["pts"][:, :1024], non_blocking=True)
                        ev_out[kj].record(s_out)
            tp.drain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
        per_frame_in, per_frame_out = W * H * 3, S * 60 + 40 * 100 + W * H * 4 + 128 * 64 + 64 + 128 * 16 + 1024 * 12
        pcie = {"value": round(B * args.pcie_steps / dt, 1), "unit": "frames/s", "steps": args.pcie_steps,
                "h2d_bytes_per_frame": per_frame_in, "d2h_bytes_per_frame": per_frame_out,
                "note": "pinned host buffers; gray + depth up, keypoints / descriptors / key lines / label image / planes / plane coefficients + first 1024 cloud points / pose down, on copy streams beside the step"}

    # ---- CPU baseline: the oracle restatement of the same stages on this box's host cores (tools/cpu_baseline.py) ----
    cpu = None
    if args.cpu_seconds > 0 and world == 1:        # rank 0 at N = 1 only
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cpu_baseline as cb
        cb.use_fast_build()                          # -O3 -march=native, compiled on this box (the tests keep checking against the -O2 build)
        import oracle_lib as _ol
        _ol._LIB = None                              # (the parity sample above loaded the checker's -O2 build: the timed legs below load the fast one)
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
               "host_cores": ncores, "cpu_model": cb.cpu_model(), "build": cb.build_flags()}

    # ---- single-frame latency (B = 1, host-pointer entry points: H2D + kernels + D2H + sync; the reference is a live B = 1 tracker).  Every stage the adapters call per
    #      frame (include/planar_adapters.hpp): ORBextractor, ExtractLineSegment + isLineGood, PlaneDetection + ComputePlanes' voxel clouds / refit + surface normals,
    #      PoseOptimization; over `--latency-frames` DISTINCT frames (se3 frames of different streams, then panned canvas windows: the regime with heap-sort fallbacks),
    #      each frame timed once after an untimed pass of the first frames: p50 / p99 / max ----
    latency = None
    if world == 1 and full and args.latency_reps > 0:
        from planarslam_amd.lines import LineSegment as LS1
        from planarslam_amd.lines import is_line_good as ilg1
        from planarslam_amd.matcher import ORBmatcher as OM1
        from planarslam_amd.planes import PlaneClouds as PC1
        from planarslam_amd.planes import SurfaceNormals as SN1
        from planarslam_amd.synth import pose_batch
        import threading
        nlat = max(8, args.latency_frames)
        lat_g, lat_d, kinds = [], [], []
        if se3:
            ns = min(B, (nlat * 4) // 5)
            for q in range(ns):
                fi = (q * 5) % args.loop
                lat_g.append(loop_g[q, fi].cpu().numpy()); lat_d.append(loop_d[q, fi].cpu().numpy().view(np.uint16)); kinds.append("se3")
        npan = nlat - len(lat_g)
        for q in range(npan):                                     # panned windows over the unrelated gray / depth canvases of rounds 1-3
            c_, ox = q % len(canv_g0), 8 * ((q // len(canv_g0)) % 12)
            lat_g.append(np.ascontiguousarray(canv_g0[c_, MARGIN:MARGIN + H, ox:ox + W])); lat_d.append(np.ascontiguousarray(canv_d0[c_, MARGIN:MARGIN + H, ox:ox + W])); kinds.append("pan")
        cs = [Context(local_rank) for _ in range(3)]
        ex1 = ORBextractor(1000, 1.2, 8, 20, 7, width=W, height=H, max_batch=1, ctx=cs[0])
        ls1 = LS1(W, H, 1, cs[1])
        pd1 = PlaneDetection(W, H, max_batch=1, ctx=cs[2]); pc1 = PC1(W, H, 1, ctx=cs[2]); sn1 = SN1(W, H, 1, cs[2])
        opt1 = Optimizer(TUM3, ctx=cs[0]); om1 = OM1(ctx=cs[0])
        pb1 = pose_batch(B=1, n_points=1000, n_lines=75, n_planes=4, seed=7)
        PSn = pd1.max_planes
        Kc = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"])
        tms = {k: [] for k in ("orb_extract", "lsd_lbd_extract", "is_line_good", "peac_segment", "plane_clouds_refit", "surface_normals", "pose_opt_4x10", "frame_3_threads")}

        def timed(key, fn, rec):
            t1 = time.perf_counter(); r = fn(); dt = (time.perf_counter() - t1) * 1e3
            if rec: tms[key].append(dt)
            return r

        def points(g, rec):
            timed("orb_extract", lambda: ex1(g), rec)

        def lines(g, d, rec, i):
            kl, _, _, nl_ = timed("lsd_lbd_extract", lambda: ls1.ExtractLineSegment(g), rec)
            timed("is_line_good", lambda: ilg1(kl, nl_, d[None], np.array([64 * i], np.int32), cam=Kc, ctx=cs[1]), rec)

        def planes(d, rec):
            pls_, lab_ = timed("peac_segment", lambda: pd1.run(d[None])[0], rec)
            pl = np.zeros((1, PSn, 8)); pl[0, :len(pls_)] = pls_
            timed("plane_clouds_refit", lambda: pc1.compute(d[None], lab_[None], pl, np.array([len(pls_)], np.int32), K=Kc), rec)
            timed("surface_normals", lambda: sn1.compute(d, K=Kc), rec)

        def frame_threads(g, d, i):
            th = [threading.Thread(target=f) for f in (lambda: points(g, False), lambda: lines(g, d, False, i), lambda: planes(d, False))]
            t1 = time.perf_counter()
            for t_ in th: t_.start()
            for t_ in th: t_.join()
            return (time.perf_counter() - t1) * 1e3
        for i in range(3):                                        # untimed: first-call allocations, code page-in
            points(lat_g[i], False); lines(lat_g[i], lat_d[i], False, i); planes(lat_d[i], False); frame_threads(lat_g[i], lat_d[i], i)
            opt1.PoseOptimization(pb1, 4, 10)
        for i in range(len(lat_g)):
            points(lat_g[i], True); lines(lat_g[i], lat_d[i], True, i); planes(lat_d[i], True)
            tms["frame_3_threads"].append(frame_threads(lat_g[i], lat_d[i], i))
            if i < args.latency_reps:
                timed("pose_opt_4x10", lambda: opt1.PoseOptimization(pb1, 4, 10), True)
        pct = lambda v: {"p50": round(float(np.percentile(v, 50)), 3), "p99": round(float(np.percentile(v, 99)), 3), "max": round(float(np.max(v)), 3)}
        latency = {k: pct(v) for k, v in tms.items() if v}
        kk = np.array(kinds)
        for nm in ("peac_segment", "plane_clouds_refit", "lsd_lbd_extract", "frame_3_threads"):
            for kind in ("se3", "pan"):
                sel = np.array(tms[nm])[kk == kind]
                if len(sel): latency[nm][kind + "_p50"] = round(float(np.percentile(sel, 50)), 3); latency[nm][kind + "_max"] = round(float(sel.max()), 3)
        cp = np.array(tms["peac_segment"]) + np.array(tms["plane_clouds_refit"]) + np.array(tms["surface_normals"])
        latency["compute_planes(peac+clouds+normals)"] = pct(cp)
        latency["extract_lines(lsd+lbd+is_line_good)"] = pct(np.array(tms["lsd_lbd_extract"]) + np.array(tms["is_line_good"]))
        # back-compatible scalars (medians): what earlier rounds' lines carried
        for nm in ("orb_extract", "lsd_lbd_extract", "peac_segment", "pose_opt_4x10"):
            latency[nm + "_p50"] = latency[nm]["p50"]
        latency.update({"frames": len(lat_g), "frames_se3": int((kk == "se3").sum()), "frames_pan": int((kk == "pan").sum()),
                        "note": "wall ms per call, ONE frame, host buffers in and out, every frame distinct and timed once; frame_3_threads = Frame::Frame's three extractor threads "
                                "(src/Frame.cc:90-95): points | lines + isLineGood | PEAC + voxel clouds / refit + normals, each on its own host thread, context and stream"})

    # ---- sub_benchmarks: BASELINE configs[1] (ORB only), configs[3] (pose-only LM, B = 256) and configs[4] (local BA) as short sub-runs of THIS command, each with its
    #      own roofline and cpu_baseline, so that the driver's default invocation carries a number for every BASELINE config that fits one GPU ----
    sub = None
    if full and world == 1 and args.sub_steps > 0:
        sub = {}
        cpu_s = min(args.cpu_seconds, 6.0)
        # configs[1]: the pipeline's own extractor on the resident streams, window copy included (== `--workload orb`)
        with torch.cuda.stream(stream):
            ex.set_profiling(True)
            for i in range(3):
                window(i, frames[i % NB], None); ex.extract_dev(frames[i % NB].data_ptr(), tp.kps[i % NB].data_ptr(), tp.desc[i % NB].data_ptr(), tp.n[i % NB].data_ptr(), B)
            torch.cuda.synchronize()
            ex.get_profile()
            t1 = time.perf_counter()
            for i in range(args.sub_steps):
                window(3 + i, frames[i % NB], None); ex.extract_dev(frames[i % NB].data_ptr(), tp.kps[i % NB].data_ptr(), tp.desc[i % NB].data_ptr(), tp.n[i % NB].data_ptr(), B)
            torch.cuda.synchronize()
            dt_orb = time.perf_counter() - t1
            oprof, ocalls = ex.get_profile()
            ex.set_profiling(False)
        okp = float(tp.n[(args.sub_steps - 1) % NB].float().mean().item())
        oalg = orb_algorithmic_bytes(ex, okp)
        odom = max(oprof, key=lambda k: oprof[k][0])
        o_ms = oprof[odom][0] / max(1, oprof[odom][1])
        sub["orb"] = {"metric": "ORB frames/sec (640x480 gray, 8-level pyramid, 1000 keypoints + 256-bit rBRIEF)", "value": round(B * args.sub_steps / dt_orb, 1), "unit": "frames/s",
                      "steps": args.sub_steps, "ms_per_step": round(dt_orb / args.sub_steps * 1e3, 3), "dtype": "u8",
                      "config": {"workload": "BASELINE configs[1]: ORB only, the se3 streams' frames, window copy inside the timed loop", "frames_per_step": B, "avg_keypoints_per_frame": round(okp, 1)},
                      "roofline": {"bound": "hbm", "kernel": odom, "achieved": round(oalg.get(odom, 0) * B / (o_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(oalg.get(odom, 0) * B / (o_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(o_ms, 4),
                                   "kernels_ms_per_step": {k: round(v[0] / max(1, ocalls), 3) for k, v in oprof.items()}}}
        if cpu_s > 0:
            r_o = cb.run(cpu_s / 2, seed=rank, full=False)
            sub["orb"]["cpu_baseline"] = {"value": round(r_o["frames"] / r_o["seconds"], 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                          "sample": "%d frames through oracle/orb_oracle.cpp (pinned to the real ORBextractor.cc), one thread" % r_o["frames"]}
        sub["pose"] = pose_line(args.sub_steps, 3, cpu_s, 0, ranks, local_rank, dev)
        sub["ba"] = ba_line(max(3, args.sub_steps // 4), 2, cpu_s / 2, args.backend, ranks, local_rank, dev)

    workload = ("configs[2]+[3] as the reference's per-frame Track(): extract (ORB + LSD/LBD lines + isLineGood + PEAC planes + voxel clouds / RANSAC refit + surface normals on three streams, ComputeStereoFromRGBD) -> TrackManhattanFrame -> "
                "SearchByProjection(Cur, Last) + LSD SearchByDescriptor + MatchORBPoints + PlaneMatcher -> TranslationOptimization 4x10 -> isInFrustum + SearchByProjection(map) + "
                "LSD SearchByProjection -> PoseOptimization 4x10 -> UnprojectStereo; pose problems assembled on the device from the matchers' outputs"
                if full else "configs[1]: ORB only, 640x480 gray, 8-level pyramid, 1000 keypoints + 256-bit rBRIEF")
    nyi = (["map maintenance: the local map is the previous two frames' keypoints, key-frame lines and map planes are fixed per stream"]
           if full else ["LSD/LBD lines", "PEAC planes", "matching", "pose optimisation"])
    out = {
        "metric": "RGB-D frames/sec (extract+match+pose-opt) @640x480; 1->8-GPU batch scaling",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8/f64" if full else "u8", "data": "synthetic",
        "config": {"workload": workload, "frames_per_gpu_per_step": B, "distinct_canvases_per_gpu": ncanv,
                   "streams": (f"se3: {B} cameras in {ncanv} textured box rooms, gray + depth ray-cast from the same pose, {args.loop}-frame constant-velocity sweeps (1.2 cm, 0.25 deg per frame) "
                               f"played forwards and backwards; resident in HBM, a strided device copy per step" if se3 else "pan: a 640x480 window moving <= 8 px per step over unrelated gray / depth canvases"),
                   "window": (f"every step shows every stream the next frame of its {args.loop}-frame loop (forwards, then backwards: {2 * args.loop - 2} steps per cycle)" if se3
                              else "every step is a new window position for every stream"),
                   "pipeline_depth": args.depth,
                   "extractor_sets": tp.NW if full else None, "device_memory": MEM if full else None,
                   "cu_partition": ({"sequential_kernels_on_cus": args.seq_cus, "which": args.seq_which, "stream": "per context" if args.seq_per_ctx else "shared"} if full and args.seq_cus else None),
                   "avg_keypoints_per_frame": round(avg_kp, 1), "input_generation_s": round(t_gen, 1),
                   "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()}, "not_yet_in_workload": nyi,
                   "parallelism": f"frame-sharded x{world}, no collective"},
        "roofline": roofline, "cpu_baseline": cpu, "parity_sample": parity, "sub_benchmarks": sub, "latency_b1_ms": latency, "value_pcie_inclusive": pcie,
        "env_overrides": ENV_OVERRIDES,
        "scaling_curve": "this line is one point (n_gpus above); the 1/2/4/8 curve exists only where the driver's SCALE record is not 'skipped'",
    }
    if quality:
        out["config"].update(quality)
    if full:
        tp.close()
    print(json.dumps(out))
    ranks.close()


def main_ba(args):
    """`--workload ba`: BASELINE configs[4] as its own command (see ba_line)."""
    import torch

    from planarslam_amd.dist import Ranks
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local_rank = pick_device(args, torch)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = Ranks(backend=args.backend, device=dev)
    line = ba_line(args.steps, args.warmup, args.cpu_seconds, args.backend, ranks, local_rank, dev)
    if line is not None:
        print(json.dumps(line))
    ranks.close()


def ba_line(steps, warmup, cpu_seconds, backend, ranks, local_rank, dev):
    """BASELINE configs[4] (0-based: the fifth entry, "Local BA"; DESIGN.md and earlier rounds also called it "config 5" counting from one): local bundle adjustment
    of 10 free key frames (+ 2 fixed ones that only observe) x 3000 point / line / plane features.  ONE problem; its landmarks (with all
    their edges) are partitioned over the ranks, every rank linearises its part, and the reduced camera system is all-reduced over RCCL twice
    per LM trial (planarslam_amd/csrc/ba.hip).  A step = one complete solve (optimize(5), outlier levels, optimize(10), erase flags) through the
    host-pointer entry point, so the upload of the graph is inside the timed region.  Total work is fixed: "scaling": "strong".  Returns the line on rank 0."""
    import types

    import numpy as np
    import torch

    from planarslam_amd import Communicator, Context, local_bundle_adjustment, shard_problem
    from planarslam_amd.synth import TUM3, ba_problem
    args = types.SimpleNamespace(steps=steps, warmup=warmup, cpu_seconds=cpu_seconds, backend=backend)
    rank, world = ranks.rank, ranks.world
    ctx = Context(local_rank)
    comm = None
    if world > 1 and args.backend == "nccl":
        box = [Communicator.unique_id() if rank == 0 else None]
        ranks.dist.broadcast_object_list(box, src=0, device=dev)
        comm = Communicator(ctx, box[0], world, rank)
    elif world > 1:                                               # gloo: the exchange of ba.hip goes through the hosted transport (ranks may share a GPU)
        from planarslam_amd import HostedCommunicator
        comm = HostedCommunicator.torch(ctx)
    prob = ba_problem(seed=99)                                    # 2400 points + 500 line end points + 100 planes, 10 keyframes (2 fixed)
    mine = shard_problem(prob, rank, world) if world > 1 else prob

    def barrier():
        torch.cuda.synchronize(); ranks.barrier(); torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = local_bundle_adjustment(mine, TUM3, ctx=ctx, comm=comm)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = local_bundle_adjustment(mine, TUM3, ctx=ctx, comm=comm)
    barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    K, L, E = len(prob["kf_fixed"]), len(prob["lm_type"]), len(prob["e_kf"])
    np_free = int((np.asarray(prob["kf_fixed"]) == 0).sum())
    K_pre, L_pre, E_pre, np_free_pre = K, L, E, np_free
    NP = 6 * np_free
    exch_a = np_free * 36 + NP + 2 + NP * NP + NP
    # algorithmic HBM bytes of one LM iteration with one trial: errors (pose 64 + landmark 32 + meas 32 + info 32 in, err 24 out) twice,
    # linearisation (the same inputs + err in, W 144 out per edge; Hll 72 + bl 24 out per landmark), Schur (W 144 per edge, Hll/bl 96 in, Dinv 72 out),
    # update (W 144 per edge, Dinv 72 + bl 24 + landmark 32 in, landmark 32 + backup 32 out)
    it_bytes = E * (2 * 184 + 184 + 24 + 144 + 144 + 144) + L * (96 + 96 + 72 + 128 + 64)
    line = {"metric": "local bundle adjustments/sec (10 keyframes x 3000 point/line/plane features, g2oAddition edges)", "value": round(args.steps / elapsed, 3),
            "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] (0-based; 'Local BA'): %d free + %d fixed key-frame vertices, 2400 map points + 250 lines (500 end-point vertices) + 100 planes "
                                   "= %d landmark vertices, %d edges; landmarks partitioned over ranks" % (np_free_pre, K_pre - np_free_pre, L_pre, E_pre),
                       "keyframes": K, "landmark_vertices": L, "edges": E, "parallelism": "landmark-partition x%d" % world,
                       "exchange": {"per_trial": "A: %d doubles (Hpp|bp|chi2|S|b) + B: 3 doubles (chi2, scale, stop)" % exch_a, "bytes_A": exch_a * 8,
                                    "transport": ("RCCL all-reduce" if args.backend == "nccl" else "hosted (gloo, host-staged)") if world > 1 else "none (single GPU)"}},
            "lm_iterations": int(res["lm_iters"]), "outlier_edges": int(res["e_outlier"].sum())}
    it_ms = elapsed / args.steps * 1e3 / max(1, int(res["lm_iters"]))
    line["roofline"] = {"bound": "hbm", "achieved": round(it_bytes / world / (it_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(it_bytes / world / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                        "note": "algorithmic bytes of one LM iteration on this rank / (wall time of the solve / LM iterations): the solve is ONE small problem, "
                                "latency-bound by ~12 dependent launches + 2 exchanges per trial, not by HBM"}
    if rank == 0 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol                                  # test infrastructure, used here only as the timed CPU baseline
        n, t1 = 0, time.perf_counter()
        while n < 2 or time.perf_counter() - t1 < min(args.cpu_seconds, 10.0):
            ol.local_ba(prob, TUM3); n += 1
        dt = (time.perf_counter() - t1) / n
        line["cpu_baseline"] = {"value": round(1.0 / dt, 3), "unit": "solves/s", "cores": 1, "kind": "port",
                                "sample": "%d solves of the same problem by oracle/ba_oracle.cpp (dense Schur + Cholesky restatement of g2o's LM), one thread" % n}
    if comm is not None:
        comm.close()
    return line if rank == 0 else None


_POSE_CPU_BATCH = None


def _pose_cpu_chunk(a):
    """worker of pose_line's all-cores CPU leg: `count` problems of the inherited batch from `lo` on (wrapping) through oracle/pose_oracle.cpp"""
    lo, count = a
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    from planarslam_amd.synth import TUM3
    nb = len(_POSE_CPU_BATCH["n_points"])
    idx = (lo + np.arange(count)) % nb
    sl = {k: np.ascontiguousarray(v[idx]) for k, v in _POSE_CPU_BATCH.items()}
    ol.pose_optimize(sl, TUM3, 0, 4, 10)
    return count


def main_pose(args):
    """`--workload pose`: BASELINE configs[3] as its own command (see pose_line)."""
    import torch

    from planarslam_amd.dist import Ranks
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local_rank = pick_device(args, torch)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = Ranks(backend=args.backend, device=dev)
    line = pose_line(args.steps, args.warmup, args.cpu_seconds, args.batch, ranks, local_rank, dev)
    if line is not None:
        print(json.dumps(line))
    ranks.close()


def pose_line(steps, warmup, cpu_seconds, batch, ranks, local_rank, dev):
    """BASELINE configs[3]: pose-only Levenberg-Marquardt, 1000 point + 150 line (75 lines x 2 end points) + 12 plane (4 planes x plane / parallel / vertical)
    edges per frame, batch = 256 frames per GPU (synth.pose_batch(B = 256, seed = 7 + 1000 * rank)), resident in HBM.  A step = Optimizer::PoseOptimization
    (src/Optimizer.cc:550-1275: four rounds of optimize(10) with outlier reclassification) for every frame of the batch: one launch of pose_opt_kernel.  Frames are
    independent: every rank has its own batch, no collective ("weak").  Returns the line on rank 0."""
    import ctypes as C
    import types

    import numpy as np
    import torch

    from planarslam_amd import Context
    from planarslam_amd._lib import PoseBatch, check, lib
    from planarslam_amd.optimizer import make_params
    from planarslam_amd.synth import TUM3, pose_batch
    args = types.SimpleNamespace(steps=steps, warmup=warmup, cpu_seconds=cpu_seconds, batch=batch)
    rank, world = ranks.rank, ranks.world
    B = args.batch or 256                                         # BASELINE configs[3] is quoted on 256
    host = pose_batch(B=B, n_points=1000, n_lines=75, n_planes=4, seed=7 + 1000 * rank)
    L = lib()
    stream = torch.cuda.Stream(device=dev)
    ctx = Context(local_rank, stream=stream.cuda_stream)
    params = make_params(TUM3)
    MP, ML, MM = host["pt_valid"].shape[1], host["ln_valid"].shape[1], host["pl_valid"].shape[1]
    keys_in = ("n_points", "n_lines", "n_planes", "pt_valid", "pt_xw", "pt_obs", "pt_inv_sigma2", "ln_valid", "ln_obs", "ln_xw", "pl_meas", "pl_valid", "pl_world")
    d = {k: torch.from_numpy(np.ascontiguousarray(host[k])).to(dev) for k in keys_in}
    d["Tcw_in"] = torch.from_numpy(np.ascontiguousarray(host["Tcw"], np.float32)).to(dev)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    d.update(Tcw_out=z((B, 16), torch.float32), pt_outlier=z((B, MP), torch.uint8), ln_outlier=z((B, ML), torch.uint8), pl_outlier=z((B, MM, 3), torch.uint8),
             n_inliers=z((B,), torch.int32), lm_iters=z((B,), torch.int32))
    pb = PoseBatch()
    pb.B, pb.max_points, pb.max_lines, pb.max_planes = B, MP, ML, MM
    for k in keys_in + ("Tcw_in", "Tcw_out", "pt_outlier", "ln_outlier", "pl_outlier", "n_inliers", "lm_iters"):
        setattr(pb, k, d[k].data_ptr())

    def barrier():
        torch.cuda.synchronize(); ranks.barrier(); torch.cuda.synchronize()

    def run(rounds, its, steps, warmup):
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                check(L.planar_pose_opt_dev(ctx.h, C.byref(pb), C.byref(params), 0, rounds, its))
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(steps):
                check(L.planar_pose_opt_dev(ctx.h, C.byref(pb), C.byref(params), 0, rounds, its))
            e1.record(stream)
            barrier()
            return ranks.max_over_ranks(time.perf_counter() - t0), e0.elapsed_time(e1) / steps, float(d["lm_iters"].float().mean().item())

    el_1, k_1, it_1 = run(1, 10, args.steps, args.warmup)        # one optimize(10): the "10 iters" protocol of BASELINE.md
    el, k_ms, lm_it = run(4, 10, args.steps, args.warmup)         # the reference's PoseOptimization: 4 x optimize(10)
    if rank != 0:
        return None
    # parity of this very batch with the real optimiser is tests/test_pose_gpu.py::test_pose_hip_equals_reference_fixture (c4_b256)
    alg = 65130 * B * lm_it                                       # bytes of one LM evaluation of one frame x measured LM iterations per frame
    line = {"metric": "pose-only LM problems/sec (1000 point + 150 line + 12 plane edges, 4 x 10 iterations, batch 256 per GPU)", "value": round(B * world * args.steps / el, 1),
            "unit": "problems/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3] (0-based): pose-only LM, synth.pose_batch(B=%d, n_points=1000, n_lines=75 (150 end-point edges), n_planes=4 (12 plane / parallel / "
                                   "vertical edges), seed=7), Optimizer::PoseOptimization = 4 rounds of optimize(10) with outlier reclassification; inputs resident in HBM" % B,
                       "frames_per_gpu_per_step": B, "avg_lm_iterations_per_frame": round(lm_it, 2),
                       "protocol_1x10": {"problems_per_s": round(B * world * args.steps / el_1, 1), "ms_per_step": round(el_1 / args.steps * 1e3, 4), "avg_lm_iterations_per_frame": round(it_1, 2)}},
            "roofline": {"bound": "hbm", "kernel": "pose_opt_kernel", "achieved": round(alg / (k_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": int(alg),
                         "note": "65 130 B per frame per LM evaluation (SURVEY §8d) x the measured LM iterations; HIP events on the kernel's stream around the timed launches.  One workgroup per "
                                 "frame runs the whole protocol on chip in FP64: 256 frames = 256 workgroups = one per CU, latency-bound (DESIGN.md §4)"}}
    if args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol                                  # test infrastructure, used here only as the timed CPU baseline
        nb = min(B, 32)
        sl = {k: np.ascontiguousarray(v[:nb]) for k, v in host.items() if k != "T_gt"}
        n, t1 = 0, time.perf_counter()
        while n < 1 or time.perf_counter() - t1 < min(args.cpu_seconds, 18.0) / 3:
            ol.pose_optimize(sl, TUM3, 0, 4, 10); n += 1
        one = nb * n / (time.perf_counter() - t1)
        line["cpu_baseline"] = {"value": round(one, 2), "unit": "problems/s", "cores": 1, "kind": "port",
                                "sample": "%d x the first %d problems of the batch through oracle/pose_oracle.cpp (g2o's LM restated, pinned to the real Optimizer.cc: tests/test_oracle_opt_ref.py), one thread" % (n, nb)}
        try:
            # all host cores: a persistent pool of forked workers (the batch and the warm oracle import are inherited, nothing is pickled but two integers), every
            # worker solves 32 problems of the batch per task; one untimed task per worker first (page-in, first call), then the timed map
            import multiprocessing as mp
            ncores = os.cpu_count() or 1
            global _POSE_CPU_BATCH
            _POSE_CPU_BATCH = {k: np.ascontiguousarray(v) for k, v in host.items() if k != "T_gt"}
            per = 128
            jobs = [((w * per) % B, per) for w in range(ncores)]
            with mp.get_context("fork").Pool(ncores) as pool:
                pool.map(_pose_cpu_chunk, [(j[0], 2) for j in jobs], chunksize=1)
                t1 = time.perf_counter()
                pool.map(_pose_cpu_chunk, jobs, chunksize=1)
                dt_all = time.perf_counter() - t1
            line["cpu_baseline"]["all_cores"] = {"value": round(per * ncores / dt_all, 1), "unit": "problems/s", "cores": ncores,
                                                 "sample": "%d forked workers (persistent pool, oracle imported and batch inherited before the fork, one untimed warm-up task each), "
                                                           "each solving %d problems of the batch" % (ncores, per)}
        except Exception as e:                                    # (a box without fork / enough memory: the one-thread figure stands)
            line["cpu_baseline"]["all_cores"] = {"error": type(e).__name__}
    return line


if __name__ == "__main__":
    main()
