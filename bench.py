#!/usr/bin/env python3
"""bench.py — throughput of the MI355X hot path (BASELINE.json metric: RGB-D frames/sec @640x480).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of `--batch` synthetic 640x480 frames that are
already resident in HBM.  Frames are independent, so ranks shard them with no data-path collective
("scaling": "weak": every rank processes its own `--batch` frames per step).  Rank 0 prints ONE JSON
line carrying the whole-job frames/s, the roofline of the dominant kernel (HIP events recorded on the
stream the kernels run on, inside the timed region) and a CPU baseline (the oracle restatement of
the reference path, timed on this box's host cores on a bounded sample).

What the workload covers is named in config.workload; stages not yet built are listed in
config.not_yet_in_workload — the number is NOT the full extract+match+pose-opt rate yet.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
W, H = 640, 480


def orb_algorithmic_bytes(ex):
    """Algorithmic HBM bytes per frame per kernel launch: each input read once,
    each output written once, nothing for data that could stay on chip."""
    sizes = [ex.level_size(l) for l in range(ex.nlevels)]
    px = [w * h for w, h in sizes]
    kp = 1000
    per = {
        "orb_copy_level0": 2 * px[0],
        # average over the nlevels-1 launches: read level l-1, write level l
        "orb_resize": sum(px[l - 1] + px[l] for l in range(1, ex.nlevels)) / max(1, ex.nlevels - 1),
        "orb_fast_cells": sum(px),                 # every level read once; survivors are a few KB
        "orb_sort": 0,                             # filled from the measured candidate count below
        "orb_octree": 0,
        "orb_blur": 2 * sum(px),
        "orb_describe": kp * (709 + 512 + 60),     # IC_Angle disc + 512 BRIEF taps + keypoint/descriptor out
    }
    return per, sum(px)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from planarslam_amd import Context, ORBextractor
    from planarslam_amd.synth import gray_image

    B = args.batch
    stream = torch.cuda.Stream(device=local_rank)
    ctx = Context(local_rank, stream=stream.cuda_stream)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, width=W, height=H, max_batch=B, ctx=ctx)

    # synthetic frames, distinct per rank; 16 distinct images tiled over the batch
    base = np.stack([gray_image(1234 + 16 * rank + i) for i in range(min(B, 16))])
    frames = torch.from_numpy(np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B]).cuda(local_rank)
    d_kps = torch.zeros((B, ex.kp_cap, 7), dtype=torch.float32, device=frames.device)
    d_desc = torch.zeros((B, ex.kp_cap, 32), dtype=torch.uint8, device=frames.device)
    d_n = torch.zeros(B, dtype=torch.int32, device=frames.device)

    def step():
        ex.extract_dev(frames.data_ptr(), d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), B)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        ex.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        prof, calls = ex.get_profile()
        ex.set_profiling(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=frames.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_kp = int(d_n.sum().item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    fps = world * B * args.steps / elapsed
    # ---- roofline of the dominant kernel (largest summed HIP-event time in the timed region) ----
    per, px_total = orb_algorithmic_bytes(ex)
    avg_kp = n_kp / B
    per["orb_describe"] = avg_kp * (709 + 512 + 60)
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms_total, dom_launches = prof[dom]
    avg_launch_ms = dom_ms_total / max(1, dom_launches)
    alg_bytes_launch = per.get(dom, 0) * B
    achieved = alg_bytes_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    kernels = {k: {"ms_per_step": round(v[0] / max(1, calls), 4), "launches_per_step": v[1] // max(1, calls)} for k, v in prof.items()}
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "avg_launch_ms": round(avg_launch_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes_launch),
                "pipeline_algorithmic_GBps": round(1961064 * fps / 1e9, 2), "kernels": kernels}

    # ---- CPU baseline: the oracle restatement of the same workload on this box's host cores ----
    cpu = None
    if args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        o = ol.OrbOracle()
        o.extract(base[0])
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.cpu_seconds:
            o.extract(base[n % len(base)])
            n += 1
        dt = time.perf_counter() - t0
        cpu = {"value": round(n / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{n} frames of the same synthetic set through oracle/orb_oracle.cpp (1 thread, {dt:.1f} s)",
               "host_cores": os.cpu_count()}

    out = {
        "metric": "RGB-D frames/sec (extract+match+pose-opt) @640x480; 1->8-GPU batch scaling",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: ORB only, 640x480 gray, 8-level pyramid, 1000 keypoints + 256-bit rBRIEF",
                   "frames_per_gpu_per_step": B, "avg_keypoints_per_frame": round(avg_kp, 1),
                   "not_yet_in_workload": ["LSD/LBD lines", "PEAC planes", "matching", "pose optimisation"],
                   "parallelism": f"frame-sharded x{world}, no collective"},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
