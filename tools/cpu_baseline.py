"""CPU baseline of bench.py: the oracle/ restatements of the hot-path stages, timed on the host cores of the box the bench runs on.

TEST / MEASUREMENT INFRASTRUCTURE: this is the only place besides tests/ and __graft_entry__.smoke() that imports oracle/.  The same
per-frame work list as bench.py's GPU step (the reference's Track()): ORB + LSD/LBD + PEAC extraction, ComputeStereoFromRGBD, TrackManhattanFrame,
SearchByProjection(Cur, Last), MatchORBPoints, line SearchByDescriptor, PlaneMatcher, isInFrustum + SearchByProjection(map), TranslationOptimization
and PoseOptimization 4x10 on a config-4-shaped problem.  Three variants:
  one_thread   everything on one thread
  threads3     extraction on three threads per frame as src/Frame.cc:90-95 does (ORB / LSD / PEAC), matching and LM on the calling thread
  all_cores    `os.cpu_count()` independent worker processes, each running the one-thread loop on its own frames (frames shard trivially)
Run as a script it is one all_cores worker:  python tools/cpu_baseline.py --seconds 8 --seed 3   -> one JSON line."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H = 640, 480
NPTS, NLINES, NPLANES = 1000, 75, 4
FAST_LIB = os.path.join(ROOT, "oracle", "liboracle_fast.so")
_BUILD = {"flags": "-O2 -ffp-contract=off (oracle/liboracle.so, the checker's build)", "lib": "oracle/liboracle.so"}


def use_fast_build():
    """Builds oracle/liboracle_fast.so with -O3 -march=native ON THIS MACHINE (SURVEY.md §8d's flags for the CPU leg) and makes oracle_lib load it in this
    process and its workers.  Falls back to the checker's -O2 build (and says so in the line) when there is no compiler."""
    import subprocess
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        os.environ["PLANAR_ORACLE_LIB"] = FAST_LIB
        cxx = subprocess.run(["g++", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
        _BUILD.update(flags="-O3 -march=native -ffp-contract=off", lib="oracle/liboracle_fast.so (built on this box)", compiler=cxx)
    except Exception as e:   # noqa: BLE001
        _BUILD["note"] = f"liboracle_fast.so could not be built here ({type(e).__name__}): timed on the -O2 build"
    return dict(_BUILD)


def build_flags():
    return dict(_BUILD)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _inputs(seed, nsrc):
    import numpy as np
    from planarslam_amd.synth import depth_image, gray_image, pose_batch
    gray = np.stack([gray_image(1234 + seed * 1000 + i) for i in range(nsrc)])
    depth = np.stack([depth_image(4321 + seed * 1000 + i) for i in range(nsrc)])
    pbn = pose_batch(B=nsrc, n_points=NPTS, n_lines=NLINES, n_planes=NPLANES, seed=7 + seed)
    return gray, depth, pbn


def run(seconds, seed=0, nsrc=4, threads3=False, full=True):
    """Process frames for about `seconds`; returns dict(frames, seconds, per-stage seconds)."""
    import numpy as np
    import oracle_lib as ol
    from planarslam_amd.synth import TUM3, scale_factors
    gray, depth, pbn = _inputs(seed, nsrc)
    o = ol.OrbOracle()
    sfs = scale_factors()
    rng = np.random.default_rng(11 + seed)
    per = {"orb": 0.0, "lsd": 0.0, "lines3d": 0.0, "peac": 0.0, "planepost": 0.0, "normals": 0.0, "extract_wall": 0.0, "stereo": 0.0, "manhattan": 0.0, "proj": 0.0, "match": 0.0, "planes": 0.0, "local": 0.0, "pose": 0.0}
    from planarslam_amd.synth import manhattan_scene
    scene = manhattan_scene(B=1, n_normals=4096, n_lines=40, seed=21 + seed)
    prev = None
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        i = n % nsrc
        res = {}

        def t_orb():
            t1 = time.perf_counter(); res["orb"] = o.extract(gray[i]); per["orb"] += time.perf_counter() - t1

        def t_lsd():
            t1 = time.perf_counter(); res["lsd"] = ol.extract_line_segment(gray[i], tie_order=0); per["lsd"] += time.perf_counter() - t1
            t1 = time.perf_counter(); res["l3"] = ol.is_line_good(res["lsd"][0], depth[i], seed=n * 64); per["lines3d"] += time.perf_counter() - t1      # Frame::ExtractLSD: same thread

        def t_peac():
            t1 = time.perf_counter(); res["peac"] = ol.peac_run(depth[i]); per["peac"] += time.perf_counter() - t1
            t1 = time.perf_counter(); res["clouds"] = ol.plane_clouds(depth[i], res["peac"][1], res["peac"][0]); per["planepost"] += time.perf_counter() - t1   # voxel clouds + refit
            t1 = time.perf_counter(); res["normals"] = ol.surface_normals(depth[i])[0]; per["normals"] += time.perf_counter() - t1     # Frame::ComputePlanes tail, same thread

        tw = time.perf_counter()
        if not full:
            t_orb()
        elif threads3:          # ctypes releases the GIL inside the oracle calls: the three extractors really run side by side
            th = [threading.Thread(target=f) for f in (t_lsd, t_peac)]
            for t in th: t.start()
            t_orb()
            for t in th: t.join()
        else:
            t_orb(); t_lsd(); t_peac()
        per["extract_wall"] += time.perf_counter() - tw
        kp, de = res["orb"]
        if full:
            nj = len(kp)
            eye = np.eye(4, dtype=np.float32).reshape(1, 16)
            tm = time.perf_counter
            t1 = tm(); st = ol.stereo_from_rgbd(kp, depth[i], eye[0], TUM3); per["stereo"] += tm() - t1
            t1 = tm(); ol.track_manhattan_frame(scene["R_last"][0], res["normals"], res["l3"]["direction"][res["l3"]["good"] > 0]); per["manhattan"] += tm() - t1
            kl, ld, leq = res["lsd"][0], res["lsd"][1], res["lsd"][2]
            planes = res["peac"][0]
            if prev is not None:
                pkp, pde, pst, pld, ppl = prev
                npv = len(pkp)
                curv = dict(n=np.array([nj], np.int32), keys_un=kp.reshape(1, nj), u_right=st["u_right"].reshape(1, nj), desc=de.reshape(1, nj, 32), Tcw=eye,
                            min_x=0.0, max_x=float(W), min_y=0.0, max_y=float(H), fx=TUM3["fx"], fy=TUM3["fy"], cx=TUM3["cx"], cy=TUM3["cy"], bf=TUM3["bf"],
                            b=TUM3["bf"] / TUM3["fx"], scale_factors=sfs)
                lastv = dict(n=np.array([npv], np.int32), Tcw=eye, usable=pst["valid"].reshape(1, npv), xw=pst["xw"].reshape(1, npv, 3),
                             octave=np.ascontiguousarray(pkp["octave"]).reshape(1, npv), angle=np.ascontiguousarray(pkp["angle"]).reshape(1, npv),
                             mp_desc=pde.reshape(1, npv, 32), mp_observed=np.ones((1, npv), np.uint8))
                t1 = tm(); pm, _ = ol.search_by_projection_frame(curv, lastv, 15.0); per["proj"] += tm() - t1
                t1 = tm()
                ol.match_orb_points(de, pde, np.ones(npv, np.uint8), np.zeros(npv, np.uint8), np.full(nj, -1, np.int32))
                if len(pld) and len(ld): ol.lsd_search_by_descriptor(pld, ld, np.ones(len(pld), np.uint8))
                per["match"] += tm() - t1
                if len(planes) and len(ppl):
                    t1 = tm()
                    coef = res["clouds"]["coef"][None] if res["clouds"]["n"] else np.zeros((1, 1, 4), np.float32)
                    mcoef = np.concatenate([ppl[:, 1:4], -(ppl[:, 1:4] * ppl[:, 4:7]).sum(1, keepdims=True)], 1).astype(np.float32)[None]
                    mpts = np.repeat(ppl[:, 4:7].astype(np.float32)[None, :, None, :], 64, 2)
                    ol.plane_search_by_coefficients(dict(n=np.array([res["clouds"]["n"]], np.int32), coef=coef, Tcw=eye),
                                                    dict(n=np.array([len(ppl)], np.int32), valid=np.ones((1, len(ppl)), np.uint8), coef=mcoef,
                                                         npts=np.full((1, len(ppl)), 64, np.int32), pts=mpts))
                    per["planes"] += tm() - t1
                # the local map of this frame: the previous frame's points once more (the same count of probes as the GPU step's older generation)
                t1 = tm()
                dist = np.linalg.norm(pst["xw"], axis=1).astype(np.float32) + 1e-6
                mpd = dict(n=np.array([npv], np.int32), valid=pst["valid"].reshape(1, npv), xw=pst["xw"].reshape(1, npv, 3), normal=(pst["xw"] / dist[:, None]).reshape(1, npv, 3),
                           min_dist=(dist * 0.3).reshape(1, npv), max_dist=(dist * 3.0).reshape(1, npv))
                curv2 = dict(curv, blocked=(pm >= 0).astype(np.uint8))
                pr = ol.is_in_frustum_points(curv2, mpd, float(np.float32(np.log(np.float32(1.2)))), 8, 0.5)
                ol.search_by_projection_map(curv2, dict(pr, n=mpd["n"], desc=pde.reshape(1, npv, 32), observed=np.ones((1, npv), np.uint8)), 3.0, 0.8)
                per["local"] += tm() - t1
            one = {k: (v[i:i + 1] if isinstance(v, np.ndarray) and v.shape[:1] == (nsrc,) else v) for k, v in pbn.items()}
            t1 = tm(); ol.pose_optimize(one, TUM3, 1, 4, 10); ol.pose_optimize(one, TUM3, 0, 4, 10); per["pose"] += tm() - t1
            prev = (kp, de, st, ld, planes)
        n += 1
    dt = time.perf_counter() - t0
    return dict(frames=n, seconds=dt, per=per)


def all_cores(seconds, workers=None, full=True):
    """`workers` processes of this script, one per host core; returns (frames/s summed over workers, workers, details)."""
    import subprocess
    workers = workers or os.cpu_count()
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--seconds", str(seconds), "--seed", str(100 + w)] + ([] if full else ["--orb-only"]),
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for w in range(workers)]
    outs = []
    for p in procs:
        so, _ = p.communicate()
        try:
            outs.append(json.loads(so.decode().strip().splitlines()[-1]))
        except Exception:
            pass
    fps = sum(o["frames"] / o["seconds"] for o in outs)
    return fps, len(outs), outs


def summarize(r):
    n = max(1, r["frames"])
    return {k: round(v / n * 1e3, 2) for k, v in r["per"].items() if v > 0}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--orb-only", action="store_true")
    a = ap.parse_args()
    r = run(a.seconds, a.seed, nsrc=2, full=not a.orb_only)
    print(json.dumps(dict(frames=r["frames"], seconds=r["seconds"])))
