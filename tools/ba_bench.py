#!/usr/bin/env python3
"""BASELINE config 5 timing: local BA, 10 keyframes x ~3000 features, HIP vs the CPU oracle (1 thread)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from planarslam_amd import Context, local_bundle_adjustment
from planarslam_amd.synth import TUM3, ba_problem
import oracle_lib as ol
pr = ba_problem(seed=99)
ctx = Context(0)
local_bundle_adjustment(pr, TUM3, ctx=ctx)
t = time.perf_counter(); n = 5
for _ in range(n): r = local_bundle_adjustment(pr, TUM3, ctx=ctx)
dt = (time.perf_counter() - t) / n
t = time.perf_counter(); o = ol.local_ba(pr, TUM3); dc = time.perf_counter() - t
print(f"config5 BA: edges {len(pr['e_kf'])} landmarks {len(pr['lm_type'])} | HIP {dt*1e3:.1f} ms/solve ({r['lm_iters']} LM iterations, {r['lm_iters']/dt:.0f} it/s) | "
      f"oracle {dc*1e3:.0f} ms ({o['lm_iters']} it) | all-reduce payload {(60*60+60)*8} B per trial")
