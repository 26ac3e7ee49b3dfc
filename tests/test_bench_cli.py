"""bench.py's command line, the parts that need no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_bench_refuses_environment_switches():
    """No timing / variant switch reaches the timed region through the environment: bench.py exits non-zero when a PLANAR_* variable other than the developer-library
    override is set (round 4's PLANAR_TRACK_SKIP / PLANAR_PEAC_AHC / PLANAR_PEAC_WIDE no longer exist in the product either: planarslam_amd/ reads none of them)."""
    env = dict(os.environ, PLANAR_TRACK_SKIP="lsd")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "refusing to run" in (p.stderr + p.stdout)


def test_the_product_reads_no_timing_switch():
    import re
    pat = re.compile(r"getenv|os\.environ")
    hits = []
    for base, _, files in os.walk(os.path.join(ROOT, "planarslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                for n, line in enumerate(open(os.path.join(base, f), errors="replace"), 1):
                    if pat.search(line) and not re.search(r"PLANAR_HIP_LIB|RANK|WORLD_SIZE|MASTER_ADDR", line):
                        hits.append(f"{f}:{n}: {line.strip()}")
    assert not hits, hits
