"""`python bench.py --gpus 2` end to end on the 1-GPU test box: the self-launch (torch.distributed.run on 127.0.0.1), the rendezvous, the frame / landmark
sharding, the barrier + max-over-ranks timing and rank 0's single JSON line.  RCCL refuses two ranks on one device, so the ranks use --backend gloo
and share GPU 0 (the BA workload then exchanges its reduced system through the hosted transport); on the driver's 8-GPU node the same code runs with
the default --backend nccl, one rank per GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}: {p.stdout[-2000:]}"
    return json.loads(lines[0])


def test_two_ranks_shard_frames_and_print_one_line():
    B, K = 64, 2
    d = _bench("--gpus", "2", "--backend", "gloo", "--steps", str(K), "--warmup", "1", "--batch", str(B), "--canvases", "16", "--cpu-seconds", "0", "--latency-reps", "0",
               "--pcie-steps", "0")
    assert d["n_gpus"] == 2 and d["steps"] == K and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert d["config"]["frames_per_gpu_per_step"] == B and d["config"]["parallelism"].startswith("frame-sharded x2")
    # whole-job value: the frames of BOTH ranks over the slowest rank's time
    assert abs(d["value"] - 2 * B / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["roofline"]["frac"] > 0 and d["config"]["avg_keypoints_per_frame"] > 100


def test_two_ranks_partition_the_bundle_adjustment():
    d = _bench("--workload", "ba", "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "solves/s" and d["value"] > 0


def test_two_ranks_each_solve_their_own_pose_batch():
    """BASELINE configs[3] at N = 2: every rank has its own batch of 256 pose problems (seed 7 + 1000 * rank), no collective on the data path; rank 0 prints the
    whole-job rate = both ranks' problems over the slowest rank's time."""
    K = 3
    d = _bench("--workload", "pose", "--gpus", "2", "--backend", "gloo", "--steps", str(K), "--warmup", "1", "--cpu-seconds", "0")
    assert d["n_gpus"] == 2 and d["steps"] == K and d["scaling"] == "weak" and d["unit"] == "problems/s"
    assert d["config"]["frames_per_gpu_per_step"] == 256
    assert abs(d["value"] - 2 * 256 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["roofline"]["kernel"] == "pose_opt_kernel" and d["roofline"]["frac"] > 0

