"""world_size-2 gloo test (CPU) of the multi-GPU plumbing bench.py uses: disjoint frame shards, barrier, max-over-ranks."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, %r)
    from planarslam_amd.dist import Ranks, whole_job_fps
    r = Ranks(backend="gloo")
    ids = list(r.frame_ids(8))
    r.barrier()
    elapsed = 0.25 if r.rank == 0 else 0.5            # rank 1 is the slow one
    mx = r.max_over_ranks(elapsed)
    total = r.sum_over_ranks(len(ids))
    print(json.dumps({"rank": r.rank, "world": r.world, "ids": ids, "max": mx, "total": total, "fps": whole_job_fps(r.world, 8, 4, mx)}))
    r.close()
''') % ROOT


def test_two_ranks_shard_frames_and_take_the_max_time(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["ids"] == list(range(0, 8)) and outs[1]["ids"] == list(range(8, 16))      # disjoint shards, no overlap
    for d in outs:
        assert d["world"] == 2 and d["max"] == 0.5 and d["total"] == 16
        assert abs(d["fps"] - 2 * 8 * 4 / 0.5) < 1e-9
