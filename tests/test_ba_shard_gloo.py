"""world_size-2 gloo test (CPU) of the BA exchange step: landmarks (with all their edges) are partitioned over ranks
(planarslam_amd.ba.shard_problem), every rank builds ITS part of the reduced camera system, and the all-reduce (sum) —
RCCL on the GPUs, gloo here — must give the unsharded system.  The per-shard systems come from the oracle (test
infrastructure); the product path does the same sum in planar_local_ba."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as ol
    from planarslam_amd.ba import shard_problem
    from planarslam_amd.synth import TUM3, ba_problem
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prob = ba_problem(seed=5, n_points=200, n_lines=40, n_planes=10)
    sh = shard_problem(prob, rank, world)
    S, b, chi = ol.ba_reduced_system(sh, TUM3, 1e-3)
    buf = torch.from_numpy(np.concatenate([S.ravel(), b, [chi, len(sh["lm_ids"]), len(sh["e_ids"])]]))
    dist.all_reduce(buf)                                  # the 29 KB exchange (SURVEY.md §8e)
    Sf, bf, chif = ol.ba_reduced_system(prob, TUM3, 1e-3)
    n = S.size
    out = dict(rank=rank, nlm=int(len(sh["lm_ids"])), ne=int(len(sh["e_ids"])), tot_lm=float(buf[-2]), tot_e=float(buf[-1]),
               dS=float(np.abs(buf[:n].numpy().reshape(S.shape) - Sf).max() / np.abs(Sf).max()),
               db=float(np.abs(buf[n:n + len(b)].numpy() - bf).max() / np.abs(bf).max()), dchi=abs(float(buf[-3]) - chif) / chif,
               full_lm=int(len(prob["lm_type"])), full_e=int(len(prob["e_kf"])), payload_bytes=int((n + len(b)) * 8))
    print(json.dumps(out))
    dist.destroy_process_group()
''') % (ROOT, ROOT)


def test_sharded_reduced_system_sums_to_the_full_one(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e
        outs.append(json.loads(o.strip().splitlines()[-1]))
    for d in outs:
        assert d["tot_lm"] == d["full_lm"] and d["tot_e"] == d["full_e"]          # a partition: nothing lost, nothing duplicated
        assert d["dS"] < 1e-12 and d["db"] < 1e-12 and d["dchi"] < 1e-12
        assert 0 < d["nlm"] < d["full_lm"]


def test_shard_keeps_line_endpoint_pairs_together():
    sys.path.insert(0, ROOT)
    from planarslam_amd.ba import shard_problem
    from planarslam_amd.synth import ba_problem
    prob = ba_problem(seed=6, n_points=50, n_lines=60, n_planes=5)
    seen = np.zeros(len(prob["e_kf"]), int)
    for r in range(3):
        sh = shard_problem(prob, r, 3)
        seen[sh["e_ids"]] += 1
        idx = np.nonzero(sh["e_type"] == 2)[0]
        assert len(idx) % 2 == 0 and (np.diff(idx)[0::2] == 1).all()             # (start, end) edges stay consecutive
        assert (sh["e_lm"] >= 0).all() and sh["e_lm"].max() < len(sh["lm_ids"])
    assert (seen == 1).all()
