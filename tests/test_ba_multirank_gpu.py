"""The partitioned bundle adjustment of the PRODUCT path with more than one rank: landmarks (with all their edges) are
split over ranks (planarslam_amd.ba.shard_problem), every rank runs planar_local_ba on its shard, and the two exchanges per LM trial
(ba.hip header) make every rank solve the same reduced camera system.

  * hosted transport, 2 and 3 processes sharing GPU 0: the exchange goes through torch.distributed/gloo (planar_comm_create_hosted).  RCCL
    refuses two ranks on one device, so this is how the multi-rank control flow (device-side LM decisions taken from all-reduced values
    only, stop word agreed through the exchange) runs on the 1-GPU test box.
  * RCCL transport, one process per GPU: needs >= 2 GPUs; auto-skips on the 1-GPU box (the driver's 8-GPU node runs bench.py --workload ba).
Expected: every rank returns the same keyframe poses, equal to the single-GPU solve of the unsharded problem within 1e-9 (only the
summation order of the reduced system differs), identical erase flags, and the landmarks of its shard."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from planarslam_amd import Communicator, Context, HostedCommunicator, local_bundle_adjustment, shard_problem
    from planarslam_amd.synth import TUM3, ba_problem
    transport, stop_rank = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = rank if transport == "rccl" else 0
    ctx = Context(dev)
    if transport == "rccl":
        box = [Communicator.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = Communicator(ctx, box[0], world, rank)
    else:
        comm = HostedCommunicator.torch(ctx)
    prob = ba_problem(seed=5, n_points=300, n_lines=60, n_planes=12)
    sh = shard_problem(prob, rank, world)
    stop = None
    if stop_rank >= 0:                       # only ONE rank's caller raises the flag: the others must still leave the loop with it
        import ctypes
        stop = ctypes.c_ubyte(1 if rank == stop_rank else 0)
    got = local_bundle_adjustment(sh, TUM3, ctx=ctx, comm=comm, stop_flag=stop)
    full = local_bundle_adjustment(prob, TUM3, ctx=ctx) if stop_rank < 0 else None
    out = dict(rank=rank, kf=got["kf_Tcw"].astype(float).tolist(), lm_iters=int(got["lm_iters"]), stopped=int(got["stopped"]), nlm=len(sh["lm_ids"]))
    if full is not None:
        out.update(d_kf=float(np.abs(got["kf_Tcw"].astype(np.float64) - full["kf_Tcw"]).max()), d_lm=float(np.abs(got["lm"] - full["lm"][sh["lm_ids"]]).max()),
                   flags_equal=bool(np.array_equal(got["e_outlier"], full["e_outlier"][sh["e_ids"]])), full_iters=int(full["lm_iters"]), n_out=int(got["e_outlier"].sum()))
    print("RESULT " + json.dumps(out))
    comm.close()
    dist.destroy_process_group()
''') % ROOT


def _run(tmp_path, world, transport, stop_rank=-1):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), transport, str(stop_rank)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung: the ranks did not agree on the control flow")
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    return outs


def _check(outs):
    kf0 = np.array(outs[0]["kf"])
    for d in outs:
        assert np.array_equal(np.array(d["kf"]), kf0)              # every rank solved the same reduced system, bit for bit
        assert d["d_kf"] <= 1e-6 and d["d_lm"] <= 1e-4, d            # float32 poses; only the summation order differs (1e-4 m on badly observed depths, as in test_ba_gpu)
        assert d["flags_equal"] and d["n_out"] > 0
        assert d["lm_iters"] == d["full_iters"] and d["stopped"] == 0
        assert d["nlm"] > 0


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_ba_hosted_transport_processes_share_one_gpu(tmp_path, world):
    _check(_run(tmp_path, world, "hosted"))


def test_stop_flag_raised_on_one_rank_stops_all_ranks(tmp_path):
    outs = _run(tmp_path, 2, "hosted", stop_rank=1)
    assert all(d["stopped"] == 1 for d in outs) and len({d["lm_iters"] for d in outs}) == 1
    assert np.array_equal(np.array(outs[0]["kf"]), np.array(outs[1]["kf"]))


def test_partitioned_ba_rccl_one_process_per_gpu(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs: RCCL refuses two ranks on one device (the hosted-transport test above covers the multi-rank logic here)")
    _check(_run(tmp_path, 2, "rccl"))
