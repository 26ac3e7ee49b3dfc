"""Host-side mirror of Optimizer::LocalBundleAdjustment's numerical core (reference src/Optimizer.cc:1853-2680) over the
C ABI, plus the landmark sharding used when the reduced camera system is all-reduced over GPUs."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import BAProblem, BAResult, Context, check, lib
from .optimizer import make_params

_KEYS = ("kf_Tcw", "kf_fixed", "lm_type", "lm_init", "e_kf", "e_lm", "e_type", "e_meas", "e_inv_sigma2")


def shard_problem(prob: dict, rank: int, world: int) -> dict:
    """Partition LANDMARKS (and every edge of a landmark) round-robin over ranks; keyframes are replicated.  The two endpoint
    vertices of a line stay together (their edges are classified as a pair).  Returns the rank's sub-problem plus
    'lm_ids' / 'e_ids' (indices into the full problem) to scatter results back."""
    L = len(prob["lm_type"])
    owner = np.zeros(L, np.int64)
    # line endpoints: consecutive LINE edges (start, end) name the two landmarks of one line -> same group
    group = np.arange(L)
    et, el = prob["e_type"], prob["e_lm"]
    idx = np.nonzero(et == 2)[0]
    for a, b in zip(idx[0::2], idx[1::2]):
        group[el[b]] = group[el[a]]
    uniq, inv = np.unique(group, return_inverse=True)
    owner = inv % world
    lm_ids = np.nonzero(owner == rank)[0]
    remap = -np.ones(L, np.int64); remap[lm_ids] = np.arange(len(lm_ids))
    e_ids = np.nonzero(owner[el] == rank)[0]
    out = dict(kf_Tcw=prob["kf_Tcw"], kf_fixed=prob["kf_fixed"], lm_type=prob["lm_type"][lm_ids], lm_init=np.ascontiguousarray(prob["lm_init"][lm_ids]),
               e_kf=prob["e_kf"][e_ids], e_lm=remap[el[e_ids]].astype(np.int32), e_type=et[e_ids], e_meas=np.ascontiguousarray(prob["e_meas"][e_ids]),
               e_inv_sigma2=prob["e_inv_sigma2"][e_ids], lm_ids=lm_ids, e_ids=e_ids)
    return out


class Communicator:
    """planar_comm: RCCL communicator created from an id made on rank 0 (`Communicator.unique_id()`)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(lib().planar_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx: Context, uid: bytes, nranks: int, rank: int):
        self.L = lib()
        h = C.c_void_p()
        self._uid = C.create_string_buffer(uid, 128)
        check(self.L.planar_comm_create(ctx.h, self._uid, nranks, rank, C.byref(h)))
        self.h, self.nranks, self.rank = h, nranks, rank

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_comm_destroy(self.h)
            self.h = None


class HostedCommunicator:
    """planar_comm over a transport this program owns: all_reduce(buf: np.ndarray[float64], op: 'sum' | 'max') in place.
    `HostedCommunicator.torch(ctx)` uses the default torch.distributed group (gloo or nccl) - several processes may then share one GPU."""

    def __init__(self, ctx: Context, all_reduce, nranks: int, rank: int):
        self.L = lib()
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_int)

        def _cb(_user, buf, n, op):
            try:
                all_reduce(np.ctypeslib.as_array(buf, shape=(n,)), "sum" if op == 0 else "max")
                return 0
            except Exception:   # an exception must not unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        self._cb = FN(_cb)
        h = C.c_void_p()
        check(self.L.planar_comm_create_hosted(ctx.h, self._cb, None, nranks, rank, C.byref(h)))
        self.h, self.nranks, self.rank = h, nranks, rank

    @classmethod
    def torch(cls, ctx: Context):
        import torch
        import torch.distributed as dist

        def ar(a, op):
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
        return cls(ctx, ar, dist.get_world_size(), dist.get_rank())

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_comm_destroy(self.h)
            self.h = None


def local_bundle_adjustment(prob: dict, params: dict, its1: int = 5, its2: int = 10, ctx: Context | None = None, comm=None, stop_flag=None):
    """Runs planar_local_ba on a synth.ba_problem()-style dict (or a shard of it).  Returns kf_Tcw, lm, e_outlier, lm_iters.
    comm: Communicator (RCCL) or HostedCommunicator; stop_flag: a ctypes.c_ubyte the caller may set (bool* pbStopFlag of the reference)."""
    L = lib()
    ctx = ctx or Context(0)
    a = {k: np.ascontiguousarray(prob[k]) for k in _KEYS}
    K, NL, NE = len(a["kf_fixed"]), len(a["lm_type"]), len(a["e_kf"])
    out = dict(kf_Tcw=np.zeros((K, 16), np.float32), lm=np.zeros((NL, 4), np.float64), e_outlier=np.zeros(NE, np.uint8))
    P = BAProblem(K, a["kf_Tcw"].ctypes.data, a["kf_fixed"].ctypes.data, NL, a["lm_type"].ctypes.data, a["lm_init"].ctypes.data, NE,
                  a["e_kf"].ctypes.data, a["e_lm"].ctypes.data, a["e_type"].ctypes.data, a["e_meas"].ctypes.data, a["e_inv_sigma2"].ctypes.data)
    R = BAResult(out["kf_Tcw"].ctypes.data, out["lm"].ctypes.data, out["e_outlier"].ctypes.data, 0, 0)
    prm = make_params(params)
    check(L.planar_local_ba(ctx.h, C.byref(P), C.byref(prm), its1, its2, C.byref(R), C.byref(stop_flag) if stop_flag is not None else None, comm.h if comm else None))
    out["lm_iters"], out["stopped"] = R.lm_iterations, R.stopped
    return out
