"""Multi-GPU plumbing of the batched path: frames are independent, so N ranks shard them with NO data-path collective.
`torch.distributed` (RCCL when the backend is "nccl") is used only for the start/stop barrier and the max-over-ranks
time (bench.py contract).  The same code runs with the gloo backend on CPU for the world_size-2 tests."""
from __future__ import annotations

import os


class Ranks:
    def __init__(self, backend: str | None = None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
            dist.init_process_group(backend or "nccl", **kw)
            self.dist = dist

    def frame_ids(self, frames_per_rank: int):
        """Global ids of the frames this rank owns (weak scaling: every rank owns `frames_per_rank` frames)."""
        start = self.rank * frames_per_rank
        return range(start, start + frames_per_rank)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device if self.dist.get_backend() == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device if self.dist.get_backend() == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def whole_job_fps(world: int, frames_per_rank: int, steps: int, max_elapsed_s: float) -> float:
    """`value` of the bench line: frames processed by ALL ranks divided by the slowest rank's time."""
    return world * frames_per_rank * steps / max_elapsed_s
