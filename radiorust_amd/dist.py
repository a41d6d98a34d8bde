"""Control-plane helpers for N independent channels on N GPUs.

The hot path shards by IQ channel (one chain per GPU): there is no
data-path collective.  torch.distributed is used only to line the ranks up
(barrier) and to take the maximum of the per-rank elapsed time — `nccl` (= RCCL)
on the GPU box, `gloo` in the CPU tests.
"""
from __future__ import annotations

import os


class Ranks:
    def __init__(self, backend: str | None = None, device_index: int | None = None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = backend
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            backend = backend or "nccl"
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank if device_index is None else device_index)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend

    # one independent IQ channel per rank: seeds 1..N
    def channel_seed(self) -> int:
        return self.rank + 1

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def whole_job_rate(samples_per_rank_per_step: int, steps: int, world: int, elapsed_max_s: float) -> float:
    """MSamples/s of the whole job: every rank processed its own channel."""
    return float(world) * samples_per_rank_per_step * steps / elapsed_max_s / 1e6
