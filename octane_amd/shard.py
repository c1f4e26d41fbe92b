"""Multi-GPU sharding of the flow path: independent image pairs, one process per GPU.

The path partitions over pairs: pair b belongs to rank b % world -- the same rule
octane_vof_batch_run applies to its device list -- and there is no collective in the data path.  The only
communication is the bench's barrier / max-over-ranks timing and an optional gather of per-pair checksums,
both through torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""
from __future__ import annotations

import os


def pairs_for_rank(npairs: int, rank: int, world: int) -> list[int]:
    return [b for b in range(npairs) if b % world == rank]


def init_from_env(backend: str, device_id=None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    kw = {}
    if device_id is not None:
        kw["device_id"] = device_id
    dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def max_over_ranks(seconds: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj, dst: int = 0):
    """Rank `dst` gets the list of every rank's object (result gather: small, host-side)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def whole_job_mpix(world_pixels_per_step: int, steps: int, seconds_max: float) -> float:
    return world_pixels_per_step * steps / seconds_max / 1e6
