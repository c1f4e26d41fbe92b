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


def rank_census(backend: str, device=None, local: int | None = None):
    """What the collective library saw, for the record of an N-rank bench line: every rank reports who it is and which device it
    drives, and a sum all-reduce of ones over the backend (RCCL on device tensors under "nccl") has to come back as the world size.
    Collective: every rank calls it.  Rank 0 gets {"backend", "world_size", "allreduce_of_ones", "ranks": [{rank, local_rank, host,
    pid, device, name, uuid, pci_bus_id}, ...]} -- distinct uuids = distinct GPUs --, the others None."""
    import socket
    import torch
    import torch.distributed as dist
    me = {"rank": int(os.environ.get("RANK", "0")), "local_rank": local, "host": socket.gethostname(), "pid": os.getpid()}
    if device is not None and getattr(device, "type", "cpu") == "cuda":
        me["device"] = str(device)
        try:                                            # a property this torch does not have must not cost an N-rank run its line
            pr = torch.cuda.get_device_properties(device)
            me.update(name=pr.name, uuid=str(getattr(pr, "uuid", "")) or None,
                      pci_bus_id=(f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}" if hasattr(pr, "pci_bus_id") else None))
        except Exception as e:                          # noqa: BLE001
            me["device_info_error"] = repr(e)
    if not dist.is_available() or not dist.is_initialized():
        return {"backend": None, "world_size": 1, "allreduce_of_ones": 1, "ranks": [me]}
    me["rank"] = dist.get_rank()
    # plain tensor collectives only (the same kind as the timing's all-reduce): a sum of ones, and an all-gather of every rank's report as
    # 1 KiB of bytes -- no pickled-object collective in an N-rank run whose one purpose is its headline line
    import json
    dev = device if backend == "nccl" else None
    try:
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        blob = json.dumps(me).encode()[:1024]
        mine = torch.zeros(1024, dtype=torch.uint8, device=dev)
        mine[:len(blob)] = torch.tensor(list(blob), dtype=torch.uint8, device=dev)
        allb = [torch.zeros(1024, dtype=torch.uint8, device=dev) for _ in range(dist.get_world_size())]
        dist.all_gather(allb, mine)
        if dist.get_rank() != 0:
            return None
        ranks = []
        for t in allb:
            raw = bytes(t.cpu().tolist()).rstrip(b"\0")
            try:
                ranks.append(json.loads(raw.decode()))
            except ValueError:
                ranks.append({"unparsed": raw[:80].decode(errors="replace")})
        return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "allreduce_of_ones": int(one.item()),
                "ranks": sorted(ranks, key=lambda r: r.get("rank", -1)),
                "distinct_devices": len({r.get("uuid") or r.get("pci_bus_id") or (r.get("host"), r.get("device")) for r in ranks})}
    except Exception as e:                              # noqa: BLE001 -- the census must never cost the line it decorates
        return {"error": repr(e), "world_size": dist.get_world_size()} if dist.get_rank() == 0 else None


def whole_job_mpix(world_pixels_per_step: int, steps: int, seconds_max: float) -> float:
    return world_pixels_per_step * steps / seconds_max / 1e6
