"""The collective transport of the one-band-per-process row-band solve, carried by torch.distributed.

north_star names "RCCL over xGMI only for halo exchange / result gather"; SURVEY 5 / 8(e) ask for a collective-library transport next
to the peer-mapped one.  The library (octane_amd/csrc/vof_tiled.hip, OCTANE_TRANSPORT_COLLECTIVE) hands everything that crosses ranks
-- per PCG iteration the seven partial sums of every workgroup and a few rows per inner edge, per linearisation two rows of the flow,
per level the bands of the flow -- to two callbacks (include/octane_vof.h, octane_vof_exchange): an all-gather and a batch of
point-to-point transfers, on DEVICE memory of the calling rank.  This module implements them:

* backend ``nccl`` (= RCCL on ROCm; ranks on distinct GPUs of one node, the bytes travel over xGMI): the device pointers are
  aliased as torch tensors (``__cuda_array_interface__``, no copy) and go to ``dist.all_gather`` / ``dist.batch_isend_irecv``;
* backend ``gloo`` (or any backend without GPU point-to-point; also the only way several ranks can share ONE GPU, which RCCL
  refuses, profiles/r2_rccl_same_device.txt): the same transfers on host staging copies (the all-gather as point-to-point
  transfers: gloo's own all_gather needs 226 ms per call on the GPU boxes of this pool, profiles/r4_exchange_costs.txt).

Either way the bytes that arrive are the bytes that were sent, the kernels and their order are the in-place transport's, every rank
folds the partial sums of all ranks in the same order: the flow is the copy transport's bit for bit
(tests/test_gpu_tiled_mp.py).  The reference has no counterpart (it picks one device, ref src/oct_variational_optical_flow.cu:1251-1265);
what has to be preserved is its one global PCG per linearisation (ref .cu:1105-1195).
"""
from __future__ import annotations

import ctypes as C

from . import capi


class _DevMem:
    """A span of device memory as the CUDA array interface describes it (torch.as_tensor aliases it, no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}


class TorchExchange:
    """octane_vof_exchange on torch.distributed.  `device`: this rank's torch device; `group`: the process group (default: world).
    Keep the object alive as long as the plan that uses it."""

    def __init__(self, device, group=None, staged: bool | None = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device, self.group = torch, dist, torch.device(device), group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        backend = str(dist.get_backend(group))
        # RCCL moves device tensors itself; everything else is staged through the host
        self.staged = (backend != "nccl" and self.device.type != "cpu") if staged is None else bool(staged)
        self.name = f"torch.distributed/{backend}" + (" (host-staged)" if self.staged else " (device buffers)")
        self.calls = {"all_gather": 0, "sendrecv": 0, "bytes": 0}
        self._ag = capi.XCHG_ALL_GATHER_FN(self._all_gather)
        self._sr = capi.XCHG_SENDRECV_FN(self._sendrecv)
        self._struct = capi.Exchange(None, self._ag, self._sr, self.name.encode()[:47])

    def c_struct(self) -> "capi.Exchange":
        return self._struct

    def _tensor(self, ptr: int, nbytes: int):
        if self.device.type == "cpu":      # host memory (the CPU rehearsal of the protocol, tests/test_shard_gloo.py)
            return self.torch.frombuffer((C.c_ubyte * int(nbytes)).from_address(int(ptr)), dtype=self.torch.uint8)
        return self.torch.as_tensor(_DevMem(int(ptr), int(nbytes)), device=self.device)

    def _sync(self):
        if self.device.type != "cpu":
            self.torch.cuda.synchronize(self.device)

    def _global(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    # ---- the callbacks (host-synchronous: the data is in place when they return) ----
    def _all_gather(self, user, send, recv, nbytes):
        try:
            torch, dist = self.torch, self.dist
            mine = self._tensor(send, nbytes)
            outs = [self._tensor(recv[c], nbytes) if c != self.rank else None for c in range(self.world)]
            if self.staged or self.device.type == "cpu":
                # point-to-point, not dist.all_gather: gloo's all_gather needs 226 ms per call for 115 KB on the GPU boxes of this pool
                # (its chunked ring over loopback TCP; 1.4 ms in the build container), a batch of sends and receives 1 ms
                h = mine.cpu() if self.staged else mine
                got = {c: (torch.empty_like(h) if self.staged else outs[c]) for c in range(self.world) if c != self.rank}
                p2p = []
                for c in range(self.world):
                    if c != self.rank:
                        p2p.append(dist.P2POp(dist.isend, h, self._global(c), group=self.group))
                        p2p.append(dist.P2POp(dist.irecv, got[c], self._global(c), group=self.group))
                for req in dist.batch_isend_irecv(p2p):
                    req.wait()
                if self.staged:
                    for c, g in got.items():
                        outs[c].copy_(g)
            else:
                own = torch.empty_like(mine)
                dist.all_gather([o if o is not None else own for o in outs], mine, group=self.group)
            self._sync()
            self.calls["all_gather"] += 1
            self.calls["bytes"] += int(nbytes) * (self.world - 1)
            return 0
        except Exception as e:  # pragma: no cover
            print(f"octane TorchExchange.all_gather failed on rank {self.rank}: {e!r}", flush=True)
            return 1

    def _sendrecv(self, user, n, ops):
        try:
            torch, dist = self.torch, self.dist
            if n <= 0:                       # (torch's batch_isend_irecv refuses an empty batch)
                self.calls["sendrecv"] += 1
                return 0
            p2p, back = [], []
            for i in range(n):
                op = ops[i]
                t = self._tensor(op.buf, op.bytes)
                if self.staged:
                    if op.send:
                        h = t.cpu()
                    else:
                        h = torch.empty(op.bytes, dtype=torch.uint8)
                        back.append((t, h))
                    t = h
                p2p.append(dist.P2POp(dist.isend if op.send else dist.irecv, t, self._global(op.peer), group=self.group))
                self.calls["bytes"] += int(op.bytes) if op.send else 0
            for req in dist.batch_isend_irecv(p2p):
                req.wait()
            for t, h in back:
                t.copy_(h)
            self._sync()
            self.calls["sendrecv"] += 1
            return 0
        except Exception as e:  # pragma: no cover
            print(f"octane TorchExchange.sendrecv failed on rank {self.rank}: {e!r}", flush=True)
            return 1
