"""Child process of tests/test_gpu_tiled_mp.py::test_torch_exchange_on_the_nccl_backend_single_rank: torch.distributed on backend nccl
(= RCCL) with the one rank a one-GPU box allows, octane_amd.exchange.TorchExchange in its DEVICE-BUFFER mode (no host staging) on memory that
torch did not allocate (a plan's arena), through the C callback types."""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octane_amd import capi, exchange  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    ex = exchange.TorchExchange(torch.device("cuda", 0))
    assert not ex.staged and "nccl" in ex.name, ex.name
    st = ex.c_struct()
    # foreign device memory: the library's own allocation (a plan's arena), not torch's caching allocator
    pl = capi.Plan(256, 128, 1, capi.FlowParams(kiters=1))
    hip = C.CDLL(None)
    ptr = C.c_void_p()
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    assert hip.hipMalloc(C.byref(ptr), 1 << 16) == 0
    t = torch.as_tensor(exchange._DevMem(ptr.value, 1 << 16), device="cuda:0")
    t.copy_(torch.arange(1 << 16, dtype=torch.int64, device="cuda:0").to(torch.uint8))
    want = int(t.to(torch.int64).sum())
    recv = (C.c_void_p * 1)(None)
    rc1 = st.all_gather(None, ptr.value, recv, 1 << 16)            # RCCL all-gather of one rank, on the aliased foreign buffer
    rc2 = st.sendrecv(None, 0, None)                               # an empty batch
    ok = rc1 == 0 and rc2 == 0 and int(t.to(torch.int64).sum()) == want and ex.calls["all_gather"] == 1
    print(f"NCCL_EXCHANGE_RESULT name={ex.name!r} rc={rc1},{rc2} ok={ok}", flush=True)
    pl.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
