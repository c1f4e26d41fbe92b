import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch, torch.distributed as dist
    from octane_amd import capi, exchange
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ex = exchange.TorchExchange(torch.device("cpu")); st = ex.c_struct()
    n = 7*2048
    mine = np.arange(n, dtype=np.float64); mirror = np.zeros((world, n))
    recv = (C.c_void_p * world)(*[None if c == rank else mirror[c].ctypes.data for c in range(world)])
    rows = np.zeros((64, 320), np.float32)
    peer = 1 - rank
    ops = (capi.Xfer * 4)(capi.Xfer(peer, 1, rows[0:2].ctypes.data, 2*320*4), capi.Xfer(peer, 0, rows[2:4].ctypes.data, 2*320*4),
                          capi.Xfer(peer, 1, rows[4:5].ctypes.data, 320*4), capi.Xfer(peer, 0, rows[5:6].ctypes.data, 320*4))
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100): st.all_gather(None, mine.ctypes.data, recv, mine.nbytes)
        t1 = time.perf_counter()
        for _ in range(100): st.sendrecv(None, 4, ops)
        t2 = time.perf_counter()
        if rank == 0: print(f"all_gather {1e3*(t1-t0)/100:.2f} ms/call, sendrecv {1e3*(t2-t1)/100:.2f} ms/call", flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    import torch.multiprocessing as mp, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
