"""The collective transport's callbacks on CPU: three gloo ranks drive octane_amd.exchange.TorchExchange exactly as the library's
row-band loop does (octane_amd/csrc/vof_tiled.hip: gather_parts, exchange_rows, the gather of the flow bands) -- through the C
function-pointer types of include/octane_vof.h, on host memory standing in for the bands' planes.  What is checked is the contract the
C side relies on: rank c's all-gather contribution arrives at recv[c]; between a pair of ranks the k-th send matches the k-th receive;
a middle band exchanges with both neighbours in one batch; nothing outside the named rows is touched.  (The GPU tests run the same
callbacks on device memory, staged through the host: tests/test_gpu_tiled_mp.py.)"""
import ctypes as C
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


PITCH, ROWS_PER_BAND, NPLANES = 64, 32, 3


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from octane_amd import capi, exchange
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ex = exchange.TorchExchange(torch.device("cpu"))
    st = ex.c_struct()
    h = world * ROWS_PER_BAND
    y0, y1 = rank * ROWS_PER_BAND, (rank + 1) * ROWS_PER_BAND
    # every rank's planes hold ITS OWN value pattern on its band's rows and a poison (-1) elsewhere
    planes = [np.full((h, PITCH), -1.0, np.float32) for _ in range(NPLANES)]
    for p, pl in enumerate(planes):
        pl[y0:y1] = 1000 * rank + 10 * p + np.arange(y0, y1, dtype=np.float32)[:, None] / 1000.0
    # --- the partial sums: all_gather into a mirror laid out [rank][block], own block untouched ---
    nparts = 7 * 16
    mine = (np.arange(nparts, dtype=np.float64) + 100.0 * rank)
    mirror = np.full((world, 2, nparts), -5.0)
    recv = (C.c_void_p * world)(*[None if c == rank else mirror[c, 1].ctypes.data for c in range(world)])
    rc = st.all_gather(None, mine.ctypes.data, recv, mine.nbytes)
    ok = rc == 0
    for c in range(world):
        want = -5.0 if c == rank else np.arange(nparts) + 100.0 * c
        ok = ok and np.array_equal(mirror[c, 1], np.broadcast_to(want, (nparts,))) and np.all(mirror[c, 0] == -5.0)
    # --- edge rows, as exchange_rows builds them: spec (plane, a_lo, a_hi, b_lo, b_hi) ---
    specs = [(0, 0, 2, 0, 2), (1, 0, 1, 0, 1), (2, 1, 2, 0, 0)]          # p-like (two rows each way), r-like (one), wy-like (row y0 - 2 only)
    ops = []

    def add(peer, send, p, ya, yb):
        if yb > ya:
            ops.append(capi.Xfer(peer, send, planes[p][ya:yb].ctypes.data, (yb - ya) * PITCH * 4))
    for p, a_lo, a_hi, b_lo, b_hi in specs:
        if rank > 0:
            add(rank - 1, 0, p, y0 - a_hi, y0 - a_lo); add(rank - 1, 1, p, y0 + b_lo, y0 + b_hi)
        if rank < world - 1:
            add(rank + 1, 0, p, y1 + b_lo, y1 + b_hi); add(rank + 1, 1, p, y1 - a_hi, y1 - a_lo)
    arr = (capi.Xfer * len(ops))(*ops)
    ok = ok and st.sendrecv(None, len(ops), arr) == 0

    def owner_value(p, y):
        return 1000 * (y // ROWS_PER_BAND) + 10 * p + y / 1000.0
    for p, a_lo, a_hi, b_lo, b_hi in specs:
        got = set()
        if rank > 0:
            got |= set(range(y0 - a_hi, y0 - a_lo))
        if rank < world - 1:
            got |= set(range(y1 + b_lo, y1 + b_hi))
        for y in range(h):
            if y0 <= y < y1 or y in got:
                ok = ok and np.all(planes[p][y] == np.float32(owner_value(p, y)))
            else:
                ok = ok and np.all(planes[p][y] == -1.0)          # untouched
    # --- the flow bands at the end of a level: everybody gets everybody's rows ---
    ops = []
    for c in range(world):
        if c == rank:
            continue
        add(c, 1, 0, y0, y1)
        add(c, 0, 0, c * ROWS_PER_BAND, (c + 1) * ROWS_PER_BAND)
    arr = (capi.Xfer * len(ops))(*ops)
    ok = ok and st.sendrecv(None, len(ops), arr) == 0
    ok = ok and all(np.all(planes[0][y] == np.float32(owner_value(0, y))) for y in range(h))
    q.put((rank, bool(ok), ex.name, dict(ex.calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_three_gloo_ranks_move_partial_sums_edge_rows_and_flow_bands():
    import torch.multiprocessing as mp
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1, 2] and all(r[1] for r in res), res
    assert all(r[2].startswith("torch.distributed/gloo") for r in res)
    assert all(r[3]["all_gather"] == 1 and r[3]["sendrecv"] == 2 for r in res)
