"""The N>1 path on CPU: two gloo ranks shard a batch of pairs exactly as bench.py / octane_vof_batch_run do
(pair b -> rank b % world), solve their share, and rank 0 gathers per-pair checksums; the union must be every
pair exactly once and equal to a single-process run.  The per-pair work here is the CPU oracle (this is a test
of the sharding and timing logic, which has no GPU dependence)."""
import os
import socket
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _solve_pair(b):
    from octane_amd import synth
    from oracle import oct_oracle as oo
    a, c = synth.lattice_scene(40, 32, seed=100 + b)
    u, v, _ = oo.flow(a, c, oo.FlowParams(kiters=2, liters=1, cgiters=5))
    return zlib.crc32(u.tobytes() + v.tobytes())


def _worker(rank, world, port, npairs, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    import torch.distributed as dist
    from octane_amd import shard
    r, w = shard.init_from_env("gloo")
    assert (r, w) == (rank, world)
    mine = shard.pairs_for_rank(npairs, r, w)
    dist.barrier()
    t0 = time.perf_counter()
    res = {b: _solve_pair(b) for b in mine}
    if r == 1:
        time.sleep(0.3)                     # make the ranks' times differ
    dist.barrier()
    local = time.perf_counter() - t0 if r == 1 else 0.01
    tmax = shard.max_over_ranks(local)
    gathered = shard.gather_objects(res, dst=0)
    census = shard.rank_census("gloo")                  # what bench.py --gpus N adds to its line as "ranks"
    if r == 0:
        q.put((gathered, tmax, census))
    dist.destroy_process_group()


def test_two_ranks_shard_a_batch_and_rank0_gathers():
    import torch.multiprocessing as mp
    npairs, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, npairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, tmax, census = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert census["backend"] == "gloo" and census["world_size"] == world and census["allreduce_of_ones"] == world
    assert [r["rank"] for r in census["ranks"]] == [0, 1] and len({r["pid"] for r in census["ranks"]}) == world
    assert sorted(gathered[0]) == [0, 2, 4] and sorted(gathered[1]) == [1, 3]
    merged = {**gathered[0], **gathered[1]}
    assert merged == {b: _solve_pair(b) for b in range(npairs)}
    assert tmax >= 0.3                       # MAX over ranks, not rank 0's own time


def test_pair_assignment_matches_the_batch_entry_rule():
    from octane_amd import shard
    for world in (1, 2, 3, 8):
        seen = sorted(b for r in range(world) for b in shard.pairs_for_rank(64, r, world))
        assert seen == list(range(64))
        assert all(b % world == r for r in range(world) for b in shard.pairs_for_rank(64, r, world))
    assert shard.whole_job_mpix(8 * 25_000_000, 2, 0.5) == pytest.approx(800.0)
