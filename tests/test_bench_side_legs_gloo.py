"""bench.py's multi-GPU side legs (round 5: after the timed pair headline of a default `--gpus N` run the same ranks measure BASELINE
configs[3] and configs[4]) must never cost the headline: a leg that RAISES on one rank is reported and ends the sequence on every rank,
a leg that HANGS is ended by the watchdog, rank 0 prints the headline with whatever finished and every rank leaves with exit code 0.
The mechanism has no GPU dependence: here two gloo ranks on the CPU run stand-in legs through bench.multi_gpu_legs."""
import json
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scenario, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OCTANE_BENCH_SECONDARY_BUDGET_S="3")
    import time
    import torch
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def ok_leg():
        dist.barrier()
        return ({"value": 1.0} if rank == 0 else None), 0

    def raising_leg():
        if rank == 1:
            raise RuntimeError("boom on rank 1")
        return ({"value": 2.0} if rank == 0 else None), 0

    def hanging_leg():
        if rank == 1:
            time.sleep(3600)          # a rank that never comes back from a collective
        dist.barrier()
        return ({"value": 3.0} if rank == 0 else None), 0

    legs = {"ok": (("a", ok_leg), ("b", ok_leg)),
            "raise": (("a", ok_leg), ("b", raising_leg), ("c", ok_leg)),
            "hang": (("a", ok_leg), ("b", hanging_leg), ("c", ok_leg))}[scenario]

    def emit(side):
        if rank == 0:
            with open(out_path, "w") as f:
                json.dump({"metric": "headline", "value": 42.0, "secondary_multi_gpu": side}, f)
    side = bench.multi_gpu_legs(None, None, None, None, torch, dist, world, rank, 0, None, emit, legs=legs)
    emit(side)
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["ok", "raise", "hang"])
def test_side_legs_never_cost_the_headline(tmp_path, scenario):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    out = tmp_path / "line.json"
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scenario, str(out))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, (scenario, p.exitcode)          # every rank leaves with exit code 0 whatever the legs did
    d = json.load(open(out))
    assert d["value"] == 42.0                                   # the headline is printed in every scenario
    side = d["secondary_multi_gpu"]
    if scenario == "ok":
        assert side["a"]["value"] == 1.0 and side["b"]["value"] == 1.0 and "error" not in side
    elif scenario == "raise":
        assert side["a"]["value"] == 1.0 and "boom on rank 1" in side["b"]["error"] and "c" not in side      # the sequence ends there
    else:
        assert side["a"]["value"] == 1.0 and "did not finish" in side["error"] and "b" not in side and "c" not in side
