"""bench.py's multi-GPU side legs (round 5: after the timed pair headline of a default `--gpus N` run, rank 0 measures BASELINE configs[3]
and configs[4] in CHILD jobs of N fresh ranks each) must never cost the headline: a leg that fails, CRASHES (a GPU fault kills a rank) or
HANGS is reported in `secondary_multi_gpu`, the other parent ranks wait on the rendezvous store, every parent rank goes on to print /
leave with exit code 0.  The mechanism has no GPU dependence: here two gloo ranks on the CPU run stand-in child commands through
bench.multi_gpu_legs."""
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


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    py = sys.executable
    legs = (
        ("fine", [py, "-c", "import json, os; print('noise'); print(json.dumps({'value': 12.5, 'saw_rank_env': 'RANK' in os.environ}))"], 30.0),
        ("fails", [py, "-c", "import json, sys; print(json.dumps({'value': None, 'error': 'no transport'})); sys.exit(3)"], 30.0),
        ("crashes", [py, "-c", "import os; os.abort()"], 30.0),
        ("hangs", [py, "-c", "import time; time.sleep(3600)"], 2.0),
        ("fine_again", [py, "-c", "import json; print(json.dumps({'value': 7}))"], 30.0),
    )
    side = bench.multi_gpu_legs(dist, world, rank, legs=legs)
    assert (side is None) == (rank != 0)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"metric": "headline", "value": 42.0, "secondary_multi_gpu": side}, f)
    dist.barrier()                      # the parents' process group is intact after the legs
    dist.destroy_process_group()


def test_side_legs_never_cost_the_headline(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    out = tmp_path / "line.json"
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(out))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, p.exitcode                      # every parent rank leaves with exit code 0 whatever the legs did
    d = json.load(open(out))
    assert d["value"] == 42.0
    side = d["secondary_multi_gpu"]
    assert side["fine"]["value"] == 12.5 and side["fine"]["exit_code"] == 0 and side["fine"]["saw_rank_env"] is False   # the child gets a clean launcher environment
    assert side["fails"]["exit_code"] == 3 and side["fails"]["error"] == "no transport"
    assert side["crashes"]["exit_code"] != 0 and "no JSON line" in side["crashes"]["error"]
    assert "did not finish within 2 s" in side["hangs"]["error"] and side["hangs"]["leg_seconds"] < 20
    assert side["fine_again"]["value"] == 7                     # the sequence goes on after a crash and a hang
