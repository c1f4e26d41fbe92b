"""The row-band solve with one band per PROCESS (octane_vof_mp_*, the one-process-per-GPU launch): HIP IPC mappings
of the other ranks' arenas, a phase barrier in POSIX shared memory.  On a one-GPU box the ranks share device 0 -- the
protocol, the IPC mapping and the kernels are the ones several GPUs would run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, args, timeout=300):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_band_worker.py"), *map(str, args)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:              # never leave a rank behind (exact PIDs we started)
            if p.poll() is None:
                p.kill()
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world,args", [(2, (320, 288, 3, 2, 12, 1)), (3, (300, 420, 3, 1, 10, 1)), (2, (260, 200, 2, 2, 10, 1, "hint"))])
def test_one_band_per_process_matches_plain_plan(world, args):
    codes, outs = _run(world, args)
    assert all(c == 0 for c in codes), "\n".join(outs)
    line = [l for l in outs[0].splitlines() if l.startswith("MP_RESULT")]
    assert line and "ok=True" in line[0], outs[0]
    assert "banded=0" not in line[0]
