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


def _run(world, args, timeout=300, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
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


def test_one_band_per_process_with_the_copy_transport():
    """The same with OCTANE_TILED_TRANSPORT=copy: what crosses ranks is pulled through the IPC mappings by runtime copies, the
    kernels read local memory only."""
    codes, outs = _run(2, (320, 288, 3, 2, 12, 1), extra_env={"OCTANE_TILED_TRANSPORT": "copy"})
    assert all(c == 0 for c in codes), "\n".join(outs)
    line = [l for l in outs[0].splitlines() if l.startswith("MP_RESULT")]
    assert line and "ok=True" in line[0], outs[0]


def test_an_abandoned_persistent_solve_fails_every_rank():
    """the process form runs its replicated levels through the persistent mid-level solve as well.  With the
    test hook on in both ranks the 320 x 250 level's solve is abandoned; both ranks' octane_vof_mp_run must return the error
    together, last_iterations() must say -2, and the following run (hook off) must give the flow of the first one bit for bit."""
    codes, outs = _run(2, (640, 500, 2, 1, 10, 200000, "fault"))
    assert all(c == 0 for c in codes), "\n".join(outs)
    for o in outs:
        line = [l for l in o.splitlines() if l.startswith("MP_FAULT_RESULT")]
        assert line and "ok=True" in line[0], o


def test_a_dead_rank_does_not_leave_the_survivor_spinning():
    """a rank that dies mid-run.  Two ranks solve in a loop; rank 1 is killed (its exact PID) while they run.
    Rank 0 must come back from octane_vof_mp_run with an error within the barrier's time-out (4 s here, 120 s by default)
    -- one time-out, not one per remaining phase boundary --, every later call must fail at once, and close must return."""
    import time
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OCTANE_MP_TIMEOUT_S="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_band_worker.py"), "640", "576", "2", "3", "30", "1", "loop"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    try:
        line = ""
        t0 = time.time()
        while "MP_LOOP_RUNNING" not in line and time.time() - t0 < 240:      # the first torch import of a fresh box is slow
            line = procs[0].stdout.readline()
            if not line and procs[0].poll() is not None:
                break
        assert "MP_LOOP_RUNNING" in line, line
        time.sleep(2.0)                      # both ranks are inside octane_vof_mp_run now
        procs[1].kill()
        t_kill = time.time()
        out0 = procs[0].communicate(timeout=60)[0]
        waited = time.time() - t_kill
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [l for l in out0.splitlines() if l.startswith("MP_DEAD_RESULT")]
    print(out0)
    assert res and procs[0].returncode == 0, out0
    assert waited < 25.0, f"the survivor needed {waited:.1f} s after the kill (time-out 4 s)"
