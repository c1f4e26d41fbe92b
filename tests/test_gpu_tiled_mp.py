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


def _result(out):
    import json
    line = [l for l in out.splitlines() if l.startswith("MP_RESULT")]
    assert line and "ok=True" in line[0], out
    crc = line[0].split("crc=")[1].split()[0]
    return crc, json.loads(line[0].split("info=")[1])


def test_collective_transport_over_torch_distributed_gloo_equals_the_copy_transport():
    """the third transport.  Two ranks share GPU 0; everything that crosses ranks -- the partial sums, the rows
    beyond the band edges, the flow rows and bands -- travels through torch.distributed (gloo: staged through the host; on a node the
    same code runs on backend nccl = RCCL over xGMI with the device buffers themselves, octane_amd/exchange.py).  The kernels read
    local memory only, as with the copy transport, and every rank folds all ranks' partial sums in the same order: the flow must be
    the copy transport's BIT FOR BIT, within 2e-5 of the plain plan and of the oracle, with equal iteration counts."""
    args = (320, 288, 3, 2, 12, 1)
    codes, outs = _run(2, args, extra_env={"OCTANE_TILED_TRANSPORT": "copy"})
    assert all(c == 0 for c in codes), "\n".join(outs)
    crc_copy, info_copy = _result(outs[0])
    codes, outs = _run(2, args, extra_env={"OCTANE_TILED_TRANSPORT": "collective", "OCTANE_TEST_EXCHANGE": "1"})
    assert all(c == 0 for c in codes), "\n".join(outs)
    crc_coll, info_coll = _result(outs[0])
    print("COLLECTIVE", crc_copy, crc_coll, info_coll)
    assert info_copy["transport_used"] == "copy" and info_coll["transport_used"] == "collective" and info_coll["forced"]
    assert info_coll["exchange"].startswith("torch.distributed/gloo")
    assert info_coll["exchange_calls"]["all_gather"] > 0 and info_coll["exchange_calls"]["sendrecv"] > 0
    assert crc_coll == crc_copy


def test_collective_transport_with_bands_on_the_q_recomputing_kernel():
    """Bands of 2 Mpixel run the q-recomputing kernel, which wants other rows of its neighbours than the stored-q kernel: p on TWO rows
    beyond each edge, r on one, and wy of the row above the upper ring row after every assembly (exchange_rows' four-plane batch and the
    one-sided wy transfer).  Collective against copy, bit for bit; both within 2e-5 of the plain plan with equal iteration counts."""
    args = (2048, 2112, 1, 1, 6, 1)
    codes, outs = _run(2, args, extra_env={"OCTANE_TILED_TRANSPORT": "copy"})
    assert all(c == 0 for c in codes), "\n".join(outs)
    crc_copy, _ = _result(outs[0])
    codes, outs = _run(2, args, extra_env={"OCTANE_TILED_TRANSPORT": "collective", "OCTANE_TEST_EXCHANGE": "1"})
    assert all(c == 0 for c in codes), "\n".join(outs)
    crc_coll, info = _result(outs[0])
    assert info["transport_used"] == "collective" and info["exchange_calls"]["sendrecv"] > 0
    assert crc_coll == crc_copy


def test_collective_transport_three_ranks_with_hint():
    """Three ranks (a middle band with two neighbours), first-guess hint term, the collective transport chosen by force."""
    codes, outs = _run(3, (300, 420, 3, 1, 10, 1), extra_env={"OCTANE_TILED_TRANSPORT": "collective", "OCTANE_TEST_EXCHANGE": "1"})
    assert all(c == 0 for c in codes), "\n".join(outs)
    crc, info = _result(outs[0])                 # ok=True: within 2e-5 of the plain plan and of the oracle, equal iteration counts
    assert info["transport_used"] == "collective" and info["exchange_calls"]["sendrecv"] > 0


def test_first_contact_selfcheck_keeps_the_in_place_transport_when_it_works():
    """Default environment: octane_vof_mp_selfcheck solves a small frame under inplace first; on this box (IPC mappings of one GPU)
    that passes, so the plan reads in place and says so."""
    codes, outs = _run(2, (320, 288, 3, 2, 12, 1), extra_env={"OCTANE_TEST_EXCHANGE": "1"})
    assert all(c == 0 for c in codes), "\n".join(outs)
    _, info = _result(outs[0])
    print("SELFCHECK", info)
    assert info["transport_used"] == "inplace" and info["selfcheck"] == "first candidate passed" and not info["forced"]
    assert info["peer_ok"] and info["bands"] == 2 and info["check_rel_l2"] and info["check_rel_l2"][0] <= 2e-5


@pytest.mark.parametrize("bits,want,exchange", [
    # (bit 0 alone -- in-place reads return the wrong block -> the copy transport -- is drilled in the thread form, tests/test_gpu_tiled.py)
    (5, "collective", "1"),        # in-place reads give the wrong sums AND the copy transport leaves its mirror unfilled -> the collective library
    (8, "collective", "1"),        # the IPC mappings cannot be opened at all -> the collective library, no check candidates before it
])
def test_first_contact_selfcheck_downgrades_automatically(bits, want, exchange):
    """VERDICT r3 item 2(b), the drill: with the DIAGNOSTIC library's fault hook (OCTANE_TEST_BREAK_TRANSPORT; the product library has
    none) a transport gives wrong sums / cannot be set up.  The self-check must notice, say so once on stderr, fall back to the next
    transport on every rank alike -- and the solve that follows must still be right (ok=True: plain plan, oracle, iteration counts)."""
    from octane_amd import capi
    if not os.path.exists(capi.DIAG_LIB_PATH):
        pytest.skip("diagnostic library not built")
    codes, outs = _run(2, (320, 288, 3, 2, 12, 1), extra_env={"OCTANE_TEST_EXCHANGE": exchange, "OCTANE_LIB": capi.DIAG_LIB_PATH,
                                                              "OCTANE_TEST_BREAK_TRANSPORT": str(bits)})
    assert all(c == 0 for c in codes), "\n".join(outs)
    _, info = _result(outs[0])
    print("DOWNGRADE", bits, info)
    assert info["transport_used"] == want
    if bits == 8:
        assert not info["peer_ok"]
    else:
        assert info["selfcheck"] == "downgraded" and info["candidates_tried"] >= 2
        assert "self-check" in outs[0]


def test_no_usable_transport_is_an_error_on_every_rank_not_a_hang():
    """IPC unavailable (drill) and no collective library registered: octane_vof_mp_connect has to fail on both ranks, with a message
    that names the way out."""
    from octane_amd import capi
    if not os.path.exists(capi.DIAG_LIB_PATH):
        pytest.skip("diagnostic library not built")
    codes, outs = _run(2, (320, 288, 3, 2, 12, 1), extra_env={"OCTANE_LIB": capi.DIAG_LIB_PATH, "OCTANE_TEST_BREAK_TRANSPORT": "8"})
    assert all(c == 3 for c in codes), "\n".join(outs)
    assert all("octane_vof_mp_set_exchange" in o for o in outs)


def test_an_abandoned_persistent_solve_fails_every_rank():
    """the process form runs its replicated levels through the persistent mid-level solve as well.  With the
    test hook on in both ranks the 320 x 250 level's solve is abandoned; both ranks' octane_vof_mp_run must return the error
    together, last_iterations() must say -2, and the following run (hook off) must give the flow of the first one bit for bit."""
    from octane_amd import capi
    if not os.path.exists(capi.DIAG_LIB_PATH):
        pytest.skip("diagnostic library not built")
    codes, outs = _run(2, (640, 500, 2, 1, 10, 200000, "fault"), extra_env={"OCTANE_LIB": capi.DIAG_LIB_PATH})   # the hook lives in the diagnostic library only
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


def test_rccl_exchange_library_single_rank():
    """liboctane_xchg_rccl.so (octane_amd/csrc/exchange_rccl.cpp): the collective transport's callbacks on RCCL itself, for C++ host
    programs.  Two ranks cannot share this box's one GPU (RCCL refuses: profiles/r2_rccl_same_device.txt), so what can be checked here is
    what one rank can do: the library loads, RCCL initialises a communicator on the device, the callbacks run (an all-gather that has
    nobody else to hear from, an empty batch of transfers) and everything is torn down.  The protocol the callbacks carry is tested
    through torch.distributed above and on CPU (tests/test_exchange_gloo.py)."""
    import ctypes as C
    from octane_amd import capi
    path = os.path.join(ROOT, "octane_amd", "liboctane_xchg_rccl.so")
    if not os.path.exists(path):
        pytest.skip("liboctane_xchg_rccl.so has not been built (no RCCL?)")
    capi.lib()
    X = C.CDLL(path)
    ident = C.create_string_buffer(128)
    assert X.octane_rccl_unique_id(ident) == 0
    ex = capi.Exchange()
    X.octane_rccl_exchange_create.argtypes = [C.POINTER(capi.Exchange), C.c_void_p, C.c_int, C.c_int, C.c_int]
    assert X.octane_rccl_exchange_create(C.byref(ex), ident, 0, 1, 0) == 0
    try:
        assert ex.name.decode().startswith("RCCL ")
        import torch
        buf = torch.arange(64, dtype=torch.float32, device="cuda")
        recv = (C.c_void_p * 1)(None)
        assert ex.all_gather(ex.user, buf.data_ptr(), recv, buf.numel() * 4) == 0
        assert ex.sendrecv(ex.user, 0, None) == 0
        assert float(buf.sum()) == 2016.0
    finally:
        X.octane_rccl_exchange_destroy.argtypes = [C.POINTER(capi.Exchange)]
        X.octane_rccl_exchange_destroy(C.byref(ex))


def test_torch_exchange_on_the_nccl_backend_single_rank():
    """octane_amd/exchange.py in the mode a multi-GPU node would run it in: backend nccl (= RCCL), device buffers aliased as tensors, no
    host staging.  RCCL wants one rank per GPU, so this box can give it one rank: process-group set-up on RCCL, the aliasing of memory
    torch did not allocate, an all-gather through RCCL on it and an empty batch of transfers -- everything but a second device."""
    port = _free_port()
    env = dict(os.environ, MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_exchange_worker.py")], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("NCCL_EXCHANGE_RESULT")]
    assert r.returncode == 0 and line and "ok=True" in line[0], r.stdout[-2000:] + r.stderr[-2000:]
    assert "torch.distributed/nccl (device buffers)" in line[0]
