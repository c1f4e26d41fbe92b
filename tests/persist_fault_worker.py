"""Child process of the abandoned-solve drills in tests/test_gpu_persist.py.  The hook that makes a workgroup of the persistent mid-level
solve stay away (octane_vof_tune "persist_fault") exists in the DIAGNOSTIC library only since round 4 (no test hook in
the product kernel), so the drills run in a process that loads that library (OCTANE_LIB).  Same library source otherwise: what is drilled
-- the give-up path of the kernel's waits, the repair of the run by the host code -- is the product's text.
usage: persist_fault_worker.py host|device|bands"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from octane_amd import capi, synth  # noqa: E402


def _persist_state(capi, pl):
    import ctypes as C
    L = capi.lib()
    L.octane_vof_plan_persist_state.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    n = C.c_int()
    return L.octane_vof_plan_persist_state(pl._h, C.byref(n)), n.value


def drill_host(capi):
    """The persistent solve needs all its workgroups resident; one that never shows up (a test hook makes the last workgroup
    leave at once -- what a co-tenant process holding a CU would amount to) must not leave the others spinning: their waits give
    up after 0.25 s and every workgroup leaves.  Host-buffer call (round 3): the library notices, says so once on
    stderr, solves the pair AGAIN with one launch per iteration and returns that flow -- bit-equal to a plan with the persistent
    solve switched off.  Round 4: the plan stays in that mode for its next 16 runs, not for life, and says so through
    octane_vof_plan_persist_state; then the persistent solve is back, with the bits it had before."""
    import time
    nx, ny = 640, 500
    a, b = synth.lattice_scene(nx, ny, seed=5)
    prm = capi.FlowParams(kiters=1, liters=1, cgiters=10)
    pl = capi.Plan(nx, ny, 1, prm)
    po = capi.Plan(nx, ny, 1, prm)
    try:
        good = pl.run_host(a, b)
        its = pl.last_iterations()
        assert _persist_state(capi, pl) == (1, 0)
        po.tune("persist", 0)
        plain = po.run_host(a, b)
        pl.tune("persist_fault", 1)
        t0 = time.perf_counter()
        got = pl.run_host(a, b)                       # first attempt abandoned, second without the persistent solve
        dt = time.perf_counter() - t0
        pl.tune("persist_fault", 0)
        print(f"PERSIST fault drill (host buffers): solved again without the persistent solve after {dt:.2f} s")
        assert dt < 5.0 and pl.last_iterations() == its
        assert np.array_equal(got[0], plain[0]) and np.array_equal(got[1], plain[1])
        assert _persist_state(capi, pl) == (-16, 1)   # off for the next 16 runs, one abandoned solve so far
        for k in range(15):
            again = pl.run_host(a, b)
        assert np.array_equal(again[0], plain[0]) and _persist_state(capi, pl) == (-1, 1)
        again = pl.run_host(a, b)                     # the 16th run after the event: the persistent solve again
        assert _persist_state(capi, pl) == (1, 1)
        assert np.array_equal(good[0], again[0]) and np.array_equal(good[1], again[1])
    finally:
        pl.tune("persist_fault", 0)
        pl.close(); po.close()


def drill_device(capi):
    """a caller that drives the plan with device buffers and its own stream synchronisation (bench.py, torch
    users).  The abandoned run used to be visible only as last_iterations() == -2; now the first of octane_vof_plan_wait /
    octane_vof_plan_last_iterations called after the caller has synchronised makes the run again -- one launch per iteration, from
    the plan's own copy of the inputs, on the run's stream, into the run's output buffers -- and reports the repaired run: the
    iteration count of a good run, the flow of a plan without the persistent solve, bit for bit.  And nothing sticks: with the hook
    off and the persistent solve back on, the next run is the good one again."""
    import torch
    nx, ny = 640, 500
    a, b = synth.lattice_scene(nx, ny, seed=5)
    prm = capi.FlowParams(kiters=1, liters=1, cgiters=10)
    pl = capi.Plan(nx, ny, 1, prm)
    po = capi.Plan(nx, ny, 1, prm)
    try:
        good = pl.run_host(a, b)
        its = pl.last_iterations()
        po.tune("persist", 0)
        plain = po.run_host(a, b)
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        du, dv = torch.zeros(ny, nx, device="cuda"), torch.zeros(ny, nx, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        pl.tune("persist_fault", 1)
        pl.run_device(da.data_ptr(), db.data_ptr(), du.data_ptr(), dv.data_ptr(), s)
        torch.cuda.synchronize()
        stale = du.cpu().numpy().copy()
        assert pl.last_iterations() == its            # the asking call repaired the run ...
        torch.cuda.synchronize()
        assert not np.array_equal(stale, plain[0])    # (what the abandoned run had left was not the flow)
        assert np.array_equal(du.cpu().numpy(), plain[0]) and np.array_equal(dv.cpu().numpy(), plain[1])     # ... into the run's own buffers
        assert _persist_state(capi, pl) == (-16, 1)
        pl.tune("persist_fault", 0)
        pl.tune("persist", 1)                         # (the tune key switches the persistent solve on for good)
        du.zero_(); dv.zero_()
        torch.cuda.synchronize()
        pl.run_device(da.data_ptr(), db.data_ptr(), du.data_ptr(), dv.data_ptr(), s)
        torch.cuda.synchronize()
        pl.wait()
        assert pl.last_iterations() == its
        assert np.array_equal(du.cpu().numpy(), good[0]) and np.array_equal(dv.cpu().numpy(), good[1])
    finally:
        pl.tune("persist_fault", 0)
        pl.close(); po.close()


def drill_bands(capi):
    """the row bands run their replicated levels through the same persistent solve.  An abandoned one on any
    band used to go unnoticed (success, silently invalid flow, a sticky abort word); now octane_vof_tiled_wait / _fetch return
    the error, last_iterations() is -2, and the next solve is good again."""
    nx, ny = 640, 500
    a, b = synth.lattice_scene(nx, ny, seed=5)
    prm = capi.FlowParams(kiters=2, liters=1, cgiters=10)
    tp = capi.TiledPlan(nx, ny, 1, prm, nbands=2, devices=capi.band_devices(2), min_band_pixels=200_000)   # 320 x 250 replicated, 640 x 500 banded
    pl = capi.Plan(nx, ny, 1, prm)
    try:
        assert tp.banded_levels == 1
        good = tp.run_host(a, b)
        its = tp.last_iterations()
        pl.tune("persist_fault", 1)                   # the hook is process-wide
        with pytest.raises(capi.OctaneError) as e:
            tp.run_host(a, b)
        assert "resident" in str(e.value) and tp.last_iterations() == -2
        pl.tune("persist_fault", 0)
        again = tp.run_host(a, b)
        assert tp.last_iterations() == its
        assert np.array_equal(good[0], again[0]) and np.array_equal(good[1], again[1])
    finally:
        pl.tune("persist_fault", 0)
        tp.close(); pl.close()



if __name__ == "__main__":
    assert os.path.basename(capi.LIB_PATH) == "liboctane_vof_diag.so", capi.LIB_PATH
    {"host": drill_host, "device": drill_device, "bands": drill_bands}[sys.argv[1]](capi)
    print("DRILL_OK " + sys.argv[1], flush=True)
