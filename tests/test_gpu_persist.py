"""The whole-solve-in-one-launch PCG of the mid-size pyramid levels (octane_amd/csrc/pcg_persist.hip: the level resident in
registers and LDS, one workgroup per sub-domain, one grid barrier per iteration) against

* its own per-launch ("stepped") form -- the same kernel running 1 or 7 iterations per launch with the complete state in the
  level's planes, so that the kernel boundary provides the visibility between workgroups: BIT FOR BIT.  A stale read through
  the in-launch hand-off (edge pixels and partial sums of the neighbouring workgroups) would show up here;
* the one-launch-per-iteration kernels it replaces (k_pcg_fused: other tiles, another grouping of the fp64 partial sums):
  within 1e-6, equal iteration counts;
* the oracle (ref src/oct_variational_optical_flow.cu:1105-1195).
"""
import numpy as np
import pytest

from conftest import rel_l2
from octane_amd import synth

pytestmark = pytest.mark.gpu


def _run(capi, a, b, prm, u0=None, v0=None, **knobs):
    nc, ny, nx = (1,) + a.shape if a.ndim == 2 else a.shape
    if knobs:            # tuning knobs exist in the diagnostic library only (the product's sources + -DOCTANE_DIAG=1); plain runs stay on the product
        try:
            capi = capi.diag()
        except ImportError as e:
            pytest.skip(str(e))
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    try:
        for k, v in knobs.items():
            pl.tune(k, v)
        u, v = pl.run_host(a, b, u0, v0)
        its = pl.last_iterations()
    finally:
        pl.close()
    return u, v, its


@pytest.mark.parametrize("nx,ny,prm", [
    (640, 500, dict(kiters=1, liters=2, cgiters=9)),       # 10 x 16 sub-domains of 64 x 32: every kind of neighbour
    (333, 257, dict(kiters=2, liters=1, cgiters=12)),      # ragged last column (13 px) and a last sub-domain of one row
    (1000, 1000, dict(kiters=1, liters=1, cgiters=6)),     # 256 workgroups, one per CU
    (1250, 1250, dict(kiters=1, liters=1, cgiters=5)),     # R1's largest resident level: 240 sub-domains of 64 x 112
])
def test_persistent_solve_equals_its_per_launch_form_bit_for_bit(capi, nx, ny, prm):
    a, b = synth.lattice_scene(nx, ny, seed=nx + ny)
    up, vp, ip = _run(capi, a, b, prm)
    assert np.isfinite(up).all() and np.isfinite(vp).all()
    for step in (1, 7):
        us, vs, i_s = _run(capi, a, b, prm, persist_step=step)
        ndiff = int((us != up).sum() + (vs != vp).sum())
        print(f"PERSIST {nx}x{ny} {prm}: one launch vs {step} iteration(s) per launch: {ndiff} values differ, iterations {ip}/{i_s}")
        assert i_s == ip
        assert ndiff == 0, f"{ndiff} values differ between the persistent solve and {step} iteration(s) per launch"


@pytest.mark.parametrize("nx,ny,nc,prm", [
    (640, 500, 1, dict(kiters=3, liters=2, cgiters=20)),
    (333, 257, 2, dict(kiters=2, liters=2, cgiters=15)),
    (1250, 1250, 1, dict(kiters=2, liters=1, cgiters=10)),
])
def test_persistent_solve_equals_one_launch_per_iteration(capi, nx, ny, nc, prm):
    a, b = synth.lattice_scene(nx, ny, seed=3 * nx + ny, nchan=nc)
    up, vp, ip = _run(capi, a, b, prm)
    uf, vf, i_f = _run(capi, a, b, prm, persist=0)
    d = rel_l2(up, vp, uf, vf)
    print(f"PERSIST {nx}x{ny}x{nc} {prm}: persistent vs one launch per iteration relL2 {d:.3e}, iterations {ip}/{i_f}")
    assert ip == i_f
    assert d < 1e-5          # two groupings of the fp64 partial sums over hundreds of iterations (measured 3e-6)


@pytest.mark.parametrize("slots", [1, 2, 4, 6, 8, 10, 12, 14, 16])
def test_every_slot_count_of_the_persistent_solve(capi, oracle, slots):
    """The kernel is built for 1, 2, 4, 6 ... 16 slots of 8 rows per thread; a 448 x 300 level fits those from 4 on (7 x 10 ... 7 x 3
    sub-domains), a 320 x 200 level the one- and two-slot forms (5 x 25 and 5 x 13: they are limited to 128 workgroups).  From 12
    slots on the neighbour weights come from the L2 workspace instead of registers.  Each against the oracle."""
    nx, ny = (448, 300) if slots >= 4 else (320, 200)
    a, b = synth.lattice_scene(nx, ny, seed=71)
    prm = dict(kiters=1, liters=2, cgiters=14)
    uo, vo, io = oracle.flow(a, b, oracle.FlowParams(**prm), dot_threads=oracle.REF_GRID_THREADS)
    u, v, its = _run(capi, a, b, prm, persist_p=slots)
    d = rel_l2(u, v, uo, vo)
    print(f"PERSIST {nx}x{ny} slots={slots}: relL2 vs oracle {d:.3e}, iterations {its}/{io}")
    assert its == io
    assert d < 2e-5


def test_persistent_solve_stops_where_the_reference_stops(capi, oracle):
    """A solve that meets the tolerance before the cap (ref .cu:1131): every workgroup has to leave the loop in the same
    iteration, and the pending x update has to be applied."""
    nx, ny = 300, 200
    a, b = synth.lattice_scene(nx, ny, seed=13)
    b = (a + 0.02 * (b - a)).astype(np.float32)            # almost identical images: tiny right-hand side
    prm = dict(kiters=1, liters=2, cgiters=200)
    uo, vo, io = oracle.flow(a, b, oracle.FlowParams(**prm), dot_threads=oracle.REF_GRID_THREADS)
    u, v, its = _run(capi, a, b, prm)
    uf, vf, i_f = _run(capi, a, b, prm, persist=0)
    print(f"PERSIST early stop: iterations oracle {io}, persistent {its}, per-launch {i_f} (cap {2 * 3 * 200})")
    assert io < 2 * 3 * 200, "the case is meant to stop on the tolerance"
    assert its == i_f
    assert abs(its - io) <= 2                               # the stop test sits on float sums: the last iteration may differ
    assert rel_l2(u, v, uf, vf) < 1e-5


def test_identical_images_persistent(capi):
    a, _ = synth.lattice_scene(400, 300, seed=3)
    u, v, its = _run(capi, a, a, dict(kiters=1))
    assert its == 0 and not u.any() and not v.any()


def test_three_instruction_reciprocal_equals_the_division_everywhere(capi):
    """pcg_persist.hip forms 1 / diagonal as v_rcp_f32 + one fused Newton step.  It has to be the correctly rounded quotient --
    what `1.0f / x` gives -- for every x it can meet: checked on ALL positive normal floats whose reciprocal is normal."""
    import ctypes as C
    out = (C.c_ulonglong * 3)()
    try:
        L = capi.diag().lib()       # the self-test's export lives in the diagnostic library (include/octane_vof_dev.h); same device code
    except ImportError as e:
        pytest.skip(str(e))
    assert L.octane_selftest_rcp(0, out) == 0
    print(f"PERSIST reciprocal self-test: {out[0]} patterns, {out[1]} mismatches" + (f" (e.g. 0x{out[2]:08x})" if out[1] else ""))
    assert out[0] == 0x7E000000 - 0x01000000
    assert out[1] == 0


def _fault_drill(capi, mode):
    """Runs one drill of tests/persist_fault_worker.py in a child process on the diagnostic library (the only one with the hook)."""
    import os
    import subprocess
    import sys
    if not os.path.exists(capi.DIAG_LIB_PATH):
        pytest.skip("liboctane_vof_diag.so has not been built (make -C octane_amd/csrc DIAG=1)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "persist_fault_worker.py"), mode], env=dict(os.environ, OCTANE_LIB=capi.DIAG_LIB_PATH),
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and ("DRILL_OK " + mode) in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def test_a_missing_workgroup_makes_the_solve_give_up_instead_of_hanging(capi):
    """The persistent solve needs all its workgroups resident; one that never shows up (the diagnostic library's hook makes the last
    workgroup leave at once -- what a co-tenant process holding a CU would amount to) must not leave the others spinning: their waits
    give up after 0.25 s and every workgroup leaves.  Host-buffer call: the library notices, says so once on stderr, solves the pair
    AGAIN with one launch per iteration and returns that flow -- bit-equal to a plan with the persistent solve switched off.  Round 4
    (ADVICE r3): the plan stays in that mode for its next 16 runs, not for life, and says so through octane_vof_plan_persist_state;
    then the persistent solve is back, with the bits it had before.  (Body: tests/persist_fault_worker.py drill_host.)"""
    r = _fault_drill(capi, "host")
    assert "solving the pair again" in r.stderr


def test_an_abandoned_solve_on_device_buffers_is_repaired_by_the_first_call_that_asks(capi):
    """a caller that drives the plan with device buffers and its own stream synchronisation (bench.py, torch
    users).  The first of octane_vof_plan_wait / octane_vof_plan_last_iterations called after the caller has synchronised makes the
    abandoned run again -- one launch per iteration, from the plan's own copy of the inputs, on the run's stream, into the run's
    output buffers -- and reports the repaired run.  (Body: tests/persist_fault_worker.py drill_device.)"""
    _fault_drill(capi, "device")


def test_an_abandoned_solve_fails_the_row_band_solve_too(capi):
    """The row bands run their replicated levels through the same persistent solve: octane_vof_tiled_wait / _fetch return the error,
    last_iterations() is -2, and the next solve is good again.  (Body: tests/persist_fault_worker.py drill_bands.)"""
    _fault_drill(capi, "bands")


def test_the_product_library_has_no_fault_hook(capi):
    """the product library has no tuning knob at all (octane_vof_tune is not exported), let alone the drill's."""
    assert not hasattr(capi.lib(), "octane_vof_tune")
    pl = capi.Plan(64, 64, 1, capi.FlowParams(kiters=1))
    try:
        with pytest.raises(AttributeError):
            pl.tune("persist_fault", 1)
    finally:
        pl.close()


def test_stamped_diagnostic_build_gives_the_same_flow_and_counts_its_iterations(capi):
    """pcg_persist_diag.hip is the persistent solve's own source compiled with shader-clock stamps at the seams of an iteration.
    Since round 3 it lives in a library of its own (liboctane_vof_diag.so, `make -C octane_amd/csrc DIAG=1`; the product library has
    neither the stamped copies nor their exports), so this runs in a child process that loads that library (tests/diag_worker.py).
    Instrumentation must not change a bit of the flow -- the stamped build, the unstamped build of the same library and the PRODUCT
    library all give the same bits -- and the counters it leaves (octane_vof_mid_stamps) have to add up."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    if not os.path.exists(capi.DIAG_LIB_PATH):
        pytest.skip("liboctane_vof_diag.so has not been built (make -C octane_amd/csrc DIAG=1)")
    nx, ny, prm = 640, 500, dict(kiters=1, liters=1, cgiters=9)
    a, b = synth.lattice_scene(nx, ny, seed=5)
    up, vp, ip = _run(capi, a, b, prm)                     # the product library, in this process
    h = hashlib.sha1(up.tobytes()); h.update(vp.tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "diag_worker.py")], env=dict(os.environ, OCTANE_LIB=capi.DIAG_LIB_PATH),
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("DIAG_RESULT ")]
    assert r.returncode == 0 and line, r.stdout + r.stderr
    d = json.loads(line[0][12:])
    s = d["stamps"]
    groups = 10 * 16                                      # 640 x 500 in sub-domains of 64 x 32
    print(f"PERSIST diag: product / diagnostic library / stamped build flows {h.hexdigest()[:12]} / {d['plain'][:12]} / {d['diag'][:12]}, "
          f"iterations {ip}/{d['its_plain']}/{d['its_diag']}, stamped workgroup-iterations {s[14]} + {s[30]}")
    assert d["plain"] == d["diag"] == h.hexdigest() and d["its_plain"] == d["its_diag"] == ip
    assert s[14] + s[30] == groups * ip                   # every workgroup stamped every iteration it ran
    assert all(s[i] + s[16 + i] > 0 for i in (0, 1, 2, 4, 6))
