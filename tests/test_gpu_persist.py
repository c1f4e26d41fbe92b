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


@pytest.mark.parametrize("slots", [4, 8, 12, 16])
def test_every_slot_count_of_the_persistent_solve(capi, oracle, slots):
    """The kernel is built for 4, 8, 12 and 16 slots of 8 rows per thread; a 448 x 300 level fits all of them (7 x 10, 7 x 5,
    7 x 4 and 7 x 3 sub-domains).  Each against the oracle."""
    nx, ny = 448, 300
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
    L = capi.lib()
    L.octane_selftest_rcp.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
    assert L.octane_selftest_rcp(0, out) == 0
    print(f"PERSIST reciprocal self-test: {out[0]} patterns, {out[1]} mismatches" + (f" (e.g. 0x{out[2]:08x})" if out[1] else ""))
    assert out[0] == 0x7E000000 - 0x01000000
    assert out[1] == 0


def test_a_missing_workgroup_makes_the_solve_give_up_instead_of_hanging(capi):
    """The persistent solve needs all its workgroups resident; one that never shows up (a test hook makes the last workgroup
    leave at once -- what a co-tenant process holding a CU would amount to) must not leave the others spinning: their waits give
    up after 0.25 s, every workgroup leaves, the call returns an error, and the plan works again afterwards."""
    import time
    nx, ny = 640, 500
    a, b = synth.lattice_scene(nx, ny, seed=5)
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(kiters=1, liters=1, cgiters=10))
    try:
        good = pl.run_host(a, b)
        pl.tune("persist_fault", 1)
        t0 = time.perf_counter()
        with pytest.raises(capi.OctaneError) as e:
            pl.run_host(a, b)
        dt = time.perf_counter() - t0
        pl.tune("persist_fault", 0)
        print(f"PERSIST fault drill: error after {dt:.2f} s: {e.value}")
        assert dt < 5.0 and "resident" in str(e.value)
        again = pl.run_host(a, b)
        assert np.array_equal(good[0], again[0]) and np.array_equal(good[1], again[1])
    finally:
        pl.tune("persist_fault", 0)
        pl.close()


def test_stamped_diagnostic_build_gives_the_same_flow_and_counts_its_iterations(capi):
    """pcg_persist_diag.hip is the persistent solve's own source compiled with shader-clock stamps at the seams of an iteration
    (octane_vof_tune(plan, "persist_diag", 1); tools/probe_mid_stamps.py).  Instrumentation must not change a bit of the flow,
    and the counters it leaves (octane_vof_mid_stamps) have to add up: workgroups x iterations at [14] + [30], cycles at every seam."""
    import ctypes as C
    nx, ny, prm = 640, 500, dict(kiters=1, liters=1, cgiters=9)
    a, b = synth.lattice_scene(nx, ny, seed=5)
    L = capi.lib()
    L.octane_vof_mid_stamps.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * 32)()
    up, vp, ip = _run(capi, a, b, prm)
    assert L.octane_vof_mid_stamps(0, buf) == 0            # clear
    try:
        ud, vd, idg = _run(capi, a, b, prm, persist_diag=1)
        assert L.octane_vof_mid_stamps(0, buf) == 0
    finally:
        pl = capi.Plan(64, 64, 1, capi.FlowParams(kiters=1))
        pl.tune("persist_diag", 0)
        pl.close()
    s = list(buf)
    ndiff = int((ud != up).sum() + (vd != vp).sum())
    groups = 10 * 16                                      # 640 x 500 in sub-domains of 64 x 32
    print(f"PERSIST diag: {ndiff} values differ, iterations {ip}/{idg}, stamped workgroup-iterations {s[14]} + {s[30]}")
    assert ndiff == 0 and idg == ip
    assert s[14] + s[30] == groups * ip                   # every workgroup stamped every iteration it ran
    assert all(s[i] + s[16 + i] > 0 for i in (0, 1, 2, 4, 6))
