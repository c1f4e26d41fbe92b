"""The assembly kernel's fast exact forms (round 3; k_assemble, ref .cu:611-1097).  The reference divides by alpha five or six times
per pixel, takes 1 / (s + 1) three times per channel and 1 / sqrt(x + 1e-6) twice -- in double, rounded to float afterwards.  An IEEE
fp64 division costs ~14 instructions on gfx950, a third of the kernel's issue time.  The library replaces them by short sequences
(x * (1 / alpha) + one exact residual step; v_rcp_f64 / v_rsq_f64 + two Newton steps) ONLY where the sequence gives the bits of the
IEEE result on every float input -- which a device self-test establishes by trying every one of them, per alpha, before a plan may
use the form."""
import ctypes as C
import os

import numpy as np
import pytest

from octane_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alpha", [5.0, 12.0, 8.0, 3.0, 0.7])
def test_fast_forms_are_used_only_where_they_are_exact_on_every_float(capi_diag, alpha):
    capi = capi_diag      # the self-tests' exports live in the diagnostic library (include/octane_vof_dev.h); plan creation runs the same check in the product
    L = capi.lib()
    out = (C.c_ulonglong * 8)()
    assert L.octane_selftest_assembly_math(0, alpha, out) == 0
    bits = L.octane_selftest_assembly_math_bits(0, alpha)
    n_div, bad_div, n_rcp, bad_rcp, n_rsq, bad_rsq, first, which = list(out)
    print(f"ASM-MATH alpha={alpha}: x/alpha {bad_div} mismatches of {n_div}; 1/(s+1) {bad_rcp} of {n_rcp}; 1/sqrt(x+1e-6) {bad_rsq} of {n_rsq}"
          + (f" (e.g. 0x{first:08x} in test {which})" if bad_div + bad_rcp + bad_rsq else "") + f"; forms in use: {bits:03b}")
    assert n_div == 2 * (0x7F800000) + 2 and n_rcp == n_rsq == 0x7F800000 + 1  # every float but the NaNs / every float >= 0, the infinities included (round 4)
    assert bits >= 0
    for bit, bad in ((1, bad_div), (2, bad_rcp), (4, bad_rsq)):
        assert bool(bits & bit) == (bad == 0), (bit, bad)
    if alpha == 5.0:
        assert bits & 1, "the three-instruction division by the default alpha must be exact"


def test_fast_forms_do_not_change_a_bit_of_the_flow(capi_diag):
    """Same pair, the assembly with and without the fast forms (tune("asm_fast", 0): IEEE divisions throughout), all three GNC steps,
    Zimmer and Brox data terms: the flows have to be the same bits (the coefficient planes are compared with the oracle bit for bit in
    test_gpu_parity.py, which runs with the fast forms on)."""
    capi = capi_diag      # tune("asm_fast") exists in the diagnostic library only
    nx, ny = 333, 217
    a, b = synth.lattice_scene(nx, ny, seed=44)
    for prm in (dict(kiters=3, liters=2, cgiters=10), dict(kiters=2, liters=1, cgiters=8, dozim=0, alpha=12.0, lambda_=0.25)):
        outs = []
        for fast in (1, 0):
            pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
            try:
                pl.tune("asm_fast", fast)
                outs.append(pl.run_host(a, b))
            finally:
                pl.close()
        assert np.isfinite(outs[0][0]).all()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), prm
