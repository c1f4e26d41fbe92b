"""Pins of the CPU oracle (oracle/vof_oracle.c) against what exists of the reference:

* golden vectors produced by the reference's OWN CPU helper functions (oct_bicubic.cc,
  oct_gaussian.cc, oct_zoom.cc compiled unmodified -> tests/golden/ref_helpers.npz, script
  tests/golden/make_ref_goldens.py); these share their formulas with the CUDA hot path;
* the interior-mean flows recorded from the reference solver on the S1 scene
  (BASELINE.md 2).

The whole-solver parity is otherwise unpinned (the reference ships no tests).
"""
import ctypes as C
import os

import numpy as np
import pytest

from octane_amd import synth


def test_bicubic_matches_reference_helper(oracle, golden_ref):
    img = np.ascontiguousarray(golden_ref["bic_img"])
    ny, nx = img.shape
    L = oracle.lib()
    got = np.array([L.oct_oracle_bicubic(img, float(u), float(v), nx, ny) for u, v in golden_ref["bic_pts"]])
    want = golden_ref["bic_vals"]          # reference evaluates in double on the same float image
    assert np.abs(got - want).max() <= 4e-5 * max(1.0, np.abs(want).max())
    # includes the quirky points: (-0.25,-0.75) and coordinates beyond the last pixel
    assert len(want) >= 300


def test_gaussian_taps_match_reference_helper(oracle, golden_ref):
    L = oracle.lib()
    for m in range(1, 9):
        f = np.float32(0.5 ** m)
        fs = L.oct_oracle_blur_halfwidth(f)
        want = golden_ref[f"taps_{m}"]
        assert len(want) == 2 * fs + 1, "window rule fs = max(5, (int)(2/sqrt(2f)))"
        gk = np.zeros(2 * fs + 1, np.float32)
        L.oct_oracle_gauss_taps(f, fs, gk)
        np.testing.assert_allclose(gk, want, rtol=3e-6, atol=0)
        assert abs(gk.sum() - 1.0) < 1e-5          # normalised over ALL 2fs+1 taps ...
        assert gk[:-1].sum() < 1.0                 # ... although only 2fs of them are applied


def test_blur_decimate_matches_reference_helper(oracle, golden_ref):
    img = np.ascontiguousarray(golden_ref["zo_img"])
    ny, nx = img.shape
    L = oracle.lib()
    f = np.float32(0.5)
    fs = L.oct_oracle_blur_halfwidth(f)
    gk = np.zeros(2 * fs + 1, np.float32)
    L.oct_oracle_gauss_taps(f, fs, gk)
    t1 = np.zeros_like(img)
    t2 = np.zeros_like(img)
    L.oct_oracle_blur_rows(img, t1, gk, nx, ny, 1, fs)
    L.oct_oracle_blur_cols(t1, t2, gk, nx, ny, 1, fs)
    lx, ly = oracle.level_dims(nx, ny, f)
    want = golden_ref["zo_out"]
    assert want.shape == (ly, lx)
    dec = np.zeros((ly, lx), np.float32)
    L.oct_oracle_decimate(t2, dec, nx, ny, 1, f)
    np.testing.assert_allclose(dec, want, rtol=2e-6, atol=2e-4)


def test_flow_upsample_matches_reference_helper(oracle, golden_ref):
    flow = np.ascontiguousarray(golden_ref["zi_flow"])
    want = golden_ref["zi_out"]
    cy, cx = flow.shape
    fy, fx = want.shape
    up = np.zeros((fy, fx), np.float32)
    # sf = 1 so only the resampling is compared (the device version divides by scaleF)
    oracle.lib().oct_oracle_upsample_flow(flow, up, cx, cy, fx, fy, 1.0)
    # the CPU helper keeps the four column cubics in double; the device formula (and the
    # oracle) rounds them to float in between: a few float ulps of the |flow| <= 8 values
    np.testing.assert_allclose(up, want, rtol=0, atol=5e-6)


# (n, true shift, recorded interior-mean flow of the REFERENCE solver) -- BASELINE.md section 2.
# "interior" is the central half of the frame; defaults alpha=5 lambda=1 kiters=4 liters=3 cgiters=30.
RECORDED = [
    (64, (1.5, -0.75), (1.4983, -0.7498)),
    (128, (2.0, 1.0), (2.0000, 1.0000)),
    (512, (3.0, -2.0), (3.0004, -2.0010)),
]


@pytest.mark.parametrize("n,shift,recorded", RECORDED)
def test_solver_reproduces_recorded_reference_answers(oracle, n, shift, recorded):
    a, b = synth.gaussian_scene(n, shift)
    u, v, its = oracle.flow(a, b)
    m = n // 4
    got = (u[m:n - m, m:n - m].mean(), v[m:n - m, m:n - m].mean())
    assert abs(got[0] - recorded[0]) < 1.5e-4 and abs(got[1] - recorded[1]) < 1.5e-4, got
    if n >= 128:
        assert its == 4 * 3 * 3 * 30     # the tolerance exit never fires


def test_fma_flavour_is_within_the_noise_floor(oracle):
    """FMA-contracted vs non-contracted builds of the reference differ by
    1.8e-6 .. 2.9e-6 relative L2; the two oracle flavours must show the same order."""
    from conftest import rel_l2
    a, b = synth.gaussian_scene(128, (2.0, 1.0))
    u, v, _ = oracle.flow(a, b)
    uf, vf, _ = oracle.flow(a, b, flavour="fma")
    d = rel_l2(uf, vf, u, v)
    assert 0 < d < 1e-5


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference sources not on this machine")
def test_live_reference_helper_agrees_with_committed_goldens(oracle, golden_ref):
    """Where the reference is present, rebuild oracle/_ref from it and re-check one golden."""
    oracle.build()
    R = C.CDLL(oracle.ref_helpers_path())
    f = R._Z17oct_bicubic_floatPfddiii
    f.restype = C.c_double
    f.argtypes = [np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"), C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    img = np.ascontiguousarray(golden_ref["bic_img"])
    ny, nx = img.shape
    for (u, v), want in list(zip(golden_ref["bic_pts"], golden_ref["bic_vals"]))[:50]:
        assert f(img, float(u), float(v), nx, ny, 1) == want
