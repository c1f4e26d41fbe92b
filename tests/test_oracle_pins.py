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
    from conftest import SANITIZE
    # (the -O1 sanitizer builds of `make sanitize` may leave the products uncontracted: the two flavours can then coincide)
    assert (0 < d or SANITIZE) and d < 1e-5


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


# ---- the warp's bilinear weights and the clamp, against the reference's own plain-C++ code -----------------------------------
@pytest.fixture(scope="module")
def golden_bil():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_binterp_bc.npz"))


def test_clamp_matches_reference_oct_bc(oracle, golden_bil):
    """ref include/oct_bc.h:1-20 (oct_bc<T>, compiled from the reference into oracle/_ref; goldens by
    tests/golden/make_ref_binterp_goldens.py) is the CPU twin of the device clamp oct_bc_cu (.cu:26-41): clamp -- not
    reflect -- to [0, n-1] and report whether it clamped.  The oracle's clamp_coord must agree bit for bit, flag included."""
    L = oracle.lib()
    hit = C.c_int()
    xs, ns = golden_bil["bc_x"], golden_bil["bc_n"]
    got = np.array([(L.oct_oracle_clamp_coord(float(x), int(n), C.byref(hit)), hit.value) for x, n in zip(xs, ns)])
    assert np.array_equal(got[:, 0].astype(np.float32), golden_bil["bc_float"])
    assert np.array_equal(got[:, 1].astype(np.int32), golden_bil["bc_float_hit"])
    # the double and int instantiations say the same thing (the quirk is in the comparison x >= nx, not in the type)
    assert np.array_equal(got[:, 0], golden_bil["bc_double"])
    assert np.array_equal(got[:, 1].astype(np.int32), golden_bil["bc_double_hit"])
    gi = np.array([(L.oct_oracle_clamp_coord(float(x), int(n), C.byref(hit)), hit.value) for x, n in zip(golden_bil["bc_xi"], golden_bil["bc_ni"])])
    assert np.array_equal(gi[:, 0].astype(np.int32), golden_bil["bc_int"]) and np.array_equal(gi[:, 1].astype(np.int32), golden_bil["bc_int_hit"])
    assert golden_bil["bc_float_hit"].sum() > 20 and (golden_bil["bc_float_hit"] == 0).sum() > 100     # both branches exercised


def test_bilinear_warp_matches_reference_oct_binterp(oracle, golden_bil):
    """ref src/oct_binterp.cc:24-41 (oct_binterp_coefs, oct_coef_binterp; double) is the CPU twin of the device warp's
    oct_binterp_coefs_cu / oct_coef_binterp_cu (.cu:56-71; float).  On float-representable positions the double weights are
    exact, so the oracle's float weights must equal them to float rounding -- bit for bit wherever Sterbenz' lemma makes
    the float subtraction exact (cells >= 1) -- and the interpolated value must be the double one within float rounding of
    a four-term sum.  Cell selection (last cell capped at n-2, .cu:738-745) is checked against the generator's rule."""
    L = oracle.lib()
    nx, ny = int(golden_bil["bil_nx"]), int(golden_bil["bil_ny"])
    p4 = (C.c_float * 4)(); cell = (C.c_int * 2)(); hit = (C.c_int * 2)()
    px, py, f = golden_bil["bil_px"], golden_bil["bil_py"], golden_bil["bil_f"]
    want_p, want_v, want_v2 = golden_bil["bil_p"], golden_bil["bil_val"], golden_bil["bil_val_reused"]
    nexact = 0
    for k in range(len(px)):
        v = L.oct_oracle_bilinear(float(px[k]), float(py[k]), nx, ny, *(float(t) for t in f[k]), p4, cell, hit)
        assert (cell[0], cell[1]) == (int(golden_bil["bil_x0"][k]), int(golden_bil["bil_y0"][k])), k
        assert (hit[0], hit[1]) == (0, 0), k
        p = np.array(list(p4), np.float64)
        np.testing.assert_allclose(p, want_p[k], rtol=0, atol=6e-8, err_msg=str(k))      # float subtraction near 1 - tiny
        if cell[0] >= 1 and cell[1] >= 1:
            assert np.array_equal(p.astype(np.float32), want_p[k].astype(np.float32)), k
            nexact += 1
        assert abs(p[0] + p[1] - 1.0) < 2e-7 and abs(p[2] + p[3] - 1.0) < 2e-7
        assert abs(v - want_v[k]) <= 4 * np.spacing(np.float32(255.0)), (k, v, want_v[k])
        v2 = L.oct_oracle_bilinear(float(px[k]), float(py[k]), nx, ny, *(float(t) for t in f[k, ::-1]), None, None, None)
        assert abs(v2 - want_v2[k]) <= 4 * np.spacing(np.float32(255.0)), k
    assert nexact > 500
    # beyond the level the position is clamped and flagged (the assembly then zeroes the derivatives, .cu:768-779)
    v = L.oct_oracle_bilinear(-2.5, float(ny) + 3.0, nx, ny, 1.0, 2.0, 3.0, 4.0, p4, cell, hit)
    assert (hit[0], hit[1]) == (1, 1) and (cell[0], cell[1]) == (0, ny - 2) and v == 3.0


def test_reference_binterp_and_bc_goldens_are_current(oracle, golden_bil):
    """Where the reference is present (this container, not the GPU box): the committed goldens are what its code returns now."""
    path = oracle.ref_helpers_path()
    if not os.path.exists(path) or not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference build here")
    R = C.CDLL(path)
    R.oct_ref_bc_float.restype = C.c_float
    R.oct_ref_bc_float.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_int)]
    R.oct_ref_binterp_coefs.restype = C.c_double
    R.oct_ref_binterp_coefs.argtypes = [C.c_double] * 10 + [C.POINTER(C.c_double)]
    hit = C.c_int()
    got = np.array([R.oct_ref_bc_float(float(x), int(n), C.byref(hit)) for x, n in zip(golden_bil["bc_x"], golden_bil["bc_n"])], np.float32)
    assert np.array_equal(got, golden_bil["bc_float"])
    buf = (C.c_double * 4)()
    for k in (0, 5, 100, 599):
        x0, y0 = float(golden_bil["bil_x0"][k]), float(golden_bil["bil_y0"][k])
        v = R.oct_ref_binterp_coefs(float(golden_bil["bil_px"][k]), float(golden_bil["bil_py"][k]), x0, x0 + 1, y0, y0 + 1,
                                    *(float(t) for t in golden_bil["bil_f"][k]), buf)
        assert v == golden_bil["bil_val"][k]


def test_band_range_table_equals_the_reference_table(capi):
    """`octane_bandminmax` (the C-ABI) and the C++ shim `oct_bandminmax` (liboctane_host.so, the reference's
    own signature, ref src/oct_normalize_geo.cc:9) against the WHOLE table dumped from the reference source itself
    (tests/golden/ref_bandminmax.npz, made by tests/golden/make_ref_bandminmax_goldens.py from oracle/_ref): bands 1 .. 16 bit
    for bit -- the reference assigns double literals to floats --, and every band number outside 1 .. 16 leaves the caller's
    values untouched in the shim (the reference's if-chain has no else) and is an error code in the C-ABI.  No GPU involved."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_bandminmax.npz"))
    sentinel = np.float32(g["sentinel"])
    from conftest import host_libdir
    host = os.path.join(host_libdir(), "liboctane_host.so")
    shim = None
    if os.path.exists(host):
        capi.lib()                                   # liboctane_vof.so first: the shim links against it
        shim = C.CDLL(host)._Z14oct_bandminmaxiRfS_   # void oct_bandminmax(int, float &, float &)
        shim.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        shim.restype = None
    known = 0
    for band, mx, mn in zip(g["bands"], g["maxch"], g["minch"]):
        untouched = mx == sentinel and mn == sentinel
        if untouched:
            with pytest.raises(capi.OctaneError) as e:
                capi.bandminmax(int(band))
            assert e.value.code == capi.E_INVALID
        else:
            known += 1
            got = capi.bandminmax(int(band))
            assert np.float32(got[0]).tobytes() == np.float32(mx).tobytes() and np.float32(got[1]).tobytes() == np.float32(mn).tobytes(), (band, got, mx, mn)
        if shim is not None:
            a, b = C.c_float(float(sentinel)), C.c_float(float(sentinel))
            shim(int(band), C.byref(a), C.byref(b))
            assert np.float32(a.value).tobytes() == np.float32(mx).tobytes() and np.float32(b.value).tobytes() == np.float32(mn).tobytes(), (band, a.value, b.value)
    assert known == 16 and [int(b) for b, m in zip(g["bands"], g["maxch"]) if m != sentinel] == list(range(1, 17))
    # re-checked against the live reference build wherever /root/reference exists
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "liboct_ref_helpers.so")
    if os.path.isdir("/root/reference/src") and os.path.exists(ref):
        R = C.CDLL(ref)
        R.oct_ref_bandminmax.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        R.oct_ref_bandminmax.restype = None
        for band, mx, mn in zip(g["bands"], g["maxch"], g["minch"]):
            a, b = C.c_float(float(sentinel)), C.c_float(float(sentinel))
            R.oct_ref_bandminmax(int(band), C.byref(a), C.byref(b))
            assert a.value == mx and b.value == mn


def test_float_reciprocal_through_double_is_the_float_division(oracle):
    """The reference forms 1 / M and the psi' functions as (float)(1. / (double)y) (ref .cu:80,141-149); the HIP path uses a float
    reciprocal (rcp_exact: v_rcp_f32 + one fused Newton step, compared with 1.0f / y on every float by octane_selftest_rcp).  The two
    are the same number for EVERY positive normal float: a quotient of two 24-bit significands rounded to 53 bits and then to 24
    rounds as it would at once (53 >= 2 * 24 + 2).  Checked here by brute force on the host -- all 2 130 706 432 of them, a few
    seconds with OpenMP -- with the non-inlined C of oracle/vof_oracle.c, so round 2's "1 ulp with probability 2^-29" caveat
    about the preconditioner is gone: the probability is zero."""
    L = oracle.lib("omp")
    L.oct_oracle_check_reciprocal_double_rounding.restype = C.c_longlong
    L.oct_oracle_check_reciprocal_double_rounding.argtypes = [C.POINTER(C.c_longlong)]
    n = C.c_longlong()
    bad = L.oct_oracle_check_reciprocal_double_rounding(C.byref(n))
    print(f"(float)(1. / (double)y) vs 1.0f / y on {n.value} positive normal floats: {bad} mismatches")
    assert n.value == 0x7F800000 - 0x00800000 and bad == 0
