"""Patch-matching flow (-sosm): the one path whose oracle is PINNED against the reference itself -- the reference's
translation unit is plain C++, builds with g++ from where it lies and produced tests/golden/ref_sosm.npz
(tests/golden/make_ref_sosm_goldens.py)."""
import os

import numpy as np
import pytest

from octane_amd import synth

CASES = ("default", "guess", "r1s3", "r3s1", "r2s0", "r0s2", "flat", "far")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_sosm.npz"))


@pytest.mark.parametrize("name", CASES)
def test_restatement_reproduces_reference_outputs_bitwise(oracle, golden, name):
    rad, srad = (int(x) for x in golden[name + "_prm"])
    u0 = golden[name + "_u0"] if name + "_u0" in golden else None
    v0 = golden[name + "_v0"] if name + "_v0" in golden else None
    u, v = oracle.sosm(golden[name + "_a"], golden[name + "_b"], rad, srad, u0, v0)
    assert np.array_equal(u, golden[name + "_u"]) and np.array_equal(v, golden[name + "_v"])


def test_goldens_say_what_they_should(golden):
    """Sanity of the fixtures themselves.  The result is the winning integer offset plus a sub-pixel correction, so it
    never exceeds srad + 1/2; the scene moves by about (2.5, -1): u saturates at srad = 2, v is resolved.  With a first
    guess the reference compares image 1 AT the guessed position with image 2 around it (both patches move), so the
    guess selects which neighbourhood is tracked rather than bridging a large displacement."""
    for name in CASES:
        srad = int(golden[name + "_prm"][1])
        assert np.abs(golden[name + "_u"]).max() <= srad + 0.5 and np.abs(golden[name + "_v"]).max() <= srad + 0.5
    assert np.median(golden["default_u"][10:-10, 10:-10]) == 2.0
    assert abs(np.median(golden["default_v"][10:-10, 10:-10]) + 1.0) < 0.1
    assert not np.array_equal(golden["guess_u"], golden["default_u"])


def test_spiral_visits_every_window_position_once(oracle):
    import ctypes as C
    L = oracle.lib()
    for srad in (0, 1, 2, 3, 5):
        n = (2 * srad + 1) ** 2
        buf = (C.c_int * (2 * n))()
        L.oct_oracle_sosm_spiral.restype = C.c_int
        count = L.oct_oracle_sosm_spiral(srad, buf)
        pts = [(buf[2 * i], buf[2 * i + 1]) for i in range(count)]
        assert count == n and len(set(pts)) == n and pts[0] == (0, 0)
        assert all(abs(a) <= srad and abs(b) <= srad for a, b in pts)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference sources not on this machine")
@pytest.mark.parametrize("rad,srad", [(2, 2), (1, 2), (3, 3)])
def test_restatement_equals_live_reference_build(oracle, rad, srad):
    oracle.build()
    a, b = (x[0] for x in synth.lattice_scene(140, 90, seed=100 + rad * 7 + srad))
    rng = np.random.RandomState(rad + srad)
    u0 = (3 * rng.randn(90, 140)).astype(np.float32)
    v0 = (3 * rng.randn(90, 140)).astype(np.float32)
    ur, vr = oracle.ref_sosm(a, b, rad, srad, u0, v0)
    uo, vo = oracle.sosm(a, b, rad, srad, u0, v0)
    assert np.array_equal(ur, uo) and np.array_equal(vr, vo)
    uo2, vo2 = oracle.sosm(a, b, rad, srad, u0, v0, flavour="omp")
    assert np.array_equal(ur, uo2) and np.array_equal(vr, vo2)
