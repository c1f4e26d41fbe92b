"""Channel resampling of the readers (octane_amd/csrc/zoom_host.cpp, ref src/oct_zoom.cc) against the reference's own
functions: tests/golden/ref_helpers.npz holds inputs and outputs of oct_zoom_out_float / oct_zoom_in_float compiled
unmodified from the reference (tests/golden/make_golden_ref.py).  Bit-exact.  Host code, no GPU."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def _lib():
    from octane_amd import capi
    capi.lib()                                       # liboctane_host.so links the flow library
    import subprocess
    from conftest import host_libdir, host_make_args
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "octane_amd", "csrc"), "-s", "-f", "Makefile.host", *host_make_args()])
    L = C.CDLL(os.path.join(host_libdir(), "liboctane_host.so"))
    zo = L._Z18oct_zoom_out_floatPfS_iidii
    zo.argtypes = [F, F, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
    zi = L._Z17oct_zoom_in_floatPfS_iiiiii
    zi.argtypes = [F, F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    zs = L._Z13oct_zoom_sizeiiRiS_d
    zs.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_double]
    return zo, zi, zs


def test_zoom_out_float_is_the_references_bit_for_bit(golden_ref):
    zo, _, zs = _lib()
    img = np.ascontiguousarray(golden_ref["zof_img"])
    ny, nx = img.shape
    for tag, f in (("05", 0.5), ("04", 0.4), ("10", 1.0)):
        want = golden_ref["zof_out_" + tag]
        a, b = C.c_int(), C.c_int()
        zs(nx, ny, C.byref(a), C.byref(b), f)
        assert (b.value, a.value) == want.shape
        got = np.zeros_like(want)
        zo(img, got, nx, ny, f, 0, 0)
        assert np.array_equal(got, want), tag


def test_zoom_out_float_writes_the_channels_own_plane(golden_ref):
    """The reference adds `cnum` instead of cnum * plane (oct_zoom.cc:85); here channel 1 lands in plane 1 and plane 0
    is left alone (documented difference; channel 0 is identical, see above)."""
    zo, _, _ = _lib()
    img = np.ascontiguousarray(golden_ref["zof_img"])
    ny, nx = img.shape
    want = golden_ref["zof_out_04"]
    two = np.full((2,) + want.shape, -7.0, np.float32)
    zo(img, two, nx, ny, 0.4, 0, 1)
    assert np.array_equal(two[1], want) and (two[0] == -7.0).all()


def test_zoom_in_float_is_the_references_bit_for_bit(golden_ref):
    _, zi, _ = _lib()
    src = np.ascontiguousarray(golden_ref["zif_src"])
    cy, cx = src.shape
    want = golden_ref["zif_out_c1"]
    got = np.full_like(want, -1.0)
    zi(src, got, cx, cy, want.shape[2], want.shape[1], 1, 1)
    assert np.array_equal(got, want)
    near = golden_ref["zif_near"]
    got = np.zeros_like(near)
    zi(src, got, cx, cy, near.shape[1], near.shape[0], 0, 0)
    assert np.array_equal(got, near)
    # and the flow up-sampling case the solver's goldens already hold (odd sizes, factor not 2)
    flow = np.ascontiguousarray(golden_ref["zi_flow"])
    up = golden_ref["zi_out"]
    got = np.zeros_like(up)
    zi(flow, got, flow.shape[1], flow.shape[0], up.shape[1], up.shape[0], 0, 1)
    assert np.array_equal(got, up)
