"""Round 5 (VERDICT r4 "Next round" item 8, optional): a CROSS-CHECK of the restatement with the reference's own kernel text.

oracle/Makefile `refhip` passes ref src/oct_variational_optical_flow.cu and src/oct_pix2uv_cuda.cu through the image's hipify-perl where
they lie (translated text in a scratch directory outside the repository, the binary under oracle/_ref/, git-ignored) and builds them with hipcc for gfx950; here that
library -- the reference's cooperative kernel, its CSR matrix, its float atomics, its managed memory -- runs on the MI355X next to the HIP
path and the CPU oracle on the same inputs.  It is a TOOL STAND-IN (hipify + hipcc for nvcc, ocml for libdevice, 64-wide wavefronts):
by this build's rules it pins nothing and DESIGN.md keeps saying "parity unpinned"; it is the only independent witness the oracle has.
The reference's dot products add floats with atomicAdd in an order the hardware picks, and hipcc contracts multiply-adds as nvcc does:
agreement is expected at the 1e-6 ... 1e-5 level, asserted below the north-star bar of 1e-4.  Skipped where the library is not there."""
import numpy as np
import pytest

from conftest import rel_l2
from octane_amd import synth

pytestmark = pytest.mark.gpu
BAR = 1e-4


@pytest.fixture(scope="module")
def refhip(oracle):
    if oracle.refhip_lib() is None:
        pytest.skip("oracle/_ref/liboct_ref_hip.so is not there (make -C oracle refhip needs /root/reference)")
    return oracle


@pytest.mark.parametrize("case", ["s1_512", "lattice_1300x1040", "lattice_200x150_nc2", "disc_300x280_single_level"])
def test_reference_kernel_text_agrees_with_oracle_and_hip_path(capi, refhip, case):
    oracle = refhip
    u0 = v0 = None
    if case == "s1_512":            # BASELINE configs[0]
        a, b = synth.gaussian_scene(512, (3.0, -2.0)); a, b = a[None], b[None]
        prm = dict(alpha=5.0, lambda_=1.0)
    elif case == "lattice_1300x1040":
        a, b = synth.lattice_scene(1300, 1040, seed=77)
        prm = dict(kiters=2, liters=1, cgiters=15)
    elif case == "lattice_200x150_nc2":
        a, b = synth.lattice_scene(200, 150, seed=5, nchan=2)
        prm = dict(kiters=3, liters=2, cgiters=12)
    else:
        a, b = synth.disc_scene(300, 280, seed=300 * 3 + 280)
        prm = dict(kiters=1)
    nc, ny, nx = a.shape
    ur, vr = oracle.refhip_flow(a, b, oracle.FlowParams(**prm), u0, v0)
    assert np.isfinite(ur).all() and np.abs(ur).max() > 0, "the hipified reference returned nothing (cooperative launch refused?)"
    g = oracle.REF_GRID_THREADS
    uo, vo, _ = oracle.flow(a, b, oracle.FlowParams(**prm), u0=u0, v0=v0, dot_threads=g)
    uf, vf, _ = oracle.flow(a, b, oracle.FlowParams(**prm), u0=u0, v0=v0, flavour="fma", dot_threads=g)
    ug, vg = capi.flow(a, b, capi.FlowParams(**prm), u0, v0)
    d_ro, d_rf, d_rg, d_go = rel_l2(ur, vr, uo, vo), rel_l2(ur, vr, uf, vf), rel_l2(ur, vr, ug, vg), rel_l2(ug, vg, uo, vo)
    print(f"REFHIP case={case} {nx}x{ny}x{nc} {prm}: reference kernel text (hipified) vs oracle strict {d_ro:.3e}, vs oracle FMA build {d_rf:.3e}, "
          f"vs HIP path {d_rg:.3e}; HIP path vs oracle strict {d_go:.3e}")
    assert d_ro < BAR and d_rf < BAR and d_rg < BAR


def test_reference_navigation_kernel_text_agrees_with_the_three_builds(capi, refhip):
    """pix2uv: the hipified reference kernel (hipcc contracts multiply-adds as nvcc's -fmad=true does) against the library's three builds
    and the oracle's site switches: it has to equal the fused candidates, not the strict one."""
    oracle = refhip
    nx, ny = 500, 300
    rng = np.random.RandomState(0)
    u = (rng.randn(ny, nx) * 3).astype(np.float32)
    v = (rng.randn(ny, nx) * 3).astype(np.float32)
    nav = capi.Nav(pph=35786023.0, req=6378137.0, rpol=6356752.31414, lam0=-75.0 * 3.14159265 / 180.0, xScale=5.6e-05, xOffset=-0.101332,
                   yScale=-5.6e-05, yOffset=0.128212, g2xOffset=-0.101332, g2yOffset=0.128212, minX=100, minY=50, nx=nx, ny=ny)
    nav_o = oracle.Nav()
    for f, _ in capi.Nav._fields_:
        setattr(nav_o, f, getattr(nav, f))
    ref = oracle.refhip_pix2uv(nav_o, 0.0, 300.0, u, v)

    def diff(a, b):
        return int((a[0] != b[0]).sum() + (a[1] != b[1]).sum())
    d = {"hip strict": diff(ref, capi.pix2uv(nav, 0.0, 300.0, u, v, 0, capi.NAV_GEOS)),
         "hip float sites (the shim)": diff(ref, capi.pix2uv(nav, 0.0, 300.0, u, v, 0, capi.NAV_GEOS | capi.NAV_FMAD_FLOAT)),
         "hip all fused": diff(ref, capi.pix2uv(nav, 0.0, 300.0, u, v, 0, capi.NAV_GEOS | capi.NAV_FMAD)),
         "oracle strict": diff(ref, oracle.pix2uv(nav_o, 0.0, 300.0, u, v)),
         "oracle float sites": diff(ref, oracle.pix2uv(nav_o, 0.0, 300.0, u, v, sites=oracle.P2U_FLOAT_SITES)),
         "oracle all sites": diff(ref, oracle.pix2uv(nav_o, 0.0, 300.0, u, v, sites=oracle.P2U_ALL_SITES))}
    print(f"REFHIP pix2uv, {2 * nx * ny} shorts, hipified reference kernel against: {d}")
    assert ref[4] == 300.0
    assert d["hip float sites (the shim)"] <= 2 and d["hip all fused"] <= 2 and d["oracle float sites"] <= 2
    assert d["hip strict"] > 100 and d["oracle strict"] > 100
    assert np.array_equal(ref[2], (100 * u).astype(np.int16)) and np.array_equal(ref[3], (100 * v).astype(np.int16))
