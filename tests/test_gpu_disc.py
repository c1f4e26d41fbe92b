"""Round 5 (VERDICT r4 "Next round" item 1): solver parity on inputs shaped like the data the path exists for.

synth.disc_scene is what ref src/oct_navcal_cuda.cu:29-93 hands the solver from a full-disk file: the Earth disc on a background of
exact zeros, the limb taper (subpoint distance 0.021 ... 0.0212 rad^2), radiances that went through int16 counts (plateaus of equal
values: exactly-zero gradients), sensor noise, a saturated patch, 1 - 3 channels.  What these planes exercise that the smooth
families never did: the `0 * x` shortcuts and the fast fp64 forms of k_assemble on exact zeros (ref .cu:657-724: psi' on flat
regions), border-free interior tiles crossing the disc edge, the warp clamps next to the limb with a non-zero first guess
(ref .cu:732-779).

Two kinds of assertion:

* BIT EQUALITY of everything before the first reduction -- level images, gradients, coefficient planes, right-hand sides -- in all
  three GNC steps, 1 / 2 / 3 channels, small frames and one of 4.4 Mpixel: a hard assertion, no tolerance.
* The flow against the primary oracle.  The zero background makes the linear systems ill-conditioned (outside the disc there is no
  data term: a weighted Laplacian that 30 PCG iterations do not converge), and single roundings are amplified ~100 x more than on
  the lattice scenes: the oracle's OWN valid variants (FMA-contracted build, 8 x finer launch geometry) are 1e-4 ... 4e-4 apart on
  multi-level solves of this family (oracle-only, reproducible on the CPU: tools/disc_parity.py, profiles/r5_disc_parity.txt).
  The bar is therefore set by the oracle itself, case by case: where its variants agree to 1e-5 the suite's 2e-5 is asserted (all
  large single-level solves, BASELINE-sized pyramids: configs[3] at quarter scale 1.05e-5, configs[1]'s shape 3.6e-6); where they do
  not, the HIP path has to be no further from the primary oracle than 3 x the furthest oracle variant.  Equal iteration counts
  always.  profiles/r5_disc_parity.txt also shows the distance growing linearisation by linearisation -- for the HIP path and for the
  oracle's FMA build alike, at the same linearisation (one solve of the finest level's first GNC step lifts both from 6e-5 to 2e-3)."""
import numpy as np
import pytest

from conftest import rel_l2
from octane_amd import synth

pytestmark = pytest.mark.gpu

BAR = 1e-4
INVESTIGATE = 2e-5
COEF = tuple(zip(range(7), (0, 1, 2, 5, 6, 7, 8), ("a1", "a2", "a4", "a7", "a8", "bu", "bv")))


def _guess(nx, ny, m, scale=0.8):
    """A first guess that is not zero next to the limb and exactly zero in space (as a -firstguess file's is)."""
    tu, tv = synth.true_lattice_flow(nx, ny)
    return (scale * tu * m).astype(np.float32), (scale * tv * m).astype(np.float32)


@pytest.mark.parametrize("nc", [1, 2, 3])
def test_disc_pyramid_gradients_and_first_assembly_are_bit_exact(capi, oracle, nc):
    """Level images, gradients and the first linearisation of the coarsest level on the disc scene: bit for bit the oracle's."""
    nx, ny = 296, 248
    a, b = synth.disc_scene(nx, ny, seed=31 + nc, nchan=nc)
    assert (a == 0).mean() > 0.15 and (np.diff(a[0], axis=1) == 0).mean() > 0.15       # space and plateaus are really there
    prm = dict(kiters=3, liters=1, cgiters=2)
    tr_o, tr_g = {}, {}
    oracle.flow(a, b, oracle.FlowParams(**prm), trace=tr_o)
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    pl.set_trace(tr_g)
    pl.run_host(a, b)
    pl.close()
    for k in range(3):
        for tag in ("img1", "img2", "gx1", "gy1", "gx2", "gy2", "gxx", "gxy", "gyy"):
            assert np.array_equal(tr_g[(tag, k, -1, -1)], tr_o[(tag, k, -1, -1)]), (tag, k)
    g, o = tr_g[("coef7", 0, 0, 0)], tr_o[("coef", 0, 0, 0)]
    for gi, oi, nm in COEF:
        assert np.array_equal(g[gi], o[oi]), nm


@pytest.mark.parametrize("nc,dozim,lambdac", [(1, 1, 0.0), (1, 1, 0.3), (1, 0, 0.0), (2, 1, 0.0), (3, 1, 0.0), (3, 0, 0.4)])
def test_disc_assembly_of_every_gnc_step_is_bit_exact(capi, oracle, nc, dozim, lambdac):
    """cgiters = 0: the three assemblies of the level (al1 = 1, 0.5, 0) see the first guess -- non-zero next to the limb, so the warp
    crosses the taper and reads exact zeros beyond it -- and must equal the oracle's planes bit for bit, also where every gradient is
    exactly 0 (space, plateaus, the saturated patch).  All template instances of k_assemble: 1 - 3 channels, Zimmer / Brox, with and
    without the hint term."""
    nx, ny = 310, 270
    a, b = synth.disc_scene(nx, ny, seed=77 + nc, nchan=nc, centre=(0.35, 0.6), span=0.9)
    m = synth.disc_mask(nx, ny, (0.35, 0.6), 0.9)
    u0, v0 = _guess(nx, ny, m)
    prm = dict(kiters=1, liters=1, cgiters=0, dozim=dozim, lambdac=lambdac)
    tr_o, tr_g = {}, {}
    oracle.flow(a, b, oracle.FlowParams(**prm), u0=u0, v0=v0, trace=tr_o)
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    pl.set_trace(tr_g)
    pl.run_host(a, b, u0, v0)
    pl.close()
    for gnc in range(3):
        g, o = tr_g[("coef7", 0, gnc, 0)], tr_o[("coef", 0, gnc, 0)]
        for gi, oi, nm in COEF:
            assert np.array_equal(g[gi], o[oi]), (nc, gnc, nm, int((g[gi] != o[oi]).sum()))
        assert np.isfinite(g).all()
    space = m == 0
    assert not tr_g[("coef7", 0, 2, 0)][5][space & np.roll(space, 3, 1) & np.roll(space, -3, 1) & np.roll(space, 3, 0) & np.roll(space, -3, 0)].any()   # rhs is exactly 0 deep in space


def test_disc_assembly_is_bit_exact_on_a_large_level(capi, oracle):
    """4.4 Mpixel: the level runs k_assemble's border-free interior tiles, and the disc edge, the taper and the saturated patch cross
    them; first guess non-zero next to the limb.  All three GNC steps, bit for bit."""
    nx, ny = 2300, 1900
    a, b = synth.disc_scene(nx, ny, seed=91, centre=(0.45, 0.55), span=0.95)
    m = synth.disc_mask(nx, ny, (0.45, 0.55), 0.95)
    u0, v0 = _guess(nx, ny, m)
    prm = dict(kiters=1, liters=1, cgiters=0)
    tr_o, tr_g = {}, {}
    oracle.flow(a, b, oracle.FlowParams(**prm), u0=u0, v0=v0, trace=tr_o, flavour="omp")
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
    pl.set_trace(tr_g)
    pl.run_host(a, b, u0, v0)
    pl.close()
    for gnc in range(3):
        g, o = tr_g[("coef7", 0, gnc, 0)], tr_o[("coef", 0, gnc, 0)]
        for gi, oi, nm in COEF:
            assert np.array_equal(g[gi], o[oi]), (gnc, nm, int((g[gi] != o[oi]).sum()))


def _flows(capi, oracle, a, b, prm, u0=None, v0=None):
    g = oracle.REF_GRID_THREADS
    P = oracle.FlowParams(**prm)
    oracle.set_threads(oracle.host_cpu_share())
    uo, vo, io = oracle.flow(a, b, P, u0=u0, v0=v0, flavour="omp", dot_threads=g)
    var = {"fma": oracle.flow(a, b, P, u0=u0, v0=v0, flavour="fma", dot_threads=g)[:2],
           "grid_x8": oracle.flow(a, b, P, u0=u0, v0=v0, flavour="omp", dot_threads=8 * g)[:2]}
    nc, ny, nx = a.shape
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    try:
        ug, vg = pl.run_host(a, b, u0, v0)
        ig = pl.last_iterations()
    finally:
        pl.close()
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    return (ug, vg, ig), (uo, vo, io), var


def _judge(case, got, prim, var, mask):
    """Equal iteration counts always.  The bar is set by the oracle itself, per case: where its own valid variants agree to within half of
    the suite's 2e-5 the HIP path has to be within 2e-5 of the primary oracle (TIGHT); where they do not -- the zero background's
    ill-conditioning at work -- it has to be no further from the primary than 3 x the furthest variant (and never beyond 2e-3)."""
    ug, vg, ig = got
    uo, vo, io = prim
    d = rel_l2(ug, vg, uo, vo)
    spread = {k: rel_l2(x, y, uo, vo) for k, (x, y) in var.items()}
    floor = max(spread.values())
    inside = mask == 1
    di = rel_l2(ug[inside], vg[inside], uo[inside], vo[inside])
    tight = floor <= INVESTIGATE / 2
    print(f"PARITY-DISC case={case}: d_primary={d:.3e} (inside the disc {di:.3e}) oracle_spread={ {k: f'{x:.2e}' for k, x in spread.items()} } "
          f"iterations oracle/gpu={io}/{ig} {'TIGHT (2e-5)' if tight else 'ILL-CONDITIONED: held to 3 x the oracle spread'}")
    assert ig == io
    if tight:
        assert d < INVESTIGATE, f"{case}: {d:.3e} from the primary oracle (oracle spread {floor:.2e})"
    else:
        assert d < 3 * floor, f"{case}: {d:.3e} from the primary oracle, the oracle's own variants are within {floor:.2e}"
        assert d < 20 * BAR
    return d, floor, tight


@pytest.mark.parametrize("nx,ny,nc,prm,kw,guess,expect_tight", [
    (300, 280, 1, dict(kiters=1), {}, False, True),
    (300, 280, 1, dict(kiters=1, liters=1, cgiters=10), {}, True, True),
    (260, 300, 3, dict(kiters=1, liters=1, cgiters=8), {}, False, True),
    (2300, 1900, 1, dict(kiters=1, liters=1, cgiters=7), {}, False, None),                             # the q-recomputing LDS-DMA kernel (spread 1.1e-5: at the line)
    (2300, 1900, 1, dict(kiters=1, liters=1, cgiters=7), dict(centre=(0.2, 0.3), span=0.7), True, True),
    (2090, 1730, 2, dict(kiters=2, liters=1, cgiters=6), {}, True, True),
    # two linearisations from a first guess next to the limb: the oracle's variants are 1.5e-3 apart on ONE level (measured, round 5)
    (310, 270, 2, dict(kiters=1, liters=2, cgiters=12), dict(centre=(0.35, 0.6), span=0.9), True, False),
    (300, 280, 1, dict(kiters=4), {}, False, False),
    (300, 280, 1, dict(kiters=4), dict(noise=0.0), False, None),              # no noise: the plateaus stay plateaus
    (320, 300, 2, dict(kiters=3, liters=2, cgiters=12), {}, True, None),
    (260, 300, 3, dict(kiters=3, liters=1, cgiters=8), {}, False, None),
    (400, 360, 1, dict(kiters=4), dict(centre=(0.1, 0.2), span=0.6), False, None),   # the limb through a corner of the frame
])
def test_disc_solves_match_the_oracle(capi, oracle, nx, ny, nc, prm, kw, guess, expect_tight):
    """Whole solves on the disc scene, single- and multi-level, 1 - 3 channels, with and without a first guess.  `expect_tight` pins the
    regime that was measured when the case was added (None: the oracle's spread sits near the line): a case that was TIGHT must not
    drift into the lenient regime unnoticed."""
    a, b = synth.disc_scene(nx, ny, seed=nx * 3 + ny, nchan=nc, **kw)
    m = synth.disc_mask(nx, ny, kw.get("centre", (0.5, 0.5)), kw.get("span", 1.0))
    u0, v0 = _guess(nx, ny, m) if guess else (None, None)
    got, prim, var = _flows(capi, oracle, a, b, prm, u0, v0)
    d, floor, tight = _judge(f"{nx}x{ny}x{nc}_{'_'.join(f'{k}{v}' for k, v in prm.items())}", got, prim, var, m)
    if expect_tight is not None:
        assert tight == expect_tight, f"the oracle's spread on this case moved: {floor:.2e}"
