"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's sizes --
through size-independent properties.  Bar (north_star): u/v within 1e-4 relative L2; measured
values are ~1e-6 (reduction order is the only difference), so anything above 2e-5 fails here as
'investigate'."""
import numpy as np
import pytest

from conftest import rel_l2
from octane_amd import synth

pytestmark = pytest.mark.gpu

BAR = 1e-4          # north_star tolerance: asserted for EVERY case that is not on the allow-list below
INVESTIGATE = 2e-5  # expect ~3e-6; everything not allow-listed has to stay below this

# Cases whose distance to the primary oracle may exceed INVESTIGATE, each with the reason and the numbers measured when
# the entry was made.  An entry sets the case's own bar (never "whatever the floor happens to be today"), and _check
# verifies that the entry is still justified: the oracle's own spread on that case (FMA-contracted vs strict build of the
# same source, and the reference's other valid dot-product schedules) has to exceed INVESTIGATE / 2, otherwise the entry
# is stale and the test fails.  Nothing else may be further than INVESTIGATE (hence BAR) from the oracle.
ALLOW = {
    # alpha = 12, lambda = 0.25 on a 90 x 70 frame: the truncated solve amplifies single roundings -- the oracle's strict
    # and FMA builds are 1.8e-4 apart, the GPU 1.1e-4 from the strict one
    "lat_90x70_a12_l025": dict(bar=4e-4, why="oracle strict vs FMA build 1.8e-4 apart"),
}


def _check(capi, oracle, a, b, prm_kwargs, u0=None, v0=None, case=None):
    """GPU vs the strict oracle under the reference's launch geometry (the PRIMARY oracle: what a GPU run of the
    reference adds up, oracle/vof_oracle.c dotf).  Asserts d < INVESTIGATE (2e-5 < BAR = 1e-4) unless `case` is on the
    ALLOW list, whose entry then gives the bar.  The oracle's other valid variants (FMA-contracted build, one-thread
    running sums on small frames, 8x finer launch geometry on large ones) are computed and PRINTED as information: they
    measure how sensitive the problem itself is to rounding, they do not move the bar.  Every number goes to stdout so
    that `pytest -rP` logs carry them."""
    g = oracle.REF_GRID_THREADS
    P = oracle.FlowParams(**prm_kwargs)
    uo, vo, its_o = oracle.flow(a, b, P, u0=u0, v0=v0, dot_threads=g)
    uf, vf, _ = oracle.flow(a, b, P, u0=u0, v0=v0, flavour="fma", dot_threads=g)
    nc, ny, nx = (1,) + a.shape if a.ndim == 2 else a.shape
    spread = {"fma": rel_l2(uf, vf, uo, vo)}
    others = {}
    if nx * ny <= 100_000:       # the reference's one-thread schedule (what the survey recorded)
        us, vs, _ = oracle.flow(a, b, P, u0=u0, v0=v0)
        spread["serial"] = rel_l2(us, vs, uo, vo); others["serial"] = (us, vs)
    if nx * ny > 3_000_000:      # a finer launch geometry (1.7e-4 apart on a 4.4 Mpixel solve truncated at cgiters = 3)
        u8, v8, _ = oracle.flow(a, b, P, u0=u0, v0=v0, flavour="omp", dot_threads=8 * g)
        spread["grid_x8"] = rel_l2(u8, v8, uo, vo); others["grid_x8"] = (u8, v8)
    floor = max(spread.values())
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm_kwargs))
    ug, vg = pl.run_host(a, b, u0, v0)
    its_g = pl.last_iterations()
    pl.close()
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    d = rel_l2(ug, vg, uo, vo)
    info = {k: rel_l2(ug, vg, *o) for k, o in others.items()}
    allow = ALLOW.get(case)
    bar = allow["bar"] if allow else INVESTIGATE
    print(f"PARITY case={case or '-'} {nx}x{ny}x{nc} {prm_kwargs}: d_primary={d:.3e} bar={bar:.1e} "
          f"(north-star {BAR:.0e}{', ALLOW-LISTED: ' + allow['why'] if allow else ''}) "
          f"oracle_spread={ {k: f'{x:.2e}' for k, x in spread.items()} } "
          f"d_other={ {k: f'{x:.2e}' for k, x in info.items()} } iterations oracle/gpu={its_o}/{its_g}")
    if allow:
        assert floor > INVESTIGATE / 2, f"allow-list entry {case!r} is stale: the oracle's own spread is only {floor:.2e}"
    else:
        assert bar <= BAR
    assert d < bar, f"relative L2 {d:.3e} vs the primary oracle (bar {bar:.1e}; oracle spread {floor:.2e})"
    return d, its_o, its_g


@pytest.mark.parametrize("n,shift", [(64, (1.5, -0.75)), (128, (2.0, 1.0))])
def test_s1_scene_matches_oracle(capi, oracle, n, shift):
    a, b = synth.gaussian_scene(n, shift)
    d, io, ig = _check(capi, oracle, a, b, {}, case=f"s1_{n}")
    assert io == ig


def test_config0_s1_512_matches_oracle(capi, oracle):
    """BASELINE.json configs[0]: the 512 x 512 translating-Gaussian pair (S1 of SURVEY 8d: shift (3, -2), alpha = 5,
    lambda = 1, the command line's defaults kiters 4, liters 3, cgiters 30 -> 1080 PCG iterations), in-out u / v with a
    zero first guess as ref .cu:1213 takes them.  ~6 s of oracle time per build."""
    a, b = synth.gaussian_scene(512, (3.0, -2.0))
    z = np.zeros((512, 512), np.float32)
    d, io, ig = _check(capi, oracle, a, b, dict(alpha=5.0, lambda_=1.0), u0=z, v0=z.copy(), case="config0_s1_512")
    assert io == ig == 4 * 3 * 3 * 30
    ug, vg = capi.flow(a, b, capi.FlowParams(alpha=5.0, lambda_=1.0))
    assert abs(synth.interior_mean(ug) - 3.0004) < 2e-3 and abs(synth.interior_mean(vg) + 2.0010) < 2e-3


@pytest.mark.parametrize("nx,ny,nc,prm", [
    (96, 80, 1, dict(kiters=3)),
    (200, 150, 1, dict(kiters=4)),                       # ragged: not a multiple of any tile
    (131, 67, 1, dict(kiters=3, liters=2)),              # odd sizes, odd level sizes
    (75, 53, 2, dict(kiters=2, liters=2, cgiters=12)),   # two channels (channel-0 decimation quirk)
    (64, 48, 3, dict(kiters=3, liters=1, cgiters=8)),    # three channels
    (90, 70, 1, dict(kiters=2, dozim=0)),                # -brox
    (90, 70, 1, dict(kiters=3, alpha=12.0, lambda_=0.25)),
    (257, 129, 1, dict(kiters=1, liters=2)),             # single level: no pyramid at all
    (300, 260, 1, dict(kiters=5, liters=1, cgiters=40)),
])
def test_lattice_scene_matches_oracle(capi, oracle, nx, ny, nc, prm):
    a, b = synth.lattice_scene(nx, ny, seed=nx * 7 + ny, nchan=nc)
    case = "lat_90x70_a12_l025" if prm.get("alpha") == 12.0 else f"lat_{nx}x{ny}x{nc}"
    d, io, ig = _check(capi, oracle, a, b, prm, case=case)
    assert io == ig


def test_multi_tile_persistent_loops_match_oracle(capi, oracle):
    """Large enough that every persistent kernel walks several tiles per workgroup (more than
    1024 tiles): the regime BASELINE's sizes run in.  Caught an in-place halo race once."""
    nx, ny = 1300, 1040
    a, b = synth.lattice_scene(nx, ny, seed=77)
    _check(capi, oracle, a, b, dict(kiters=2, liters=1, cgiters=15), case="multi_tile_1300x1040")


def test_one_thread_schedule_of_the_oracle_also_agrees_on_small_frames(capi, oracle):
    """The survey's recorded answers come from the one-thread schedule; on a small frame its
    summation error is still below the bar, so the GPU agrees with it too."""
    a, b = synth.gaussian_scene(64, (1.5, -0.75))
    uo, vo, _ = oracle.flow(a, b)                      # dot_threads=0
    ug, vg = capi.flow(a, b)
    assert rel_l2(ug, vg, uo, vo) < INVESTIGATE


def test_level_above_four_megapixels_matches_oracle(capi, oracle):
    """Levels of 2 * 2^20 pixels and more run the q-recomputing form of the fused PCG kernel (k_pcg_fused_q: tile + ring,
    p staged two pixels out, x updated every second launch) -- and, with OCTANE_TUNE_FUSED=0, the LDS-ring marching
    form of pass A.  2300 x 1900 has ragged last tiles in both directions; seven iterations reach every branch of the
    deferred x update (first launch, odd, even without and with a stored x)."""
    nx, ny = 2300, 1900
    a, b = synth.lattice_scene(nx, ny, seed=55)
    _check(capi, oracle, a, b, dict(kiters=1, liters=1, cgiters=7), case="q_2300x1900")


def test_large_level_with_two_channels_first_guess_and_hint_term(capi, oracle):
    """The q-recomputing kernel under everything the assembly can feed it: two channels, a first guess that is not zero
    and the hint term (lambdac), on a level of 3.6 Mpixel with ragged tiles."""
    nx, ny = 2090, 1730
    a, b = synth.lattice_scene(nx, ny, seed=57, nchan=2)
    tu, tv = synth.true_lattice_flow(nx, ny)
    u0 = (0.7 * tu).astype(np.float32); v0 = (0.7 * tv).astype(np.float32)
    _check(capi, oracle, a, b, dict(kiters=1, liters=1, cgiters=6, lambdac=0.3), u0=u0, v0=v0, case="q_2090x1730_nc2_hint")


def test_first_guess_and_hint_term(capi, oracle):
    """lambdac != 0 (only reachable with -firstguess): the hint term and its pyramid."""
    nx, ny = 120, 88
    a, b = synth.lattice_scene(nx, ny, seed=21)
    tu, tv = synth.true_lattice_flow(nx, ny)
    u0 = (tu + 0.3).astype(np.float32)
    v0 = (tv - 0.2).astype(np.float32)
    _check(capi, oracle, a, b, dict(kiters=3, lambdac=0.5), u0, v0, case="hint_120x88")


def test_early_exit_of_the_pcg_loop(capi, oracle, golden_flow):
    """A case where the reference's tolerance test ends solves early (its < cap): the device-side
    stop logic must stop at the same iteration."""
    name = "lat_60x44_brox_hint"
    a, b = golden_flow[name + "_img1"], golden_flow[name + "_img2"]
    u0 = np.full(a.shape[1:], 2.0, np.float32)
    v0 = np.full(a.shape[1:], -1.0, np.float32)
    prm = dict(kiters=2, dozim=0, lambdac=0.5, alpha=8.0, lambda_=0.5)
    d, io, ig = _check(capi, oracle, a, b, prm, u0, v0, case="early_exit_60x44")
    assert io == int(golden_flow[name + "_its"]) and io < 2 * 3 * 3 * 30
    assert ig == io


def _recurrence_rows(store, tol):
    """From the 'pcg_sums' rows of the debug tap (one per launch: the launch's own direct sums rz rr pq qz qmq rq qq over the
    state it formed, then the PcgState it left: rz used, stopped, iterations): the NEXT launch's r.z / r.r by the kernels'
    one-step recurrence (pcg_fused_q_dma.hip:254-261, pcg_kernels.hip:715-722, formed here exactly as there: alpha in float
    from floats, the quadratic in double) next to the direct sums the next launch formed over the residual it wrote."""
    out = []
    for key in sorted(k for k in store if k[0] == "pcg_sums"):
        rows = store[key].reshape(-1).view(np.float64).reshape(-1, 10)
        for k in range(len(rows) - 1):
            rz, rr, pq, qz, qmq, rq, qq, rz_used, stopped, _ = rows[k]
            if stopped or rows[k + 1][8]:
                break                                   # the launch after a stop writes nothing
            alpha = np.float32(rz_used) / np.float32(pq)              # ref .cu:1169
            a = float(alpha)
            rz_pred = rz - 2. * a * qz + a * a * qmq
            rr_pred = rr - 2. * a * rq + a * a * qq
            assert np.float32(rz_pred) == np.float32(rows[k + 1][7])   # the next launch stored exactly this as the r.z it used
            out.append((key[1:], k + 1, rr_pred, rows[k + 1][1], rz_pred, rows[k + 1][0]))
    return out


@pytest.mark.parametrize("case", ["early_exit_60x44", "q_dma_2304x1100"])
def test_one_step_recurrence_of_rz_and_rr_agrees_with_the_direct_sums(capi_diag, golden_flow, case):
    """The SECOND arithmetic freedom the HIP path takes (the first is the summation order): r.z and r.r of the residual a
    launch is about to form -- hence beta and the stop test `residc > tol` (ref .cu:1131) -- come from
    (r.z)_k = (r.z)_{k-1} - 2 alpha (q.z)_{k-1} + alpha^2 (q.M^-1 q)_{k-1}, likewise r.r, where the reference sums over the
    residual itself (ref .cu:1135-1178).  Base values are DIRECT sums of the previous launch, nothing is chained.  Logged per
    iteration (pytest -rP) and asserted: predicted and direct agree to 1e-6 relative wherever r.r > 10 tol -- on the tolerance-exit
    case of test_early_exit_of_the_pcg_loop (per-launch stored-q kernel: the persistent and single-workgroup solves switched
    off) and on a level the LDS-DMA kernel runs (>= 2 Mi pixels)."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    tol = 1e-8                                           # ref .cu:1240
    if case == "early_exit_60x44":
        name = "lat_60x44_brox_hint"
        a, b = golden_flow[name + "_img1"], golden_flow[name + "_img2"]
        u0 = np.full(a.shape[1:], 2.0, np.float32); v0 = np.full(a.shape[1:], -1.0, np.float32)
        prm = capi.FlowParams(kiters=2, dozim=0, lambdac=0.5, alpha=8.0, lambda_=0.5)
        want_its = int(golden_flow[name + "_its"])
    else:
        a, b = synth.lattice_scene(2304, 1100, seed=5)
        u0 = v0 = None
        prm = capi.FlowParams(kiters=1, liters=1, cgiters=30)
        want_its = 90
    ny, nx = a.shape[-2:]
    pl = capi.Plan(nx, ny, 1, prm)
    try:
        pl.tune("persist", 0); pl.tune("small", 0)       # one launch per iteration on every level
        store = {}
        pl.set_trace(store)
        pl.run_host(a, b, u0, v0)
        assert pl.last_iterations() == want_its           # the trace changes nothing: the same exits as the oracle's
        pl.set_trace(None)
    finally:
        pl.close()
    rows = _recurrence_rows(store, tol)
    assert len(rows) >= want_its // 2
    worst = 0.0
    for key, k, rr_p, rr_d, rz_p, rz_d in rows:
        e_rr = abs(rr_p - rr_d) / abs(rr_d) if rr_d else 0.0
        e_rz = abs(rz_p - rz_d) / abs(rz_d) if rz_d else 0.0
        print(f"RECURRENCE {case} level/gnc/l={key} launch {k:2d}: r.r predicted {rr_p:.9e} direct {rr_d:.9e} (rel {e_rr:.1e})   "
              f"r.z predicted {rz_p:.9e} direct {rz_d:.9e} (rel {e_rz:.1e})")
        if rr_d > 10 * tol:
            worst = max(worst, e_rr, e_rz)
            assert e_rr <= 1e-6 and e_rz <= 1e-6, (key, k, rr_p, rr_d, rz_p, rz_d)
        # the stop decision itself: the two values are on the same side of tol except within rounding of it
        assert (rr_p > tol) == (rr_d > tol) or abs(rr_p - rr_d) <= 1e-6 * tol + 1e-6 * abs(rr_d)
    print(f"RECURRENCE {case}: {len(rows)} launches compared, worst relative difference above 10 tol: {worst:.2e}")


def test_identical_images_give_zero_flow_without_nans(capi, oracle):
    """b == 0 -> residual 0 -> the loop never runs (ref .cu:1131); 0/0 must not appear."""
    a, _ = synth.lattice_scene(80, 64, seed=3)
    uo, vo, its = oracle.flow(a, a, oracle.FlowParams(kiters=2))
    pl = capi.Plan(80, 64, 1, capi.FlowParams(kiters=2))
    ug, vg = pl.run_host(a, a)
    assert pl.last_iterations() == its
    assert np.isfinite(ug).all() and np.abs(ug).max() < 1e-5 and np.abs(vg).max() < 1e-5
    assert np.abs(ug - uo).max() < 1e-6 and np.abs(vg - vo).max() < 1e-6


def test_constant_images(capi):
    a = np.full((1, 40, 56), 17.0, np.float32)
    u, v = capi.flow(a, a, capi.FlowParams(kiters=2))
    assert not u.any() and not v.any()


@pytest.mark.parametrize("name,prm,guess", [
    ("s1_64", dict(), None),
    ("lat_96x80_k3", dict(kiters=3), None),
    ("lat_75x53_k2_nc2", dict(kiters=2, liters=2, cgiters=12), None),
    ("lat_60x44_brox_hint", dict(kiters=2, dozim=0, lambdac=0.5, alpha=8.0, lambda_=0.5), (2.0, -1.0)),
])
def test_committed_golden_fixtures(capi, golden_flow, name, prm, guess):
    a, b = golden_flow[name + "_img1"], golden_flow[name + "_img2"]
    u0 = v0 = None
    if guess:
        u0 = np.full(a.shape[1:], guess[0], np.float32)
        v0 = np.full(a.shape[1:], guess[1], np.float32)
    u, v = capi.flow(a, b, capi.FlowParams(**prm), u0, v0)
    assert rel_l2(u, v, golden_flow[name + "_u"], golden_flow[name + "_v"]) < INVESTIGATE


def test_first_assembly_is_bit_exact(capi, oracle):
    """Everything before the first reduction reproduces the strict oracle bit for bit: pyramid,
    gradients, warp, coefficients, rhs (the library is built with -ffp-contract=off)."""
    nx, ny = 112, 72
    a, b = synth.lattice_scene(nx, ny, seed=9)
    tr_o, tr_g = {}, {}
    oracle.flow(a, b, oracle.FlowParams(kiters=3, liters=1, cgiters=2), trace=tr_o)
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(kiters=3, liters=1, cgiters=2))
    pl.set_trace(tr_g)
    pl.run_host(a, b)
    for tag in ("img1", "img2", "gx1", "gy1", "gx2", "gy2", "gxx", "gxy", "gyy", "u0", "v0"):
        assert np.array_equal(tr_g[(tag, 0, -1, -1)], tr_o[(tag, 0, -1, -1)]), tag
    for k in range(3):      # images and gradients of every level (flow-independent)
        for tag in ("img1", "img2", "gx1", "gy1", "gx2", "gy2", "gxx", "gxy", "gyy"):
            assert np.array_equal(tr_g[(tag, k, -1, -1)], tr_o[(tag, k, -1, -1)]), (tag, k)
    g = tr_g[("coef7", 0, 0, 0)]
    o = tr_o[("coef", 0, 0, 0)]
    for gi, oi, nm in zip(range(7), (0, 1, 2, 5, 6, 7, 8), ("a1", "a2", "a4", "a7", "a8", "bu", "bv")):
        assert np.array_equal(g[gi], o[oi]), nm


def test_assembly_of_every_gnc_step_is_bit_exact(capi, oracle):
    """With cgiters = 0 no solve changes the flow, so the three assemblies of a level (al1 = 1, 0.5, 0: quadratic terms
    only, the blend, robust terms only -- three code paths in k_assemble) all see the first guess and can be compared
    with the oracle bit for bit.  A non-zero first guess and a hint term make the warp and every term non-trivial."""
    nx, ny = 150, 97
    a, b = synth.lattice_scene(nx, ny, seed=19)
    tu, tv = synth.true_lattice_flow(nx, ny)
    u0 = (0.8 * tu).astype(np.float32); v0 = (0.8 * tv).astype(np.float32)
    prm = dict(kiters=1, liters=1, cgiters=0, lambdac=0.3)
    tr_o, tr_g = {}, {}
    oracle.flow(a, b, oracle.FlowParams(**prm), u0=u0, v0=v0, trace=tr_o)
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
    pl.set_trace(tr_g)
    pl.run_host(a, b, u0, v0)
    pl.close()
    for gnc in range(3):
        g = tr_g[("coef7", 0, gnc, 0)]
        o = tr_o[("coef", 0, gnc, 0)]
        for gi, oi, nm in zip(range(7), (0, 1, 2, 5, 6, 7, 8), ("a1", "a2", "a4", "a7", "a8", "bu", "bv")):
            assert np.array_equal(g[gi], o[oi]), (gnc, nm, int((g[gi] != o[oi]).sum()))
        assert np.abs(g[5]).max() > 0 and np.isfinite(g).all()


@pytest.mark.parametrize("nc", [2, 3])
def test_assembly_with_two_and_three_channels_is_bit_exact_in_every_gnc_step(capi, oracle, nc):
    """Round 4: two and three channels have template instances of k_assemble of their own for the default flags
    (Zimmer's normalisation on, no hint term; ref .cu:749-829 loops over 1 ... 3 channels alike).  With cgiters = 0 the three assemblies
    of the level -- al1 = 1, 0.5, 0 -- see the first guess: coefficient planes and right-hand sides bit for bit the oracle's."""
    nx, ny = 150, 97
    a, b = synth.lattice_scene(nx, ny, seed=23, nchan=nc)
    tu, tv = synth.true_lattice_flow(nx, ny)
    u0 = (0.8 * tu).astype(np.float32); v0 = (0.8 * tv).astype(np.float32)
    prm = dict(kiters=1, liters=1, cgiters=0)
    tr_o, tr_g = {}, {}
    oracle.flow(a, b, oracle.FlowParams(**prm), u0=u0, v0=v0, trace=tr_o)
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    pl.set_trace(tr_g)
    pl.run_host(a, b, u0, v0)
    pl.close()
    for gnc in range(3):
        g = tr_g[("coef7", 0, gnc, 0)]
        o = tr_o[("coef", 0, gnc, 0)]
        for gi, oi, nm in zip(range(7), (0, 1, 2, 5, 6, 7, 8), ("a1", "a2", "a4", "a7", "a8", "bu", "bv")):
            assert np.array_equal(g[gi], o[oi]), (nc, gnc, nm, int((g[gi] != o[oi]).sum()))
        assert np.abs(g[5]).max() > 0 and np.isfinite(g).all()


def test_runs_are_bitwise_reproducible(capi):
    """Two-stage fixed-order reductions: no float atomics, so repeated runs agree exactly
    (the CUDA reference does not: SURVEY.md 2.2, jVecXVec)."""
    a, b = synth.lattice_scene(333, 217, seed=4)
    pl = capi.Plan(333, 217, 1, capi.FlowParams(kiters=4))
    u1, v1 = pl.run_host(a, b)
    u2, v2 = pl.run_host(a, b)
    assert np.array_equal(u1, u2) and np.array_equal(v1, v2)
    u3, v3 = capi.flow(a, b, capi.FlowParams(kiters=4))
    assert np.array_equal(u1, u3) and np.array_equal(v1, v3)


@pytest.mark.parametrize("nx,ny,nc,prm,guess", [
    (333, 217, 1, dict(kiters=4), False),
    (300, 260, 3, dict(kiters=3, lambdac=0.5), True),      # the hint term: every level has its own decimated first guess
    (2200, 1100, 1, dict(kiters=3, liters=1, cgiters=5), False),
])
def test_level_setup_on_the_side_stream_changes_no_bit(capi_diag, nx, ny, nc, prm, guess):
    """Round 3: the pyramid images, the first-guess hint and the gradient fields of level k + 1 are prepared on the plan's side stream,
    in a second set of planes, while level k is solved.  Placement in time only: the flow must have the bits of the one-stream run,
    also when the same plan runs repeatedly (the sets alternate by level parity; a run reuses what the previous one left)."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    a, b = synth.lattice_scene(nx, ny, seed=nx + 7 * ny, nchan=nc)
    rng = np.random.RandomState(3)
    u0 = (1.5 * rng.randn(ny, nx)).astype(np.float32) if guess else None
    v0 = (1.5 * rng.randn(ny, nx)).astype(np.float32) if guess else None
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    try:
        pl.tune("overlap", 0)
        ur, vr = pl.run_host(a, b, u0, v0)
        its = pl.last_iterations()
        pl.tune("overlap", 1)
        for rep in range(3):
            u, v = pl.run_host(a, b, u0, v0)
            assert pl.last_iterations() == its
            assert np.array_equal(u, ur) and np.array_equal(v, vr), f"run {rep} with the side stream differs from the one-stream run"
    finally:
        pl.close()
    assert np.isfinite(ur).all()


@pytest.mark.parametrize("cgiters", [1, 2, 7, 30])
def test_deferred_x_update_is_bitwise_identical(capi_diag, cgiters):
    """Pass B applies x += alpha p for two iterations at once (every second launch) in the reference's order of
    operations; the result must equal the every-iteration form bit for bit, for even and odd iteration counts
    (an odd count leaves one update pending for the flow-update kernel)."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    nx, ny = 310, 240           # finest level above the single-workgroup solver's size
    a, b = synth.lattice_scene(nx, ny, seed=17)
    prm = capi.FlowParams(kiters=2, liters=1, cgiters=cgiters)
    outs = []
    for mode in (0, 1):
        pl = capi.Plan(nx, ny, 1, prm)
        try:
            pl.tune("defer_x", mode)       # (until round 4 an environment variable; the product library no longer reads the tuning variables)
            outs.append(pl.run_host(a, b))
        finally:
            pl.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("variant", [2, 3])
def test_unit_weight_pass_a_is_bitwise_identical(capi_diag, variant):
    """In the first GNC step every neighbour weight is exactly -1 and pass A does not read the wx / wy planes;
    the result must equal the plane-reading form bit for bit (tiled and marching forms of pass A forced)."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    nx, ny = 1150, 700
    a, b = synth.lattice_scene(nx, ny, seed=23)
    prm = capi.FlowParams(kiters=2, liters=1, cgiters=9)
    outs = []
    for unit in (0, 1):
        pl = capi.Plan(nx, ny, 1, prm)
        try:
            pl.tune("pass_a", variant)
            pl.tune("unit_w", unit)
            outs.append(pl.run_host(a, b))
        finally:
            pl.tune("pass_a", 0)
            pl.close()
    assert np.isfinite(outs[0][0]).all()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_level_too_small_is_an_error(capi):
    a = np.zeros((1, 20, 20), np.float32)
    with pytest.raises(capi.OctaneError) as e:
        capi.flow(a, a, capi.FlowParams(kiters=6))
    assert e.value.code == capi.E_TOOSMALL


def test_batch_entry_equals_single_runs(capi):
    pairs = [synth.lattice_scene(100, 76, seed=s) for s in (1, 2, 3)]
    prm = capi.FlowParams(kiters=3, liters=2)
    outs = capi.batch_flow(pairs, prm, devices=[0])
    for (a, b), (u, v) in zip(pairs, outs):
        us, vs = capi.flow(a, b, prm)
        assert np.array_equal(u, us) and np.array_equal(v, vs)


def test_config4_batch_of_64_pairs_2000(capi):
    """BASELINE.json configs[4]: a batch of 64 independent 2000 x 2000 pairs (kiters 6, the command line's liters 3 and
    cgiters 30) through octane_vof_batch_run with its default lanes.  Every pair has to come back bit-equal to a
    single-plan run of the same pair with the fixed 6 * 3 * 3 * 30 = 1620 iterations: lanes share a GPU, never a result."""
    import torch
    n, npairs = 2000, 64
    prm = capi.FlowParams(kiters=6)
    pairs = []
    for s in range(npairs):
        a, b = synth.lattice_scene(n, n, seed=20240617 + s, device="cuda")
        pairs.append((a.cpu().numpy(), b.cpu().numpy()))
    torch.cuda.synchronize()
    outs = capi.batch_flow(pairs, prm, devices=[0])
    assert len(outs) == npairs
    # Round 5: the two lanes of a batch run their persistent mid-level solves CONCURRENTLY, each capped at half the compute units
    # (tune "lane_mode" 2, +9 % per batch); the cap changes the sub-domain grid of those levels and with it the grouping of their fp64
    # partial sums, so a lane's flow equals the DEFAULT plan's to the last bits of the PCG scalars (asserted below the suite's 2e-5),
    # and equals bit for bit a single plan configured the way the lanes are -- lanes share a GPU, never a result.
    pl = capi.Plan(n, n, 1, prm)
    pl.set_lane_mode(2)
    pd = capi.Plan(n, n, 1, prm)
    ud, vd = pd.run_host(*pairs[0])
    pd.close()
    d_default = rel_l2(outs[0][0], outs[0][1], ud, vd)
    print(f"PARITY case=config4_batch64 pair 0: lane configuration vs default plan relL2 {d_default:.3e}")
    assert d_default < INVESTIGATE
    nbad, worst = 0, 0.0
    for k, ((a, b), (u, v)) in enumerate(zip(pairs, outs)):
        us, vs = pl.run_host(a, b)
        assert pl.last_iterations() == 6 * 3 * 3 * 30, k
        assert np.isfinite(u).all() and np.isfinite(v).all(), k
        same = np.array_equal(u, us) and np.array_equal(v, vs)
        nbad += 0 if same else 1
        worst = max(worst, rel_l2(u, v, us, vs))
    pl.close()
    tu, tv = synth.true_lattice_flow(n, n)
    m = n // 8
    eu = np.abs(outs[-1][0].astype(np.float64) - tu)[m:-m, m:-m].mean(); ev = np.abs(outs[-1][1].astype(np.float64) - tv)[m:-m, m:-m].mean()
    print(f"PARITY case=config4_batch64 64 x {n}x{n}: pairs differing from their single-plan run {nbad}/64, worst relL2 {worst:.3e}; "
          f"mean |flow - truth| of the last pair {eu:.4f}, {ev:.4f} px")
    assert nbad == 0, f"{nbad} of 64 pairs differ from their single-plan runs (worst relative L2 {worst:.3e})"
    assert eu < 0.05 and ev < 0.05


def test_device_pointer_entry_equals_host_entry(capi):
    import torch
    nx, ny = 192, 140
    a, b = synth.lattice_scene(nx, ny, seed=8)
    prm = capi.FlowParams(kiters=3)
    pl = capi.Plan(nx, ny, 1, prm)
    uh, vh = pl.run_host(a, b)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    du = torch.zeros(ny, nx, device="cuda")
    dv = torch.zeros(ny, nx, device="cuda")
    torch.cuda.synchronize()
    pl.run_device(da.data_ptr(), db.data_ptr(), du.data_ptr(), dv.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(du.cpu().numpy(), uh) and np.array_equal(dv.cpu().numpy(), vh)


def test_full_size_properties_2000(capi):
    """BASELINE.json configs[1] shape (2000x2000, 6 levels): no oracle at this size in test time,
    so check what the domain offers: the recovered flow follows the analytic displacement field,
    the run is reproducible, and the iteration count is the fixed kiters*3*liters*cgiters."""
    import torch
    n = 2000
    a, b = synth.lattice_scene(n, n, seed=20240614, device="cuda")
    prm = capi.FlowParams(kiters=6)
    pl = capi.Plan(n, n, 1, prm)
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
    torch.cuda.synchronize()
    u1, v1 = u.clone(), v.clone()
    assert pl.last_iterations() == 6 * 3 * 3 * 30
    tu, tv = synth.true_lattice_flow(n, n, xp=torch)
    m = n // 8
    eu = (u1.cpu().double() - tu)[m:-m, m:-m].abs()
    ev = (v1.cpu().double() - tv)[m:-m, m:-m].abs()
    assert eu.mean() < 0.05 and ev.mean() < 0.05, (eu.mean(), ev.mean())
    u.zero_(); v.zero_()
    pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(u, u1) and torch.equal(v, v1)


def test_headline_config_5000_three_forms_of_the_pcg_agree(capi_diag):
    """BASELINE.json's headline run: 5000 x 5000, 8 levels, 2160 PCG iterations.  No oracle at this size
    in test time; instead the three independent forms of the PCG iteration this library has -- one kernel recomputing q
    (the default on the two finest levels), one kernel storing q, and the two-pass form with its own kernels -- are run
    on the same pair and have to agree, the recovered flow has to follow the analytic displacement field, and the
    iteration count is the fixed kiters * 3 * liters * cgiters."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    import torch
    n = 5000
    a, b = synth.lattice_scene(n, n, seed=20240615, device="cuda")
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8))
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for name, knobs in (("q", dict(fused=1, fused_q=1)), ("stored", dict(fused=1, fused_q=0)), ("two_pass", dict(fused=0, fused_q=1))):
            for key, val in knobs.items():
                pl.tune(key, val)
            u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
            torch.cuda.synchronize()
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
            torch.cuda.synchronize()
            assert pl.last_iterations() == 8 * 3 * 3 * 30, name
            assert bool(torch.isfinite(u).all()) and bool(torch.isfinite(v).all()), name
            res[name] = (u, v)
    finally:
        pl.tune("fused", 1); pl.tune("fused_q", 1)
        pl.close()

    def dist(x, y):
        num = ((x[0] - y[0]).double() ** 2).sum() + ((x[1] - y[1]).double() ** 2).sum()
        den = (y[0].double() ** 2).sum() + (y[1].double() ** 2).sum()
        return float(torch.sqrt(num / den))
    assert dist(res["q"], res["stored"]) < INVESTIGATE
    assert dist(res["q"], res["two_pass"]) < INVESTIGATE
    tu, tv = synth.true_lattice_flow(n, n, xp=torch)
    m = n // 8
    eu = (res["q"][0].cpu().double() - tu)[m:-m, m:-m].abs().mean()
    ev = (res["q"][1].cpu().double() - tv)[m:-m, m:-m].abs().mean()
    assert eu < 0.05 and ev < 0.05, (eu, ev)


def test_full_disk_frame_runs_on_one_gpu(capi):
    """BASELINE.json configs[3] shape: a 10848 x 10848 ABI full-disk pair.  The plan needs ~17 GB of the 288 GB, so
    the frame runs whole on ONE GPU (the reference's managed CSR would need 33 GB).  Properties only: the result is
    finite, follows the analytic displacement and the solver did its fixed amount of work."""
    import torch
    n = 10848
    a, b = synth.lattice_scene(n, n, seed=20240616, device="cuda")
    prm = capi.FlowParams(kiters=8, liters=1, cgiters=10)
    pl = capi.Plan(n, n, 1, prm)
    assert 10e9 < pl.device_bytes < 30e9
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    torch.cuda.synchronize()
    pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert pl.last_iterations() == 8 * 3 * 1 * 10
    assert bool(torch.isfinite(u).all()) and bool(torch.isfinite(v).all())
    m = n // 8
    cu = u[m:-m:16, m:-m:16].double().cpu(); cv = v[m:-m:16, m:-m:16].double().cpu()
    tu, tv = synth.true_lattice_flow(n, n, xp=torch)
    assert (cu - tu[m:-m:16, m:-m:16]).abs().mean() < 0.1 and (cv - tv[m:-m:16, m:-m:16]).abs().mean() < 0.1
    pl.close()


@pytest.mark.parametrize("nx,ny,prm", [(700, 520, dict(kiters=3, liters=2, cgiters=17)),      # 128 x 8 tiles, odd iteration count
                                        (1500, 1100, dict(kiters=2, liters=1, cgiters=12)),     # 128 x 16 tiles, more workgroups than are resident
                                        (300, 260, dict(kiters=4, liters=1, cgiters=1))])       # a single iteration: only the pending update
def test_fused_iteration_equals_two_pass_form(capi_diag, nx, ny, prm):
    """One fused kernel per PCG iteration (r.z, r.r of the next residual by recurrence) against pass A + pass B (direct
    sums): same iterates up to the rounding of those scalars, same iteration counts, and bit-identical from run to run
    (the partial sums are double-buffered: workgroups of one launch do not all run at the same time)."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    a, b = synth.lattice_scene(nx, ny, seed=nx + 3 * ny)
    outs, its = {}, {}
    for fused in (0, 1):
        pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
        pl.tune("fused", fused)
        outs[fused] = [pl.run_host(a, b) for _ in range(2)]
        its[fused] = pl.last_iterations()
        pl.close()
    assert its[0] == its[1]
    assert np.array_equal(outs[1][0][0], outs[1][1][0]) and np.array_equal(outs[1][0][1], outs[1][1][1])
    assert np.isfinite(outs[1][0][0]).all()
    d = rel_l2(outs[1][0][0], outs[1][0][1], outs[0][0][0], outs[0][0][1])
    assert d < INVESTIGATE, f"fused vs two-pass: {d:.3e}"


@pytest.mark.parametrize("nx,ny", [(2500, 1750), (2503, 1699)])
def test_q_recomputing_form_equals_stored_q_form(capi_diag, nx, ny):
    """k_pcg_fused_q (levels of at least 2 * 2^20 pixels; OCTANE_TUNE_FUSED_Q=0 / tune("fused_q", 0) turns it off) does not
    store q = A p but forms it again in the next launch from the stored p, on the tile and its one-pixel ring: same
    inputs, same operations as the form that stores q.  Frame with ragged right / bottom tiles and more tiles than
    workgroups; the second size has a width that is no multiple of 4 (a float4 group straddles the frame's edge)."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    a, b = synth.lattice_scene(nx, ny, seed=91)
    prm = capi.FlowParams(kiters=2, liters=1, cgiters=11)
    outs = []
    for q in (0, 1):
        pl = capi.Plan(nx, ny, 1, prm)
        try:
            pl.tune("fused_q", q)
            outs.append(pl.run_host(a, b))
        finally:
            pl.tune("fused_q", 1)
            pl.close()
    assert np.isfinite(outs[1][0]).all()
    d = rel_l2(outs[1][0], outs[1][1], outs[0][0], outs[0][1])
    nbad = int((outs[0][0] != outs[1][0]).sum())
    # element by element the two forms compute the same bits; their persistent grids differ, hence the grouping of the
    # fp64 partial sums, hence -- rarely -- the last bit of an alpha or beta
    assert d < 1e-6, f"{d:.3e}, {nbad} pixels differ"


@pytest.mark.parametrize("nx,ny,prm", [
    (2300, 1900, dict(kiters=1, liters=1, cgiters=7)),      # ragged tiles right and below: border tiles stage through registers
    (2503, 1699, dict(kiters=2, liters=1, cgiters=11)),     # a width that is no multiple of 4
    (2560, 2048, dict(kiters=1, liters=2, cgiters=6)),      # whole tiles only
])
def test_lds_dma_form_of_the_q_recomputing_kernel_is_bit_identical(capi_diag, nx, ny, prm):
    """Whole levels run the q-recomputing kernel with the next tile's p and ring operands fetched by LDS-DMA during phase 2, the
    border-free form of the operator on tiles strictly inside the frame and the reciprocal of the diagonal by rcp_exact
    (pcg_fused_q_dma.hip; the default).  Same arithmetic, same tile walk, same partial sums as the register-staged kernel
    (tune("q_dma", 0)): the flow has to be the same bits, in all three GNC steps (unit and varying weights).  (Widths whose
    tile-column count divides the 512-workgroup grid -- 2000, 2048 -- walk the tiles in another order in the DMA kernel, hence
    group the fp64 partial sums differently: the next test.)"""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    a, b = synth.lattice_scene(nx, ny, seed=nx - ny)
    outs = {}
    for dma in (0, 1):
        pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
        try:
            pl.tune("q_dma", dma)
            outs[dma] = pl.run_host(a, b)
            its = pl.last_iterations()
        finally:
            pl.tune("q_dma", 1)
            pl.close()
    ndiff = int((outs[0][0] != outs[1][0]).sum() + (outs[0][1] != outs[1][1]).sum())
    print(f"PARITY case=q_dma {nx}x{ny} {prm}: LDS-DMA vs register staging: {ndiff} values differ ({its} iterations)")
    assert np.isfinite(outs[1][0]).all() and ndiff == 0


def test_lds_dma_kernel_with_rotated_tile_columns_agrees_with_the_register_staged_kernel(capi_diag):
    """2048 pixels = 16 tile columns, which divides the grid of 512 workgroups: the DMA kernel rotates the columns of a tile row by the
    round number so that no workgroup owns the frame's border column in every round.  Another tile order = another grouping of
    the fp64 partial sums: not the same bits as the register-staged kernel, the same flow within 1e-5 and the same iteration count."""
    capi = capi_diag      # tuning knobs (and the two-pass form) exist in the diagnostic library only: both legs run on it
    nx, ny, prm = 2048, 1800, dict(kiters=1, liters=2, cgiters=8)
    a, b = synth.lattice_scene(nx, ny, seed=77)
    outs, its = {}, {}
    for dma in (0, 1):
        pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
        try:
            pl.tune("q_dma", dma)
            outs[dma] = pl.run_host(a, b)
            its[dma] = pl.last_iterations()
        finally:
            pl.tune("q_dma", 1)
            pl.close()
    d = rel_l2(outs[1][0], outs[1][1], outs[0][0], outs[0][1])
    print(f"PARITY case=q_dma_rotated {nx}x{ny} {prm}: rel-L2 {d:.3e}, iterations {its}")
    assert its[0] == its[1] and d < 1e-5


def test_tile_walks_of_the_lds_dma_kernel_on_random_large_shapes(capi_diag):
    """Round 6: the LDS-DMA PCG kernel walks its tiles backwards on odd launches, stores the tail of a long walk allocating, and rotates
    the tile columns of tile row r by r where the host's count says that spreads the border-column tiles (octane_vof_row_rotation) --
    three re-orderings of WHO does WHICH tile WHEN.  Every one of them has to be a permutation of the tiles (the arena is poisoned with
    NaNs: a tile done twice or not at all shows) and may change nothing but the grouping of the fp64 partial sums.  A seeded sweep over
    frames whose only level runs that kernel -- 2 to 13 rounds of tiles, 17 ... 44 tile columns, ragged right / bottom edges, odd and even launch
    counts, shapes on which the rotation is on and shapes on which it is off -- against the register-staged kernel (tune "q_dma" 0:
    always the forward, unrotated walk): same iteration count, flows within 1e-5 (measured: bit-identical in nearly every case)."""
    import ctypes as C
    capi = capi_diag
    L = capi.lib()
    rng = np.random.RandomState(20261005)
    # the BASELINE widths' column counts (40: 13 rounds of tiles, so the allocating tail is on too; 20), a ragged 40th column, a small frame
    shapes = [(5000, 2700), (2500, 2500), (4993, 1400), (3196, 705)]
    while len(shapes) < 14:
        nx, ny = int(rng.randint(2100, 6000)), int(rng.randint(400, 2800))
        if nx * ny >= (2 << 20) + 4096 and nx * ny < 14_000_000:
            shapes.append((nx, ny))
    rotated = 0
    for i, (nx, ny) in enumerate(shapes):
        out3 = (C.c_int * 3)()
        rot = L.octane_vof_row_rotation(nx, ny, 512, 4, out3)
        rotated += rot
        prm = dict(kiters=1, liters=1, cgiters=5 + i % 4)
        a, b = synth.lattice_scene(nx, ny, seed=1000 + i)
        outs, its = {}, {}
        for dma in (0, 1):
            pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
            try:
                pl.tune("q_dma", dma)
                outs[dma] = pl.run_host(a, b)
                its[dma] = pl.last_iterations()
            finally:
                pl.tune("q_dma", 1)
                pl.close()
        ndiff = int((outs[0][0] != outs[1][0]).sum() + (outs[0][1] != outs[1][1]).sum())
        d = rel_l2(outs[1][0], outs[1][1], outs[0][0], outs[0][1])
        print(f"PARITY case=tile_walks {nx}x{ny} ({out3[0]} tile columns, border tiles per workgroup {out3[1]} plain / {out3[2]} rotated -> rotation "
              f"{'ON' if rot else 'off'}) {prm}: {ndiff} values differ, rel-L2 {d:.2e}, iterations {its[1]}")
        assert np.isfinite(outs[1][0]).all() and np.isfinite(outs[1][1]).all()
        assert its[0] == its[1] == 3 * prm["cgiters"] and d < 1e-5
    assert 2 <= rotated < len(shapes)                                   # both branches of the host's rule were walked
