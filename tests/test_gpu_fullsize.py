"""BASELINE.json's configurations against the oracle AT FULL SIZE (round 3; VERDICT r2, "Next round" item 1).

The earlier full-size tests compare the HIP path with the analytic truth and with itself.  Here the whole pyramid --
every level, every hand-off between the kernels that solve them (the LDS-DMA q-recomputing kernel on the two finest
levels, the stored-q kernel, the persistent mid-level solves, the single-workgroup solve: ref .cu:487-1205 is one loop)
-- is compared with the CPU oracle on the same inputs:

* configs[1]: 2000 x 2000, kiters 6, liters 3, cgiters 30;
* configs[2] = R1, the headline run: 5000 x 5000, kiters 8, liters 3, cgiters 30 (2160 PCG iterations);
* configs[3] at quarter scale, 2712 x 2712 with R1's parameters, as four row bands AND as a plain plan (10848^2 is ~8
  minutes of oracle time; the band code is size-independent and the full size is compared with the plain plan bit for
  bit in test_gpu_tiled.py);
* the R2 / R3 parameter sets of SURVEY 8d (liters 10 / cgiters 10; kiters 10 / liters 10) on frames the oracle
  finishes in seconds; kiters = 10 reaches a 10-pixel-wide coarsest level, as R3 at 5000^2 does.

The oracle is the OpenMP build of oracle/vof_oracle.c (bit-identical to the scalar strict build under the
launch-geometry dot schedule: tests/test_oracle_structure.py) with the reference's launch geometry (40960 summing
threads, ref .cu:1422).  Bars: the north-star 1e-4 is asserted HARD; anything above 2e-5 is printed as INVESTIGATE
 and, for the cases that have been measured, asserted too.  Iteration counts have to be equal."""
import time

import numpy as np
import pytest

from conftest import rel_l2
from octane_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

BAR = 1e-4
INVESTIGATE = 2e-5


def _oracle(oracle, a, b, prm, dot_threads=None, threads=None):
    """The OpenMP oracle on the cores this process may use.  A run of minutes prints nothing: a heartbeat thread keeps a
    file under gpurun_out/ fresh (the GPU boxes take seven silent minutes for a hang) -- and pytest's captured stdout would
    not count."""
    import os
    import threading
    oracle.set_threads(threads or oracle.host_cpu_share())
    stop = threading.Event()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    beat = os.path.join(root, "gpurun_out", "heartbeat_fullsize.txt")

    def heart():
        t0 = time.time()
        while not stop.wait(30.0):
            try:
                os.makedirs(os.path.dirname(beat), exist_ok=True)
                with open(beat, "a") as f:
                    f.write(f"oracle {a.shape} {prm} running for {time.time() - t0:.0f} s\n")
            except OSError:
                pass

    th = threading.Thread(target=heart, daemon=True)
    th.start()
    t = time.time()
    try:
        uo, vo, its = oracle.flow(a, b, oracle.FlowParams(**prm), flavour="omp",
                                  dot_threads=oracle.REF_GRID_THREADS if dot_threads is None else dot_threads)
    finally:
        stop.set()
        th.join()
    return uo, vo, its, time.time() - t


# ---- the oracle legs of this file, by name: scene, parameters, dot schedules ------------------------------------------------
# The oracle is CPU-only and releases the GIL inside ctypes, and the ~300 s it needs for the cases below used to be most of the GPU
# suite's wall time (534 s of the driver's 900 s limit in round 5).  Round 6: tests/conftest.py moves this file's tests to the END of
# the session and starts ONE worker thread at collection time that computes the selected cases' oracle legs, in test order, while the
# other test modules keep the GPU busy; a test then only waits for its case (usually ready) and runs its own GPU leg and comparisons.
# Same inputs, same oracle calls, same assertions as before -- only when they run changed.  (vof_oracle.c's dot-schedule switch is
# thread-local, so the foreground tests' own oracle calls do not interfere.)  Without the worker -- one test selected by hand, or the
# worker failed -- a case is computed in place.
def _cuda_scene(fn, *args, **kw):
    import torch
    a, b = fn(*args, device="cuda", **kw)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    torch.cuda.empty_cache()
    return a, b


R1 = dict(kiters=8, liters=3, cgiters=30)
CASES = {   # name: (scene builder, solver parameters, {leg: dot_threads (None = the reference's launch geometry)})
    "config1_2000": (lambda: synth.lattice_scene(2000, 2000, seed=20240614), dict(kiters=6, liters=3, cgiters=30), {"primary": None}),
    "headline_R1_5000": (lambda: _cuda_scene(synth.lattice_scene, 5000, 5000, seed=20240615), R1, {"primary": None}),
    "config3_quarter_2712": (lambda: synth.lattice_scene(2712, 2712, seed=20240615), R1, {"primary": None}),
    "R2_640x512": (lambda: synth.lattice_scene(640, 512, seed=20240614), dict(kiters=8, liters=10, cgiters=10), {"primary": None}),
    "R3_5000x800": (lambda: synth.gaussian_scene(5000, (3.0, -2.0), ny=800), dict(kiters=10, liters=10, cgiters=30), {"primary": None, "one_thread": 0}),
    "R2_5000": (lambda: _cuda_scene(synth.lattice_scene, 5000, 5000, seed=20240615), dict(kiters=8, liters=10, cgiters=10), {"primary": None}),
    "config3_quarter_disc": (lambda: synth.disc_scene(2712, 2712, seed=2712 * 3 + 2712), R1, {"primary": None, "grid_x8": -8}),
    "headline_R1_5000_disc": (lambda: _cuda_scene(synth.disc_scene, 5000, 5000, seed=20240615), R1, {"primary": None}),
    "config1_2000_nc3_disc": (lambda: synth.disc_scene(2000, 2000, seed=7, nchan=3, centre=(0.15, 0.1), span=0.55), dict(kiters=6, liters=3, cgiters=30),
                              {"primary": None}),
}


def uses_case(name):
    def deco(fn):
        fn._oracle_case = name
        return fn
    return deco


class _Prefetch:
    def __init__(self):
        self.events, self.results, self.thread = {}, {}, None

    def start(self, oracle, names):
        import threading
        names = [n for n in dict.fromkeys(names) if n in CASES]
        if not names or self.thread is not None:
            return
        for n in names:
            self.events[n] = threading.Event()
        # all the cores: the oracle scales almost linearly to 16 threads (tools/oracle_scaling.py on a GPU box: 3000^2 in 10.9 / 6.9 / 6.1 s on
        # 8 / 14 / 16 threads; two cases side by side on 7 threads each 12.3 s against 13.3 s one after the other on 14) and the suite's end is
        # oracle-bound; the foreground tests' own small oracle runs and numpy share them with it
        nthreads = max(1, oracle.host_cpu_share())

        def work():
            for n in names:
                try:
                    self.results[n] = _compute_case(oracle, n, nthreads)
                except BaseException as e:      # noqa: BLE001 -- handed to the test that asks for the case
                    self.results[n] = e
                self.events[n].set()

        self.thread = threading.Thread(target=work, name="oracle-prefetch", daemon=True)
        self.thread.start()

    def get(self, oracle, name):
        ev = self.events.get(name)
        if ev is None:
            return _compute_case(oracle, name, None)
        t0 = time.time()
        while not ev.wait(20.0):
            assert self.thread.is_alive() or ev.is_set(), "the oracle worker died"
        r = self.results.pop(name)
        if isinstance(r, BaseException):
            raise r
        r["waited_s"] = time.time() - t0
        return r


def _compute_case(oracle, name, threads):
    build, prm, legs = CASES[name]
    a, b = build()
    out = {"a": a, "b": b, "prm": prm, "legs": {}, "prefetched": threads is not None}
    for leg, dt in legs.items():
        if dt is not None and dt < 0:
            dt = -dt * oracle.REF_GRID_THREADS
        out["legs"][leg] = _oracle(oracle, a, b, prm, dot_threads=dt, threads=threads)
    return out


PREFETCH = _Prefetch()


def _case(oracle, name):
    """(a, b, prm, legs) of a case: legs[leg] = (uo, vo, iterations, oracle seconds)."""
    r = PREFETCH.get(oracle, name)
    if r.get("prefetched"):
        print(f"ORACLE-PREFETCH case={name}: computed ahead by the worker thread ({', '.join(f'{k} {v[3]:.1f} s' for k, v in r['legs'].items())}); "
              f"this test waited {r.get('waited_s', 0.0):.1f} s for it")
    return r["a"], r["b"], dict(r["prm"]), r["legs"]


def _plain(capi, a, b, prm):
    nc, ny, nx = (1,) + a.shape if a.ndim == 2 else a.shape
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    try:
        t = time.time()
        ug, vg = pl.run_host(a, b)
        return ug, vg, pl.last_iterations(), time.time() - t
    finally:
        pl.close()


def _report(case, shape, prm, d, its_o, its_g, t_o, t_g, extra=""):
    flag = "  ** INVESTIGATE (> 2e-5) **" if d > INVESTIGATE else ""
    print(f"PARITY-FULLSIZE case={case} {shape} {prm}: d_primary={d:.3e} (north-star bar {BAR:.0e}){flag} "
          f"iterations oracle/gpu={its_o}/{its_g}; oracle {t_o:.1f} s on {oracle_cores()} threads, gpu call {t_g:.2f} s {extra}")


def oracle_cores():
    from oracle import oct_oracle
    return oct_oracle.num_threads("omp")


@uses_case("config1_2000")
def test_config1_2000_six_levels_matches_oracle(capi, oracle):
    """BASELINE.json configs[1]: 2000 x 2000, 6 pyramid levels (kiters 6, liters 3, cgiters 30 -> 1620 PCG iterations): the
    finest level on the LDS-DMA q-recomputing kernel with rotated tile columns (16 tile columns divide the grid), 1000^2
    ... 63^2 on the persistent solves."""
    n = 2000
    a, b, prm, legs = _case(oracle, "config1_2000")
    uo, vo, io, to = legs["primary"]
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    d = rel_l2(ug, vg, uo, vo)
    _report("config1_2000", f"{n}x{n}", prm, d, io, ig, to, tg)
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    assert io == ig == 6 * 3 * 3 * 30
    assert d < INVESTIGATE


@uses_case("headline_R1_5000")
def test_headline_r1_5000_eight_levels_matches_oracle(capi, oracle):
    """BASELINE.json configs[2] = SURVEY 8d's R1, the configuration bench.py's headline number is measured on: 5000 x 5000,
    kiters 8, liters 3, cgiters 30.  The scene is built on the device (the CPU would take longer over the cosines than the
    GPU over the flow), copied to the host, and the SAME float32 arrays go to the oracle and through octane_vof_run's
    host-buffer path.  ~100 s and ~7 GB of oracle on the GPU box's 16 cores."""
    n = 5000
    a, b, prm, legs = _case(oracle, "headline_R1_5000")
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    uo, vo, io, to = legs["primary"]
    d = rel_l2(ug, vg, uo, vo)
    tu, tv = synth.true_lattice_flow(n, n)
    m = n // 8
    eo = (np.abs(uo - tu)[m:-m, m:-m].mean(), np.abs(vo - tv)[m:-m, m:-m].mean())
    eg = (np.abs(ug - tu)[m:-m, m:-m].mean(), np.abs(vg - tv)[m:-m, m:-m].mean())
    _report("headline_R1_5000", f"{n}x{n}", prm, d, io, ig, to, tg,
            f"[mean |flow - truth|: oracle {eo[0]:.4f}, {eo[1]:.4f} px, gpu {eg[0]:.4f}, {eg[1]:.4f} px; "
            f"CPU oracle {n * n / to / 1e6:.3f} Mpix/s]")
    assert io == ig == 8 * 3 * 3 * 30
    assert d < BAR
    assert d < INVESTIGATE


@uses_case("config3_quarter_2712")
def test_config3_quarter_scale_2712_four_bands_and_plain_match_oracle(capi, oracle):
    """BASELINE.json configs[3] at a quarter of its linear size -- 2712 x 2712, R1's parameters, four row bands with the
    banding threshold scaled by 1/16 so that, as at 10848^2, the two finest levels are banded and the rest replicated --
    and the plain plan on the same pair, both against the oracle."""
    n = 2712
    a, b, prm, legs = _case(oracle, "config3_quarter_2712")
    uo, vo, io, to = legs["primary"]
    up, vp, ip, tp_s = _plain(capi, a, b, prm)
    tp = capi.TiledPlan(n, n, 1, capi.FlowParams(**prm), nbands=4, devices=capi.band_devices(4),
                        min_band_pixels=(12 << 20) // 16)      # (round 3's threshold, scaled: two banded levels)
    try:
        nbanded = tp.banded_levels
        ut, vt = tp.run_host(a, b)
        it = tp.last_iterations()
    finally:
        tp.close()
    dp, dt = rel_l2(up, vp, uo, vo), rel_l2(ut, vt, uo, vo)
    _report("config3_quarter_2712_plain", f"{n}x{n}", prm, dp, io, ip, to, tp_s)
    _report("config3_quarter_2712_4bands", f"{n}x{n}", prm, dt, io, it, to, 0.0,
            f"[banded levels {nbanded}; banded vs plain {rel_l2(ut, vt, up, vp):.2e}]")
    assert nbanded == 2
    assert io == ip == it == 8 * 3 * 3 * 30
    assert dp < INVESTIGATE and dt < INVESTIGATE


@uses_case("R2_640x512")
def test_r2_parameter_set_matches_oracle(capi, oracle):
    """SURVEY 8d's R2: kiters 8, liters 10, cgiters 10 (exactly 300 PCG iterations per level; cgiters is an OFFlags field
    without a command-line flag, ref include/offlags.h:53) at 640 x 512: 240 assemblies, a 5 x 4 coarsest level."""
    nx, ny = 640, 512
    a, b, prm, legs = _case(oracle, "R2_640x512")
    uo, vo, io, to = legs["primary"]
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    d = rel_l2(ug, vg, uo, vo)
    _report("R2_640x512", f"{nx}x{ny}", prm, d, io, ig, to, tg)
    assert io == ig == 8 * 3 * 10 * 10
    assert d < INVESTIGATE


@uses_case("R3_5000x800")
def test_r3_parameter_set_ten_levels_matches_oracle(capi, oracle):
    """SURVEY 8d's R3 ("300 warps"): kiters 10, liters 10, cgiters 30.  kiters = 10 scales the frame by 1 / 512: a 5000-wide
    frame gets a 10-pixel-wide coarsest level, as R3 at 5000^2 does (ref .cu:49-54,488-489); 800 rows make it 10 x 2, the
    smallest level the solver accepts, and keep the oracle below a minute.  Frames whose coarsest level is 2 x 2 or 3 x 3
    diverge in the oracle itself (flows of 1e7 px, NaN) -- the reference's scheme, not a parity question -- so the smooth
    translating-Gaussian scene (S1) is used, on which the oracle's strict and FMA builds are 9e-6 apart.

    Iteration counts: with 300 linearisations per level the coarse solves converge to the tolerance (ref .cu:1131,
    `residc > tol`) and stop early, and WHEN is decided by a residual norm sitting at the rounding threshold: the oracle's own
    valid variants stop after 8942 (strict build, launch-geometry sums), 8917 (FMA-contracted build) and 8960 (one-thread
    sums) of the 9000 iterations (measured, this scene); the HIP path -- fp64 sums, r.r by recurrence -- after 8970.  This is
    the one test where the counts cannot be asked to be equal: they have to lie within 1 % of each other, the flow within
    the bar.  (Every other oracle test asserts equal counts, including the early-exit case of test_gpu_parity.py.)"""
    nx, ny = 5000, 800
    a, b, prm, legs = _case(oracle, "R3_5000x800")
    uo, vo, io, to = legs["primary"]
    us, vs, is_, _ = legs["one_thread"]                                  # the reference's one-thread schedule: another valid count
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    d = rel_l2(ug, vg, uo, vo)
    _report("R3_5000x800", f"{nx}x{ny}", prm, d, io, ig, to, tg,
            f"[oracle, one-thread sums: {is_} iterations, {rel_l2(us, vs, uo, vo):.2e} from the primary; gpu {rel_l2(ug, vg, us, vs):.2e} from it]")
    assert np.isfinite(ug).all()
    assert io < 10 * 3 * 10 * 30 and ig < 10 * 3 * 10 * 30
    assert abs(ig - io) <= 0.01 * io
    assert d < INVESTIGATE


@uses_case("R2_5000")
def test_r2_parameter_set_at_5000_matches_oracle(capi, oracle):
    """Round 5: SURVEY 8d's R2 -- kiters 8, liters 10, cgiters 10: 300 PCG iterations per level, 240 assemblies -- on the
    bench's own 5000 x 5000 pair (until round 4 a tools/ record, profiles/r4_parity_r2_r3_fullsize.txt: 1.7e-5).  ~90 s of oracle on the
    GPU box's cores."""
    n = 5000
    a, b, prm, legs = _case(oracle, "R2_5000")
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    uo, vo, io, to = legs["primary"]
    d = rel_l2(ug, vg, uo, vo)
    _report("R2_5000", f"{n}x{n}", prm, d, io, ig, to, tg)
    assert io == ig == 8 * 3 * 10 * 10
    assert d < INVESTIGATE


@uses_case("config3_quarter_disc")
def test_config3_quarter_scale_disc_scene_four_bands_and_plain_match_oracle(capi, oracle):
    """Round 5: BASELINE.json configs[3] is a FULL-DISK pair -- the Earth disc on exact zeros, the limb taper, counts,
    noise, a saturated patch (synth.disc_scene; ref src/oct_navcal_cuda.cu:81-93) -- at a quarter of its linear size, R1's parameters:
    the plain plan and four row bands (the disc edge crosses every band), both against the oracle.  Banding regroups the fp64 partial
    sums only; on the lattice scenes that has never moved a bit of the flow, on this scene the two groupings round an alpha differently
    somewhere and the zero background's conditioning carries it to 1.1e-5 (measured; the banded flow is the CLOSER of the two to the
    oracle, 6.8e-6 against 1.05e-5) -- the bar between the two is the library's own (2e-5, the self-check's)."""
    n = 2712
    a, b, prm, legs = _case(oracle, "config3_quarter_disc")
    uo, vo, io, to = legs["primary"]
    uf, vf, _, _ = legs["grid_x8"]                                       # 8 x the launch geometry: the oracle's own spread on this case
    up, vp, ip, tp_s = _plain(capi, a, b, prm)
    tp = capi.TiledPlan(n, n, 1, capi.FlowParams(**prm), nbands=4, devices=capi.band_devices(4), min_band_pixels=(12 << 20) // 16)
    try:
        nbanded = tp.banded_levels
        ut, vt = tp.run_host(a, b)
        it = tp.last_iterations()
    finally:
        tp.close()
    dp, dt, floor = rel_l2(up, vp, uo, vo), rel_l2(ut, vt, uo, vo), rel_l2(uf, vf, uo, vo)
    m = synth.disc_mask(n, n) == 1
    _report("config3_quarter_disc_plain", f"{n}x{n}", prm, dp, io, ip, to, tp_s, f"[inside the disc {rel_l2(up[m], vp[m], uo[m], vo[m]):.2e}; oracle grid x 8 vs primary {floor:.2e}]")
    _report("config3_quarter_disc_4bands", f"{n}x{n}", prm, dt, io, it, to, 0.0, f"[banded levels {nbanded}; banded vs plain {rel_l2(ut, vt, up, vp):.2e}]")
    assert nbanded == 2 and io == ip == it == 8 * 3 * 3 * 30
    assert (a == 0).mean() > 0.2
    assert dp < INVESTIGATE and dt < INVESTIGATE
    assert rel_l2(ut, vt, up, vp) < INVESTIGATE


@uses_case("headline_R1_5000_disc")
def test_headline_r1_5000_disc_scene_matches_oracle(capi, oracle):
    """Round 5: the headline configuration (5000 x 5000, kiters 8, liters 3, cgiters 30) on the DATA-SHAPED scene -- the Earth disc filling
    the frame's width on exact zeros (22 % space pixels), limb taper, int16 counts, sensor noise, a saturated patch (synth.disc_scene;
    ref src/oct_navcal_cuda.cu:81-93).  The finest levels run the LDS-DMA q-recomputing kernel with its border-free interior tiles
    crossing the disc edge, the mid-size levels the persistent solves.  At this size the oracle's own variants agree to ~1e-5, so the
    suite's 2e-5 is asserted against the primary oracle, with equal iteration counts."""
    n = 5000
    a, b, prm, legs = _case(oracle, "headline_R1_5000_disc")
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    uo, vo, io, to = legs["primary"]
    d = rel_l2(ug, vg, uo, vo)
    m = synth.disc_mask(n, n) == 1
    tu, tv = synth.true_lattice_flow(n, n)
    eg = (np.abs(ug - tu)[m].mean(), np.abs(vg - tv)[m].mean())
    _report("headline_R1_5000_disc", f"{n}x{n}", prm, d, io, ig, to, tg,
            f"[{float((a == 0).mean()):.2f} of the pixels are exact zeros; inside the disc {rel_l2(ug[m], vg[m], uo[m], vo[m]):.2e}; "
            f"mean |flow - truth| inside the disc {eg[0]:.3f}, {eg[1]:.3f} px]")
    assert io == ig == 8 * 3 * 3 * 30
    assert d < BAR
    assert d < INVESTIGATE


@uses_case("config1_2000_nc3_disc")
def test_config1_2000_three_channels_disc_scene_matches_oracle(capi, oracle):
    """Round 5: BASELINE configs[1]'s size with THREE channels (the reference's loop handles 1 ... 3 alike, ref .cu:749-829; the file reader
    resamples channels 2 and 3 onto channel 1's grid) on the data-shaped scene with the limb through a corner of the frame: the
    three-channel template instances of k_assemble at a BASELINE size, all levels."""
    n = 2000
    a, b, prm, legs = _case(oracle, "config1_2000_nc3_disc")
    uo, vo, io, to = legs["primary"]
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    d = rel_l2(ug, vg, uo, vo)
    _report("config1_2000_nc3_disc", f"{n}x{n}x3", prm, d, io, ig, to, tg, f"[{float((a[0] == 0).mean()):.2f} of the pixels are exact zeros]")
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    assert io == ig == 6 * 3 * 3 * 30
    assert d < INVESTIGATE


def test_r3_300_warps_at_5000_matches_the_committed_oracle_flow(capi):
    """SURVEY 8d's R3 -- kiters 10, liters 10, cgiters 30: the "300 warps" of BASELINE.json's metric string (ref src/main.cc:82,85,258-265;
    stop test .cu:1131) -- AT 5000 x 5000, on the bench's own pair.  Four minutes of oracle per run do not fit the suite, so the oracle's
    flow was computed once (tests/golden/make_r3_5000_oracle_golden.py, on a GPU box: same device-built scene, the oracle's FMA-contracted
    OpenMP build with the reference's launch-geometry sums -- the strict build's aliased 10 x 10 level runs away on this scene, EXPERIMENTS 4)
    and is committed REDUCED: every 16th pixel, 16 x 16 block means (every pixel enters one), fp64 sums.  Coarse solves stop by the
    tolerance test, so the iteration counts are within 1 % of each other, not equal (see test_r3_parameter_set_ten_levels_matches_oracle)."""
    import hashlib
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "golden", "r3_5000_oracle.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/r3_5000_oracle.npz has not been generated")
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_r3_5000_oracle_golden import reduce_flow
    g = np.load(path)
    n = int(g["n"])
    a, b = _cuda_scene(synth.lattice_scene, n, n, seed=20240615)
    same_inputs = hashlib.sha1(a.tobytes()).hexdigest() == str(g["sha1_a"]) and hashlib.sha1(b.tobytes()).hexdigest() == str(g["sha1_b"])
    prm = dict(kiters=int(g["kiters"]), liters=int(g["liters"]), cgiters=int(g["cgiters"]))
    ug, vg, ig, tg = _plain(capi, a, b, prm)
    assert np.isfinite(ug).all() and np.isfinite(vg).all()
    r = reduce_flow(ug, vg)
    d_pts = rel_l2(r["u_pts"], r["v_pts"], g["u_pts"], g["v_pts"])
    d_blk = rel_l2(r["u_blk"], r["v_blk"], g["u_blk"], g["v_blk"])
    io = int(g["its"])
    nrm = {k: abs(float(r[k]) - float(g[k])) / abs(float(g[k])) for k in ("u_sum", "v_sum", "u_sq", "v_sq")}
    print(f"PARITY-FULLSIZE case=R3_5000_golden {n}x{n} {prm}: d_points={d_pts:.3e} ({g['u_pts'].size} pixels) d_block_means={d_blk:.3e} "
          f"({g['u_blk'].size} blocks of 16 x 16) sums {({k: f'{x:.1e}' for k, x in nrm.items()})} (north-star bar {BAR:.0e}) "
          f"iterations oracle/gpu={io}/{ig} (cap {10 * 3 * 10 * 30}); inputs bit-identical to the fixture's: {same_inputs}; "
          f"oracle (fma_omp, {int(g['oracle_threads'])} threads) took {float(g['oracle_seconds']):.0f} s when the fixture was made, gpu call {tg:.2f} s")
    assert io < 10 * 3 * 10 * 30 and ig < 10 * 3 * 10 * 30 and abs(ig - io) <= 0.01 * io
    assert d_pts < INVESTIGATE and d_blk < INVESTIGATE
    assert max(nrm.values()) < 1e-5
