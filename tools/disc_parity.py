"""Round 5: HIP path against the oracle on DATA-SHAPED inputs -- synth.disc_scene: the Earth disc on exact zeros, the limb
taper of ref src/oct_navcal_cuda.cu:81-93, radiances through int16 counts, sensor noise, a saturated patch, 1 - 3 channels.

   python tools/disc_parity.py [quick|full] [growth]       needs a GPU; prints one line per case

The zero background makes the linear systems ill-conditioned (no data term outside the disc: a pure, weighted Laplacian that 30 PCG
iterations do not converge), so single roundings are amplified ~100 x more than on the lattice scenes: the oracle's OWN valid variants
(FMA-contracted build, 8 x finer launch geometry, one-thread sums) are 1e-4 ... 4e-4 apart on multi-level solves.  Each line therefore
carries the distance of the HIP path to the primary oracle AND the oracle's spread, both over the whole frame and inside the disc.
`growth` prints, for one multi-level case, how the distance develops linearisation by linearisation (the first assembly of the coarsest
level is bit-exact; what follows is the problem's own amplification), for the HIP path and for the oracle's FMA build side by side."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octane_amd import capi, synth  # noqa: E402
from oracle import oct_oracle as oo  # noqa: E402  (the checker)


def rl(u, v, uo, vo, m=None):
    du = (u.astype(np.float64) - uo) ** 2 + (v.astype(np.float64) - vo) ** 2
    dn = uo.astype(np.float64) ** 2 + vo.astype(np.float64) ** 2
    if m is not None:
        du, dn = du[m], dn[m]
    return float(np.sqrt(du.sum() / max(dn.sum(), 1e-300)))


def case(nx, ny, nc, prm, kw, guess=False, tag=""):
    a, b = synth.disc_scene(nx, ny, seed=nx * 3 + ny, nchan=nc, **kw)
    m = synth.disc_mask(nx, ny, kw.get("centre", (0.5, 0.5)), kw.get("span", 1.0))
    inside = m == 1
    u0 = v0 = None
    if guess:            # a first guess that is not zero near the limb (and zero in space, as -firstguess files are)
        tu, tv = synth.true_lattice_flow(nx, ny)
        u0 = (0.8 * tu * m).astype(np.float32); v0 = (0.8 * tv * m).astype(np.float32)
    P = oo.FlowParams(**prm)
    g = oo.REF_GRID_THREADS
    t = time.time()
    uo, vo, io = oo.flow(a, b, P, u0=u0, v0=v0, flavour="omp", dot_threads=g)
    to = time.time() - t
    var = {}
    var["fma"] = oo.flow(a, b, P, u0=u0, v0=v0, flavour="fma", dot_threads=g)[:2]
    var["grid_x8"] = oo.flow(a, b, P, u0=u0, v0=v0, flavour="omp", dot_threads=8 * g)[:2]
    if nx * ny <= 400_000:
        var["serial"] = oo.flow(a, b, P, u0=u0, v0=v0, flavour="omp")[:2]
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    ug, vg = pl.run_host(a, b, u0, v0)
    ig = pl.last_iterations()
    pl.close()
    d, di = rl(ug, vg, uo, vo), rl(ug, vg, uo, vo, inside)
    sp = {k: rl(x, y, uo, vo) for k, (x, y) in var.items()}
    spi = {k: rl(x, y, uo, vo, inside) for k, (x, y) in var.items()}
    dnear = min(rl(ug, vg, x, y) for x, y in var.values())
    verdict = "ok" if (d < 2e-5 and ig == io) else ("ILL-CONDITIONED" if (ig == io and d < 3 * max(sp.values())) else "BAD")
    print(f"{verdict} {tag} {nx}x{ny}x{nc} {prm} {kw} guess={guess}: d_primary {d:.2e} (inside the disc {di:.2e}); oracle spread "
          f"{ {k: f'{x:.1e}' for k, x in sp.items()} } inside { {k: f'{x:.1e}' for k, x in spi.items()} }; nearest variant {dnear:.1e}; "
          f"its {ig}/{io}; finite {bool(np.isfinite(ug).all())}; oracle {to:.1f} s", flush=True)
    return verdict


def growth(nx=600, ny=560, prm=None):
    prm = prm or dict(kiters=4)
    a, b = synth.disc_scene(nx, ny, seed=5)
    g = oo.REF_GRID_THREADS
    tro, trf, trg = {}, {}, {}
    oo.flow(a, b, oo.FlowParams(**prm), flavour="strict", dot_threads=g, trace=tro)
    oo.flow(a, b, oo.FlowParams(**prm), flavour="fma", dot_threads=g, trace=trf)
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
    pl.set_trace(trg)
    pl.run_host(a, b)
    pl.close()
    print(f"growth of the distance to the strict oracle, {nx}x{ny} {prm}: per (level, gnc, l) after the solve -- HIP path | oracle FMA build; "
          f"coefficient planes of that linearisation bit-equal?")
    for key in sorted(k for k in tro if k[0] == "u"):
        _, k, gnc, l = key
        uo, vo = tro[("u", k, gnc, l)][0], tro[("v", k, gnc, l)][0]
        dg = rl(trg[("u", k, gnc, l)][0], trg[("v", k, gnc, l)][0], uo, vo)
        df = rl(trf[("u", k, gnc, l)][0], trf[("v", k, gnc, l)][0], uo, vo)
        cg = trg[("coef7", k, gnc, l)]; co = tro[("coef", k, gnc, l)]
        eq = all(np.array_equal(cg[gi], co[oi]) for gi, oi in zip(range(7), (0, 1, 2, 5, 6, 7, 8)))
        print(f"  level {k} gnc {gnc} l {l} ({uo.shape[1]}x{uo.shape[0]}): hip {dg:.2e} | fma {df:.2e} | coef bit-equal {eq}", flush=True)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    oo.build()
    oo.set_threads(oo.host_cpu_share())
    bad = 0
    cases = [
        (300, 280, 1, dict(kiters=1), {}, False, "single-level"),
        (300, 280, 1, dict(kiters=1, liters=1, cgiters=10), {}, True, "single-level"),
        (300, 280, 1, dict(kiters=4), {}, False, "multi-level"),
        (300, 280, 1, dict(kiters=4), dict(noise=0.0), False, "multi-level, plateaus"),
        (320, 300, 2, dict(kiters=3, liters=2, cgiters=12), {}, True, "two channels"),
        (260, 300, 3, dict(kiters=3, liters=1, cgiters=8), {}, False, "three channels"),
        (400, 360, 1, dict(kiters=4), dict(centre=(0.1, 0.2), span=0.6), False, "limb through a corner"),
        (300, 280, 1, dict(kiters=3, dozim=0), {}, False, "-brox"),
        (2300, 1900, 1, dict(kiters=1, liters=1, cgiters=7), {}, False, "q kernel, disc edge through interior tiles"),
        (2300, 1900, 1, dict(kiters=1, liters=1, cgiters=7), dict(centre=(0.2, 0.3), span=0.7), True, "q kernel, limb + first guess"),
        (2090, 1730, 2, dict(kiters=2, liters=1, cgiters=6), {}, True, "q kernel, two channels"),
    ]
    if mode == "full":
        cases += [
            (2712, 2712, 1, dict(kiters=8, liters=3, cgiters=30), {}, False, "configs[3] quarter scale"),
            (2000, 2000, 1, dict(kiters=6, liters=3, cgiters=30), dict(centre=(0.15, 0.1), span=0.55), False, "configs[1] shape, limb in a corner"),
        ]
    for c in cases:
        bad += case(*c[:5], guess=c[5], tag=c[6]) == "BAD"
    if "growth" in sys.argv:
        growth()
    print(f"{len(cases)} cases, {bad} bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
