"""Randomised parity sweep: HIP path against the oracle on random frame shapes, channel counts and solver parameters
(first guess and hint term included), with the criterion of tests/test_gpu_parity.py (distance to the primary oracle
below 2e-5; cases on which the oracle's own variants are further apart than half of that are reported as CHAOTIC).  A development tool, not part of the test suite:
   python tools/fuzz_parity.py [cases] [seed] [only_case [forms]]     needs a GPU; prints one line per case and a summary;
   with only_case just that case is run, with a fifth argument the three forms of the PCG iteration are compared on it.
   OCTANE_FUZZ_FAMILY=disc|mixed (default lattice): draw the scenes from synth.disc_scene -- the Earth disc on exact zeros, limb taper,
   int16 counts, noise, a saturated patch, random centre and span (round 5) -- or alternate between the two families.  The zero
   background makes the systems ill-conditioned, so on that family most multi-level cases come out as CHAOTIC (the oracle's own
   variants further apart than 1e-5): the criterion that counts there is "never BAD" (within 3 x the oracle's own spread, equal counts)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from octane_amd import capi, synth  # noqa: E402   (the sweep runs on the PRODUCT library; only the form comparison binds the diagnostic one)
from oracle import oct_oracle as oo  # noqa: E402  (the checker)


def rel_l2(u, v, ur, vr):
    num = np.sum((u.astype(np.float64) - ur) ** 2) + np.sum((v.astype(np.float64) - vr) ** 2)
    den = np.sum(ur.astype(np.float64) ** 2) + np.sum(vr.astype(np.float64) ** 2)
    return float(np.sqrt(num / max(den, 1e-300)))


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # run this case alone (the random draws of the others are still made)
    forms = len(sys.argv) > 4                                     # and compare the three forms of the PCG iteration on it
    oo.build()
    oo.set_threads(oo.host_cpu_share())
    worst, bad, chaotic = 0.0, 0, 0
    for case in range(ncases):
        big = case % 10 == 9                       # every tenth case is large enough for the q-recomputing kernel
        if big:
            nx, ny, nc = int(rng.randint(1800, 2600)), int(rng.randint(1700, 2100)), 1
            prm = dict(kiters=int(rng.randint(1, 3)), liters=1, cgiters=int(rng.randint(3, 9)))
        else:
            nx, ny, nc = int(rng.randint(40, 700)), int(rng.randint(40, 500)), int(rng.choice([1, 1, 2, 3]))
            prm = dict(kiters=int(rng.randint(1, 5)), liters=int(rng.randint(1, 4)), cgiters=int(rng.randint(1, 31)),
                       alpha=float(rng.choice([3.0, 5.0, 8.0])), lambda_=float(rng.choice([0.5, 1.0, 2.0])),
                       dozim=int(rng.choice([0, 1])))
        while min(nx, ny) * 0.5 ** (prm["kiters"] - 1) < 16:
            prm["kiters"] -= 1
        scene_seed = int(rng.randint(1 << 30))
        u0 = v0 = None
        if not big and rng.rand() < 0.4:
            prm["lambdac"] = float(rng.choice([0.0, 0.2, 0.5]))
            u0 = (1.5 * rng.randn(ny, nx)).astype(np.float32)
            v0 = (1.5 * rng.randn(ny, nx)).astype(np.float32)
        if only >= 0 and case != only:
            continue
        fam = os.environ.get("OCTANE_FUZZ_FAMILY", "lattice")
        disc = fam == "disc" or (fam == "mixed" and case % 2 == 1)
        if disc:
            drng = np.random.RandomState(scene_seed)
            a, b = synth.disc_scene(nx, ny, seed=scene_seed, nchan=nc, centre=(float(drng.uniform(0.1, 0.9)), float(drng.uniform(0.1, 0.9))),
                                    span=float(drng.uniform(0.5, 1.1)), noise=float(drng.choice([0.0, 0.6, 1.5])), saturate=bool(drng.rand() < 0.7))
            if u0 is not None:      # a first guess is zero in space, as a -firstguess file's is
                m = a[0] != 0
                u0 = (u0 * m).astype(np.float32); v0 = (v0 * m).astype(np.float32)
        else:
            a, b = synth.lattice_scene(nx, ny, seed=scene_seed, nchan=nc)
        if forms:
            res = {}
            dcapi = capi.dev()      # the tuning knob and the two-pass form exist in the diagnostic library only
            pl = dcapi.Plan(nx, ny, nc, dcapi.FlowParams(**prm))
            for name, knobs in (("q", dict(fused=1, fused_q=1)), ("stored", dict(fused=1, fused_q=0)), ("two_pass", dict(fused=0, fused_q=1))):
                for key, val in knobs.items():
                    pl.tune(key, val)
                res[name] = pl.run_host(a, b, u0, v0)
            pl.tune("fused", 1); pl.tune("fused_q", 1)
            pl.close()
            for n1, n2 in (("q", "stored"), ("q", "two_pass"), ("stored", "two_pass")):
                print(f"   {n1} vs {n2}: {rel_l2(res[n1][0], res[n1][1], res[n2][0].astype(np.float64), res[n2][1].astype(np.float64)):.2e}")
        P = oo.FlowParams(**prm)
        g = oo.REF_GRID_THREADS
        uo, vo, its_o = oo.flow(a, b, P, u0=u0, v0=v0, flavour="omp", dot_threads=g)
        uf, vf, _ = oo.flow(a, b, P, u0=u0, v0=v0, flavour="fma", dot_threads=g)
        floor = rel_l2(uf, vf, uo, vo)
        # the reference's dot product depends on its launch geometry (oracle/vof_oracle.c, dotf): a second, finer geometry
        # is as valid an answer as the first, and on large, heavily truncated solves the two can be further apart than
        # the bar -- the HIP path (fp64 sums) has to match one of them
        us = vs = None
        others = []
        if nx * ny <= 300_000:                      # small frames: the one-thread schedule the survey's answers were recorded with
            us, vs, _ = oo.flow(a, b, P, u0=u0, v0=v0, flavour="omp")
            floor = max(floor, rel_l2(us, vs, uo, vo))
            # three variants undersample the spread: case 128 of seed 17 (372x237x3) has them within 4.8e-6 of each other while five
            # further launch geometries of the same code land 2.8e-5 ... 4.1e-5 away (profiles/r3_fuzz_parity_150.txt) -- two clusters,
            # one rounding apart.  The frame is small, so the further schedules cost milliseconds.
            for fl, dt in (("omp", 128), ("omp", 512), ("omp", 8192), ("fma", 0), ("fma", 2048)):
                x, y, _ = oo.flow(a, b, P, u0=u0, v0=v0, flavour=fl, dot_threads=dt)
                others.append((x, y))
                floor = max(floor, rel_l2(x, y, uo, vo))
        u2 = v2 = None
        if nx * ny > 1_000_000:
            u2, v2, _ = oo.flow(a, b, P, u0=u0, v0=v0, flavour="omp", dot_threads=8 * g)
            floor = max(floor, rel_l2(u2, v2, uo, vo))
        pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
        ug, vg = pl.run_host(a, b, u0, v0)
        its_g = pl.last_iterations()
        pl.close()
        # the tests' criterion (tests/test_gpu_parity.py, _check): distance to the PRIMARY oracle (strict build, the reference's
        # launch geometry) below 2e-5 -- the bar does not move with the problem.  A case above it is either a defect or a parameter
        # set on which the truncated solve amplifies single roundings (the oracle's own variants are then as far apart): the latter
        # kind is what the tests' allow-list is for, and is reported as CHAOTIC with the oracle's spread, never silently passed.
        d = rel_l2(ug, vg, uo, vo)
        d_other = min([rel_l2(ug, vg, x, y) for x, y in [(u2, v2), (us, vs)] + others if x is not None] or [float("nan")])
        bar = 2e-5
        fine = bool(np.isfinite(ug).all()) and its_g == its_o
        verdict = "ok " if (fine and d < bar) else ("CHAOTIC" if (fine and floor > bar / 2 and (not disc or d < 3 * floor)) else "BAD")
        worst = max(worst, d / bar)
        bad += 1 if verdict == "BAD" else 0
        chaotic += 1 if verdict == "CHAOTIC" else 0
        print(f"{verdict} {'disc' if disc else 'lattice'} {nx}x{ny}x{nc} {prm} guess={u0 is not None}: d_primary {d:.2e} (bar {bar:.0e}; oracle spread {floor:.1e}, nearest other oracle variant {d_other:.1e}) its {its_g}/{its_o}", flush=True)
    print(f"{ncases} cases, {bad} bad, {chaotic} chaotic (would need an allow-list entry), worst distance / bar = {worst:.2f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
