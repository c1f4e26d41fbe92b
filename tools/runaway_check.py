#!/usr/bin/env python3
"""the 10848^2 lattice scene with seed 20240616 and 8 pyramid levels "runs away" (flows of ~130 px) at R1's iteration
counts, which DESIGN 4 attributes to the reference's scheme (an aliased 85 x 85 coarsest level, no safeguard) -- on the evidence that
the plain and the banded solve agree, both of which run the same persistent kernel on that level.  This records the missing evidence:
the same pair through (1) the default plan, (2) a plan with the persistent mid-level solve OFF (one launch per iteration, other
kernels) and (3) the CPU ORACLE (oracle/vof_oracle.c, OpenMP, launch-geometry sums), with the flow's size and its distance from the
analytic truth for each, and the distances between them.  ~6 minutes of oracle at 10848^2 on 16 cores; a heartbeat file keeps the
GPU box's watchdog quiet.  With a third argument the frame is also solved as that many row bands -- with the converging seed 20240615
this is BASELINE configs[3] at FULL size against the oracle, once, as a record (too long for the test suite).
With a fourth argument `disc` the pair is the DATA-SHAPED scene of round 5 (synth.disc_scene: the Earth disc on exact zeros, limb taper, int16
counts, noise, saturated patch) -- configs[3] is a full-disk pair -- and the errors against the truth are taken inside the disc.
usage: runaway_check.py [n] [seed] [bands] [disc]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from octane_amd import capi, synth          # the default plan and the row bands run on the PRODUCT library ...
dcapi = capi.dev()                            # ... only the leg with the persistent solve switched off needs the diagnostic one (octane_vof_tune)
from oracle import oct_oracle as oo          # a tool, not the product: the oracle is the checker here


def rel(u, v, uo, vo):
    num = ((u.astype(np.float64) - uo) ** 2 + (v.astype(np.float64) - vo) ** 2).sum()
    return float(np.sqrt(num / ((uo.astype(np.float64) ** 2 + vo.astype(np.float64) ** 2).sum())))


def describe(name, u, v, tu, tv, its, secs):
    n = u.shape[0]; m = n // 8
    eu = np.abs(u[m:-m:8, m:-m:8] - tu[m:-m:8, m:-m:8]).mean(); ev = np.abs(v[m:-m:8, m:-m:8] - tv[m:-m:8, m:-m:8]).mean()
    print(f"{name:34s} iterations {its}, |u|max {np.abs(u).max():9.2f} |v|max {np.abs(v).max():9.2f}, mean |flow - truth| {eu:8.4f} {ev:8.4f} px, {secs:.1f} s", flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10848
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20240616
    prm = dict(kiters=8, liters=3, cgiters=30)
    disc = len(sys.argv) > 4 and sys.argv[4] == "disc"
    a, b = (synth.disc_scene if disc else synth.lattice_scene)(n, n, seed=seed, device="cuda")
    a, b = a.cpu().numpy(), b.cpu().numpy()
    torch.cuda.empty_cache()
    tu, tv = synth.true_lattice_flow(n, n)
    if disc:
        print(f"disc scene: {float((a == 0).mean()):.3f} of the pixels are exact zeros (space)", flush=True)
    print(f"{n} x {n}, seed {seed}, {prm}", flush=True)
    res = {}
    for name, persist in (("HIP, default plan", 1), ("HIP, persistent solve off", 0)):
        lib = capi if persist else dcapi
        pl = lib.Plan(n, n, 1, lib.FlowParams(**prm))
        if not persist:
            pl.tune("persist", 0)
        t = time.time()
        u, v = pl.run_host(a, b)
        describe(name, u, v, tu, tv, pl.last_iterations(), time.time() - t)
        pl.close()
        res[name] = (u, v)
    if len(sys.argv) > 3:                      # the same frame as row bands (virtual bands on a one-GPU box): configs[3]'s mechanism
        nb = int(sys.argv[3])
        tp = capi.TiledPlan(n, n, 1, capi.FlowParams(**prm), nbands=nb, devices=capi.band_devices(nb))
        t = time.time()
        u, v = tp.run_host(a, b)
        describe(f"HIP, {nb} row bands ({tp.banded_levels} banded levels)", u, v, tu, tv, tp.last_iterations(), time.time() - t)
        tp.close()
        res[f"HIP, {nb} row bands"] = (u, v)
    stop = threading.Event()

    def heart():
        t0 = time.time()
        while not stop.wait(30.0):
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "heartbeat_runaway.txt"), "a") as f:
                f.write(f"oracle running for {time.time() - t0:.0f} s\n")
    th = threading.Thread(target=heart, daemon=True); th.start()
    oo.set_threads(oo.host_cpu_share())
    t = time.time()
    uo, vo, its = oo.flow(a, b, oo.FlowParams(**prm), flavour="omp", dot_threads=oo.REF_GRID_THREADS)
    stop.set(); th.join()
    describe(f"CPU oracle ({oo.num_threads('omp')} threads)", uo, vo, tu, tv, its, time.time() - t)
    for name, (u, v) in res.items():
        print(f"relative L2 of '{name}' from the oracle: {rel(u, v, uo, vo):.3e}", flush=True)
    names = list(res)
    for other in names[1:]:
        print(f"relative L2 of '{other}' from '{names[0]}': {rel(*res[other], *res[names[0]]):.3e}", flush=True)


if __name__ == "__main__":
    main()
