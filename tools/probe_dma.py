"""LDS-DMA staging of p in the q-recomputing kernel (pcg_fused_q_dma.hip) against the register-staged production kernel:
bit-equality of the flow on a ragged frame and at 5000x5000, and time per finest-level iteration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octane_amd import capi, synth
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)

def run(nx, ny, prm, dma):
    a, b = synth.lattice_scene(nx, ny, seed=nx + ny)
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
    pl.tune("q_dma", dma)
    u, v = pl.run_host(a, b)
    its = pl.last_iterations()
    pl.tune("q_dma", 0)
    pl.close()
    return u, v, its

for nx, ny, prm in ((2300, 1900, dict(kiters=1, liters=1, cgiters=7)), (2503, 1699, dict(kiters=2, liters=1, cgiters=11))):
    u0, v0, i0 = run(nx, ny, prm, 0)
    u1, v1, i1 = run(nx, ny, prm, 1)
    print(f"{nx}x{ny} {prm}: {int((u0 != u1).sum() + (v0 != v1).sum())} values differ, iterations {i0}/{i1}, finite {bool(np.isfinite(u1).all())}", flush=True)
n = 5000
pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8))
for lev in (7, 6):
    r = {}
    for rep in range(3):
        for dma in (0, 1):
            pl.tune("q_dma", dma)
            r.setdefault(dma, []).append(pl.probe(lev, 40)[0] * 1e3)
    print(f"{n}x{n} level {lev}: registers {min(r[0]):.1f} us, LDS-DMA {min(r[1]):.1f} us per (even) iteration", flush=True)
pl.tune("q_dma", 0)
pl.close()
