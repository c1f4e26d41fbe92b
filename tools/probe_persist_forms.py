"""Which form of the persistent mid-level solve is off: one launch, 1 / 7 iterations per launch, against each other (bits) and
against the one-launch-per-iteration kernels (rel-L2).  usage: probe_persist_forms.py [nx ny cgiters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octane_amd import capi, synth
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
nx, ny, cg = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (640, 500, 9)))
a, b = synth.lattice_scene(nx, ny, seed=nx + ny)
prm = dict(kiters=1, liters=1, cgiters=cg)
def run(**knobs):
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(**prm))
    try:
        for k, v in knobs.items():
            pl.tune(k, v)
        u, v = pl.run_host(a, b)
        return u, v, pl.last_iterations()
    finally:
        pl.close()
def nd(x, y): return int((x[0] != y[0]).sum() + (x[1] != y[1]).sum())
def rl(x, y): return float(np.sqrt(((x[0] - y[0]) ** 2 + (x[1] - y[1]) ** 2).sum() / max(1e-30, (y[0] ** 2 + y[1] ** 2).sum())))
ref = run(persist=0)
forms = {"P": run(), "P'": run(), "S1": run(persist_step=1), "S1'": run(persist_step=1), "S7": run(persist_step=7), "S2": run(persist_step=2)}
for k, v in forms.items():
    print(f"{k:4s} iterations {v[2]}  rel-L2 vs per-launch kernels {rl(v, ref):.3e}")
for x, y in (("P", "P'"), ("S1", "S1'"), ("S1", "S7"), ("S1", "S2"), ("P", "S1"), ("P", "S7")):
    print(f"{x} vs {y}: {nd(forms[x], forms[y])} values differ")
