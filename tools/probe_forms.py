"""Fused PCG kernel forms timed in isolation on one level (the plan's placement probe: synthetic sums, real planes).
usage: python tools/probe_forms.py [size ...]     prints ms per launch for: 128x8 tile, 128x16 tile, 128x16 q-recomputing"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from octane_amd import capi  # noqa: E402
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)

sizes = [int(a) for a in sys.argv[1:]] or [625, 1250, 2500, 5000]
for n in sizes:
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=1, liters=1, cgiters=30))
    out = []
    for name, rows, q in (("128x8", 1, 0), ("128x16", 2, 0), ("128x16q", 2, 1)):
        pl.tune("fused_rows", rows)
        pl.tune("fused_q", q)
        best = min(sum(pl.probe(0, 40)) for _ in range(3))
        out.append(f"{name} {best * 1e3:8.1f} us")
    pl.tune("fused_rows", 0)
    pl.tune("fused_q", 1)
    pl.close()
    print(f"{n:5d}^2  " + "   ".join(out), flush=True)
