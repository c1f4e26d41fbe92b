import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octane_amd import capi
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8))
for rep in range(2):
    for lev in range(2, 8):
        a, b = pl.probe(lev, 30)
        w = int(n * 0.5 ** (7 - lev) + 0.5)
        print(f"level {lev} ({w}x{w}): A {a*1e3:8.2f} us  B {b*1e3:8.2f} us   A/B {a/b:.2f}", flush=True)
