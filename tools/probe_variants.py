import os; os.environ.setdefault("OCTANE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octane_amd", "liboctane_vof_diag.so"))  # the OCTANE_TUNE_* tuning variables exist in the diagnostic library only (round 5)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octane_amd import capi
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
n = 5000
os.environ['OCTANE_TUNE_PLACEMENT_TRIALS'] = '1'
for plan_i in range(3):
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8))
    res = {}
    for rep in range(2):
        for var in (2, 1, 3, 5):
            pl.tune("pass_a", var)
            for lev in (5, 6, 7):
                a, b = pl.probe(lev, 30)
                res.setdefault((var, lev), []).append(a * 1e3)
    print("plan", plan_i, " | ".join(f"L{lev}: " + " ".join(f"v{var}={min(res[(var, lev)]):.1f}" for var in (2, 1, 3, 5)) for lev in (5, 6, 7)), flush=True)
    pl.close()
