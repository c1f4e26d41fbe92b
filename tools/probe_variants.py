import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octane_amd import capi
n = 5000
os.environ['OCTANE_TUNE_PLACEMENT_TRIALS'] = '1'
for plan_i in range(3):
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8))
    print("plan", plan_i)
    for rep in range(2):
        for var in (2, 5, 1, 4, 3):
            pl.tune("pass_a", var)
            out = []
            for lev in (5, 6, 7):
                a, b = pl.probe(lev, 30)
                out.append(f"L{lev} A {a*1e3:7.2f} B {b*1e3:7.2f}")
            print(f"  variant {var}: " + " | ".join(out), flush=True)
    pl.close()
