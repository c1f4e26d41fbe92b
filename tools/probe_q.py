"""A/B of tile walks / cache-policy hints of the q-recomputing PCG kernel (timing only: octane_vof_plan_probe holds the stop
test open).  usage: probe_q.py size "xcd:nt,xcd:nt,..." [level]   e.g.  probe_q.py 5000 4:15,8:15,8:527"""
import os as _os  # the stamped kernels live in the diagnostic library (make -C octane_amd/csrc DIAG=1), never in the product
_os.environ.setdefault('OCTANE_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'octane_amd', 'liboctane_vof_diag.so'))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octane_amd import capi
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
combos = [tuple(int(y) for y in x.split(":")) for x in (sys.argv[2] if len(sys.argv) > 2 else "4:15,8:15").split(",")]
os.environ.setdefault('OCTANE_TUNE_PLACEMENT_TRIALS', '8')
for plan_i in range(2):
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8))
    lev = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    res = {}
    for rep in range(4):
        for xcd, nt in combos:
            pl.tune("q_diag", 0 if (xcd == 4 and nt == 15) else 1)      # the production kernel for its own setting, the diagnostic copy otherwise
            pl.tune("xcd", xcd); pl.tune("nt", nt)
            a, b = pl.probe(lev, 40)
            res.setdefault((xcd, nt), []).append(a * 1e3)
    pl.tune("xcd", 4); pl.tune("nt", 15); pl.tune("q_diag", 0)
    print("plan", plan_i, " | ".join(f"xcd{c[0]} nt{c[1]}: {min(res[c]):.1f}" for c in combos), "us (min of 4)", flush=True)
    pl.close()
