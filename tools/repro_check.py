import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octane_amd import capi, synth
a, b = synth.lattice_scene(333, 217, seed=4)
prm = capi.FlowParams(kiters=4)
p1 = capi.Plan(333, 217, 1, prm); p2 = capi.Plan(333, 217, 1, prm)
r = [p1.run_host(a, b), p2.run_host(a, b), p1.run_host(a, b), p2.run_host(a, b)]
for i in range(4):
    print(i, "nan" if not np.isfinite(r[i][0]).all() else "finite", "equal to run0:", np.array_equal(r[0][0], r[i][0]) and np.array_equal(r[0][1], r[i][1]),
          "max|du|", np.abs(r[0][0]-r[i][0]).max())
tr1, tr2 = {}, {}
p1.set_trace(tr1); p2.set_trace(tr2)
p1.run_host(a, b); p2.run_host(a, b)
for key in sorted(tr1, key=lambda k: (k[1], k[2], k[3], k[0])):
    if not np.array_equal(tr1[key], tr2[key], equal_nan=True):
        print("first difference at", key, "count", int((tr1[key] != tr2[key]).sum()), "nan1", int(np.isnan(tr1[key]).sum()), "nan2", int(np.isnan(tr2[key]).sum()))
        d = np.argwhere(tr1[key] != tr2[key])[:10]
        print(d)
        break
else:
    print("traces identical")
