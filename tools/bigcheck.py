import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octane_amd import capi, synth
from oracle import oct_oracle as oo
nx, ny = int(sys.argv[1]), int(sys.argv[2])
kw = dict(kiters=int(sys.argv[3]), liters=int(sys.argv[4]), cgiters=int(sys.argv[5]))
a, b = synth.lattice_scene(nx, ny, seed=77)
tr_o, tr_g = {}, {}
uo, vo, its = oo.flow(a, b, oo.FlowParams(**kw), trace=tr_o, dot_threads=int(os.environ.get('DOT', '40960')))
pl = capi.Plan(nx, ny, 1, capi.FlowParams(**kw)); pl.set_trace(tr_g)
ug, vg = pl.run_host(a, b)
print("its", its, pl.last_iterations())
def rel(g, o):
    n = np.sqrt((o.astype(np.float64)**2).sum()); return np.sqrt(((g.astype(np.float64)-o)**2).sum())/(n if n>0 else 1)
for key in sorted(tr_g, key=lambda k: (k[1], k[2], k[3], k[0])):
    tag,k,g,l = key
    if tag == "coef7":
        o = tr_o[("coef",k,g,l)]; o7 = np.stack([o[0],o[1],o[2],o[5],o[6],o[7],o[8]])
        print(key, " ".join("%.1e" % rel(tr_g[key][i], o7[i]) for i in range(7)), "nan", int(np.isnan(tr_g[key]).sum()))
    elif tag == "dx2":
        o = tr_o[("dx",k,g,l)][0]; print(key, "%.2e %.2e" % (rel(tr_g[key][0], o[:,0::2]), rel(tr_g[key][1], o[:,1::2])))
    elif key in tr_o and tag in ("u","v","u0","v0","ulev","vlev"):
        d = np.abs(tr_g[key]-tr_o[key])[0]; j,i = np.unravel_index(np.argmax(d), d.shape)
        print(key, "%.2e" % rel(tr_g[key], tr_o[key]), "max|d| %.2e at (x=%d,y=%d)" % (d.max(), i, j), "nan", int(np.isnan(tr_g[key]).sum()))
print("FINAL", rel(np.stack([ug,vg]), np.stack([uo,vo])))
