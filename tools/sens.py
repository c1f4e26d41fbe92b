import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np
from conftest import rel_l2
from octane_amd import capi, synth
from oracle import oct_oracle as oo
nx, ny = 90, 70
a, b = synth.lattice_scene(nx, ny, seed=nx*7+ny)
kw = dict(kiters=3, alpha=12.0, lambda_=0.25)
g = oo.REF_GRID_THREADS
uo, vo, _ = oo.flow(a, b, oo.FlowParams(**kw), dot_threads=g)
uf, vf, _ = oo.flow(a, b, oo.FlowParams(**kw), flavour='fma', dot_threads=g)
us, vs, _ = oo.flow(a, b, oo.FlowParams(**kw))
ug, vg = capi.flow(a, b, capi.FlowParams(**kw))
print('floor fma-vs-strict (grid):', rel_l2(uf, vf, uo, vo))
print('serial-vs-grid oracle     :', rel_l2(us, vs, uo, vo))
print('gpu vs strict grid        :', rel_l2(ug, vg, uo, vo))
print('gpu vs fma grid           :', rel_l2(ug, vg, uf, vf))
print('gpu vs serial             :', rel_l2(ug, vg, us, vs))
d = np.abs(ug-uo); j,i = np.unravel_index(np.argmax(d), d.shape); print('max |du|', d.max(), 'at', i, j, ' 99th pct', np.percentile(d, 99))
