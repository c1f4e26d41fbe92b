"""Generates tests/golden/ref_binterp_bc.npz: inputs and outputs of the REFERENCE's own plain-C++ bilinear interpolation
(src/oct_binterp.cc:24-41, oct_binterp_coefs / oct_coef_binterp) and boundary clamp (include/oct_bc.h:1-20, oct_bc<T>),
compiled unmodified from /root/reference into oracle/_ref/liboct_ref_helpers.so by `make -C oracle ref` and reached through
the forwarding wrappers of oracle/ref_wrap.cc.

The solver's warp (ref src/oct_variational_optical_flow.cu:56-71, 727-779) uses float device copies of exactly these two
functions; the oracle's restatement (oracle/vof_oracle.c: clamp_coord, warp_position, bilinear_weights, bilinear_apply) is
pinned against this data by tests/test_oracle_pins.py.

Only runs where /root/reference exists (this container).  The .npz is data (inputs + expected outputs)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oct_oracle as oo  # noqa: E402

oo.build()
R = C.CDLL(oo.ref_helpers_path())
R.oct_ref_binterp_coefs.restype = C.c_double
R.oct_ref_binterp_coefs.argtypes = [C.c_double] * 10 + [C.POINTER(C.c_double)]
R.oct_ref_coef_binterp.restype = C.c_double
R.oct_ref_coef_binterp.argtypes = [C.POINTER(C.c_double)] + [C.c_double] * 4
R.oct_ref_bc_float.restype = C.c_float
R.oct_ref_bc_float.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_int)]
R.oct_ref_bc_double.restype = C.c_double
R.oct_ref_bc_double.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_int)]
R.oct_ref_bc_int.restype = C.c_int
R.oct_ref_bc_int.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]

rng = np.random.RandomState(20241004)
out = {}

# ---- oct_bc<float>, <double>, <int>: values below, inside, on and beyond the range, for several sizes
xs = np.concatenate([rng.uniform(-3, 40, 400), [-1e-7, -0.0, 0.0, 0.5, 30.999998, 31.0, 31.000002, 32.0, 1e9, -1e9, 7.0]]).astype(np.float32)
ns = np.concatenate([rng.randint(2, 36, 400), [32, 32, 32, 32, 32, 32, 32, 32, 5, 5, 8]]).astype(np.int32)
hit = C.c_int()
bcf = np.array([(R.oct_ref_bc_float(float(x), int(n), C.byref(hit)), hit.value) for x, n in zip(xs, ns)])
bcd = np.array([(R.oct_ref_bc_double(float(x), int(n), C.byref(hit)), hit.value) for x, n in zip(xs, ns)])
xi = np.concatenate([rng.randint(-5, 45, 200), [-1, 0, 31, 32, 33]]).astype(np.int32)
ni = np.concatenate([rng.randint(2, 36, 200), [32, 32, 32, 32, 32]]).astype(np.int32)
bci = np.array([(R.oct_ref_bc_int(int(x), int(n), C.byref(hit)), hit.value) for x, n in zip(xi, ni)])
out.update(bc_x=xs, bc_n=ns, bc_float=bcf[:, 0].astype(np.float32), bc_float_hit=bcf[:, 1].astype(np.int32),
           bc_double=bcd[:, 0], bc_double_hit=bcd[:, 1].astype(np.int32),
           bc_xi=xi, bc_ni=ni, bc_int=bci[:, 0].astype(np.int32), bc_int_hit=bci[:, 1].astype(np.int32))

# ---- bilinear: positions (float-representable) inside unit cells of a 37 x 29 level, incl. cell 0, the last cell, positions
# exactly on pixels and on the level's far edges; four corner values in the 0..255 range the solver sees
nx, ny = 37, 29
npts = 600
px = rng.uniform(0, nx - 1, npts); py = rng.uniform(0, ny - 1, npts)
px[:12] = [0.0, 1e-6, 0.25, 1.0, nx - 1.0, nx - 1.000001, nx - 2.0, 17.0, 17.5, 0.999999, 3.0, 35.99]
py[:12] = [0.0, 0.75, 1e-6, 1.0, ny - 1.0, 5.5, ny - 1.000001, 11.0, 0.5, 27.999, ny - 2.0, 27.01]
px = px.astype(np.float32); py = py.astype(np.float32)
x0 = np.minimum(px.astype(np.int32), nx - 2); y0 = np.minimum(py.astype(np.int32), ny - 2)      # ref .cu:738-745
f = (rng.rand(npts, 4) * 255).astype(np.float32)
p = np.zeros((npts, 4)); val = np.zeros(npts); val2 = np.zeros(npts)
buf = (C.c_double * 4)()
for k in range(npts):
    val[k] = R.oct_ref_binterp_coefs(float(px[k]), float(py[k]), float(x0[k]), float(x0[k] + 1), float(y0[k]), float(y0[k] + 1),
                                     *(float(t) for t in f[k]), buf)
    p[k] = list(buf)
    val2[k] = R.oct_ref_coef_binterp(buf, *(float(t) for t in f[k, ::-1]))      # weights re-used on other corner values
out.update(bil_nx=np.int32(nx), bil_ny=np.int32(ny), bil_px=px, bil_py=py, bil_x0=x0, bil_y0=y0, bil_f=f, bil_p=p, bil_val=val,
           bil_val_reused=val2)

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_binterp_bc.npz"), **out)
print("wrote ref_binterp_bc.npz with", sorted(out))
