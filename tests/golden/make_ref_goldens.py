"""Generates tests/golden/ref_helpers.npz: inputs and outputs of the REFERENCE's own CPU helper
functions (oct_bicubic.cc, oct_gaussian.cc, oct_zoom.cc), compiled unmodified from
/root/reference into oracle/_ref/liboct_ref_helpers.so by `make -C oracle ref`.

Only runs where /root/reference exists (this container).  The .npz is data (inputs + expected
outputs); no reference source text is stored.  Formulas shared with the CUDA hot path:
Catmull-Rom bicubic with truncate-then-clamp indices, Gaussian taps, dropped-last-tap separable
blur + integer-coordinate decimation, half-pixel-shifted bicubic up-sampling.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oct_oracle as oo  # noqa: E402

oo.build()
R = C.CDLL(oo.ref_helpers_path())
D = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

bicubic_float = R._Z17oct_bicubic_floatPfddiii
bicubic_float.restype = C.c_double
bicubic_float.argtypes = [F, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
gauss1d = R._Z18oct_getGaussian_1DPdid
gauss1d.argtypes = [D, C.c_int, C.c_double]
zoom_out = R._Z12oct_zoom_outPdS_iidi
zoom_out.argtypes = [D, D, C.c_int, C.c_int, C.c_double, C.c_int]
zoom_in_float = R._Z17oct_zoom_in_floatPfS_iiiiii
zoom_in_float.argtypes = [F, F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]

zoom_out_float = R._Z18oct_zoom_out_floatPfS_iidii
zoom_out_float.argtypes = [F, F, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]

rng = np.random.RandomState(1234)
out = {}

# 1. bicubic samples (including the truncate-toward-zero quirk for -1 < u < 0 and out-of-range)
nx, ny = 23, 17
img = (rng.rand(ny, nx) * 255).astype(np.float32)
pts = np.concatenate([rng.uniform(-1.5, nx + 1.5, (300, 1)), rng.uniform(-1.5, ny + 1.5, (300, 1))], axis=1)
pts = np.concatenate([pts, [[0.0, 0.0], [-0.25, -0.75], [nx - 1, ny - 1], [nx - 0.5, 3.25], [5.0, 7.0]]])
pts = pts.astype(np.float32).astype(np.float64)   # exactly representable as float
vals = np.array([bicubic_float(img, float(u), float(v), nx, ny, 1) for u, v in pts])
out.update(bic_img=img, bic_pts=pts, bic_vals=vals)

# 2. Gaussian taps for the factors a scaleF=0.5 pyramid uses
taps = {}
for m in range(1, 9):
    f = 0.5 ** m
    sigma = 0.6 * np.sqrt(1.0 / (f * f) - 1.0)
    fs = max(5, int(2 * (1.0 / np.sqrt(2.0 * f))))
    gk = np.zeros(2 * fs + 1)
    gauss1d(gk, 2 * fs + 1, float(np.float32(sigma)))   # the device code holds sigma in a float
    out[f"taps_{m}"] = gk

# 3. blur + decimate at factor 0.5 (oct_zoom_out: sigma=0.6*sqrt(3), window 5, last tap dropped)
nx, ny = 41, 36
img = (rng.rand(ny, nx) * 255).astype(np.float32)
lx, ly = int(nx * 0.5 + 0.5), int(ny * 0.5 + 0.5)
dec = np.zeros((ly, lx))
zoom_out(img.astype(np.float64), dec, nx, ny, 0.5, 0)
out.update(zo_img=img, zo_out=dec)

# 4. flow up-sampling (oct_zoom_in_float with bicubic), odd -> odd sizes so the factor is not 2
cx, cy, fx, fy = 20, 18, 41, 36
flow = (rng.randn(cy, cx) * 2).astype(np.float32)
up = np.zeros((fy, fx), np.float32)
zoom_in_float(flow, up, cx, cy, fx, fy, 0, 1)
out.update(zi_flow=flow, zi_out=up)

# 5. channel resampling as the readers use it (own generator so that the entries above keep their values):
#    oct_zoom_out_float at factors 0.5, 0.4 (non-integer ratio) and 1.0 (the copy branch), channel 0;
#    oct_zoom_in_float with bicubic into channel 1 of a two-channel buffer and nearest into channel 0
rng5 = np.random.RandomState(4321)
nx, ny = 50, 35
img = (rng5.rand(ny, nx) * 255).astype(np.float32)
out["zof_img"] = img
for tag, f in (("05", 0.5), ("04", 0.4), ("10", 1.0)):
    lx, ly = int(nx * f + 0.5), int(ny * f + 0.5)
    o = np.zeros((ly, lx), np.float32)
    zoom_out_float(img, o, nx, ny, f, 0, 0)
    out["zof_out_" + tag] = o
cx, cy, fx, fy = 17, 13, 50, 35
src = (rng5.rand(cy, cx) * 255).astype(np.float32)
two = np.full((2, fy, fx), -1.0, np.float32)
zoom_in_float(src, two, cx, cy, fx, fy, 1, 1)
out.update(zif_src=src, zif_out_c1=two)
near = np.zeros((24, 33), np.float32)          # nearest: sizes for which the reference's unclamped index stays in range
zoom_in_float(src, near, cx, cy, 33, 24, 0, 0)
out.update(zif_near=near)

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_helpers.npz"), **out)
print("wrote ref_helpers.npz with", sorted(out))
