"""Generates tests/golden/ref_bandminmax.npz: the REFERENCE's own ABI band range table (src/oct_normalize_geo.cc:9-88,
oct_bandminmax), dumped from the unmodified reference source compiled into oracle/_ref/liboct_ref_helpers.so by
`make -C oracle ref` and reached through the forwarding wrapper of oracle/ref_wrap.cc.

For every band number -2 .. 20 the two outputs are preset to a sentinel (-999.25) before the call: bands the reference
knows overwrite both, bands it does not know (anything outside 1 .. 16) leave the sentinel in place -- that "untouched"
behaviour is part of what a drop-in oct_bandminmax has to reproduce (octane_amd/csrc/host_shim.cpp), and the C-ABI
octane_bandminmax reports the same thing as an error code.

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
R.oct_ref_bandminmax.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
R.oct_ref_bandminmax.restype = None

SENTINEL = np.float32(-999.25)
bands = np.arange(-2, 21, dtype=np.int32)
mx = np.zeros(bands.size, np.float32)
mn = np.zeros(bands.size, np.float32)
for k, b in enumerate(bands):
    a, c = C.c_float(float(SENTINEL)), C.c_float(float(SENTINEL))
    R.oct_ref_bandminmax(int(b), C.byref(a), C.byref(c))
    mx[k], mn[k] = a.value, c.value
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_bandminmax.npz")
np.savez_compressed(out, bands=bands, maxch=mx, minch=mn, sentinel=SENTINEL)
print(out, {int(b): (float(x), float(y)) for b, x, y in zip(bands, mx, mn)})
