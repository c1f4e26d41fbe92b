"""Where the waves of one launch of k_pcg_fused_q spend their time (shader-clock stamps at the seams of a tile).
usage: probe_stamps.py [size]"""
import os as _os  # the stamped kernels live in the diagnostic library (make -C octane_amd/csrc DIAG=1), never in the product
_os.environ.setdefault('OCTANE_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'octane_amd', 'liboctane_vof_diag.so'))
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octane_amd import capi
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8))
L = capi.lib()
L.octane_vof_plan_probe_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
names = ["own loads issued", "p staged (wait for p)", "barrier 1", "ring group", "own groups", "barrier 2", "phase 2", "tiles", "prologue (fold)", "epilogue (reduce)"]
for even, unit in ((1, 0), (0, 0), (1, 1), (0, 1)):
    buf = (C.c_ulonglong * 16)()
    rc = L.octane_vof_plan_probe_stamps(pl._h, 7, even, unit, buf)
    assert rc == 0, capi.lib().octane_last_error()
    v = list(buf)[:10]
    nw = 512 * 4
    tiles = v[7] / nw
    tot = sum(v[:7]) + v[8] + v[9]
    print(f"--- {n}x{n} launch with{'' if even else 'out'} x update, unit weights {unit}: {tiles:.2f} tiles per wave, {tot / nw:.0f} cycles per wave")
    for i in (8, 0, 1, 2, 3, 4, 5, 6, 9):
        print(f"  {names[i]:26s} {v[i] / nw:10.0f} cycles/wave  {100.0 * v[i] / tot:5.1f} %   {v[i] / max(1, v[7]):8.0f} per tile")
pl.close()
