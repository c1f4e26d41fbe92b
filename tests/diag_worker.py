"""Child process of test_gpu_persist.py::test_stamped_diagnostic_build...: runs with OCTANE_LIB pointing at the DIAGNOSTIC library
(liboctane_vof_diag.so, `make -C octane_amd/csrc DIAG=1`) -- the product library contains neither the stamped kernel copies nor their
exports, so a process that wants them loads the other library."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octane_amd import capi, synth  # noqa: E402


def run(a, b, prm, **knobs):
    pl = capi.Plan(a.shape[-1], a.shape[-2], 1, capi.FlowParams(**prm))
    try:
        for k, v in knobs.items():
            pl.tune(k, v)
        u, v = pl.run_host(a, b)
        its = pl.last_iterations()
    finally:
        for k in knobs:
            pl.tune(k, 0)
        pl.close()
    h = hashlib.sha1(u.tobytes()); h.update(v.tobytes())
    return h.hexdigest(), its


def main():
    assert os.path.basename(capi.LIB_PATH) == "liboctane_vof_diag.so", capi.LIB_PATH
    nx, ny, prm = 640, 500, dict(kiters=1, liters=1, cgiters=9)
    a, b = synth.lattice_scene(nx, ny, seed=5)
    L = capi.lib()
    L.octane_vof_mid_stamps.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * 32)()
    plain, ip = run(a, b, prm)
    assert L.octane_vof_mid_stamps(0, buf) == 0            # clear
    diag, idg = run(a, b, prm, persist_diag=1)
    assert L.octane_vof_mid_stamps(0, buf) == 0
    print("DIAG_RESULT " + json.dumps({"plain": plain, "diag": diag, "its_plain": ip, "its_diag": idg, "stamps": list(buf)}), flush=True)


if __name__ == "__main__":
    main()
