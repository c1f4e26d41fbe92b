"""Child process of the self-check drills in tests/test_gpu_tiled.py: loads the DIAGNOSTIC library (OCTANE_LIB; only it has the fault hook
OCTANE_TEST_BREAK_TRANSPORT), creates a row-band plan -- which runs the first-contact self-check --, solves one pair and reports which
transport the plan ended up with and how far its flow is from the plain plan's."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octane_amd import capi, synth  # noqa: E402


def main():
    assert os.path.basename(capi.LIB_PATH) == "liboctane_vof_diag.so", capi.LIB_PATH
    nx, ny, nbands, kit, lit, cg = (int(x) for x in sys.argv[1:7])
    a, b = synth.lattice_scene(nx, ny, seed=29)
    prm = capi.FlowParams(kiters=kit, liters=lit, cgiters=cg)
    try:
        tp = capi.TiledPlan(nx, ny, 1, prm, nbands=nbands, devices=capi.band_devices(nbands), min_band_pixels=1)
    except capi.OctaneError as e:
        print("TILED_RESULT " + json.dumps({"created": False, "msg": str(e)}), flush=True)
        return
    info = tp.transport_info()
    u, v = tp.run_host(a[0], b[0])
    its = tp.last_iterations()
    banded = tp.banded_levels
    tp.close()
    pl = capi.Plan(nx, ny, 1, prm)
    up, vp = pl.run_host(a[0], b[0])
    ip = pl.last_iterations()
    pl.close()
    d = float(np.sqrt((((u - up).astype(np.float64)) ** 2 + ((v - vp).astype(np.float64)) ** 2).sum() /
                      ((up.astype(np.float64)) ** 2 + (vp.astype(np.float64)) ** 2).sum()))
    print("TILED_RESULT " + json.dumps({"created": True, "info": info, "rel_l2": d, "its": its, "its_plain": ip, "banded": banded,
                                        "bits_equal": bool(np.array_equal(u, up) and np.array_equal(v, vp))}), flush=True)


if __name__ == "__main__":
    main()
