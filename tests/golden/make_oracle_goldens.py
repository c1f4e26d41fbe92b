"""Generates tests/golden/oracle_flow.npz: small seeded input pairs and the CPU oracle's u,v for
them.  These pin the oracle against accidental drift and give the GPU box committed
expected outputs.  (The oracle itself is pinned to the reference only as far as its header
says: see oracle/vof_oracle.c.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oct_oracle as oo  # noqa: E402
from octane_amd import synth  # noqa: E402

CASES = {
    # name: (nx, ny, nchan, params, first_guess)
    "s1_64": dict(kind="gauss", nx=64, ny=64, nc=1, shift=(1.5, -0.75), prm=dict()),
    "lat_96x80_k3": dict(kind="lattice", nx=96, ny=80, nc=1, seed=7, prm=dict(kiters=3)),
    "lat_75x53_k2_nc2": dict(kind="lattice", nx=75, ny=53, nc=2, seed=11, prm=dict(kiters=2, liters=2, cgiters=12)),
    "lat_60x44_brox_hint": dict(kind="lattice", nx=60, ny=44, nc=1, seed=5,
                                prm=dict(kiters=2, dozim=0, lambdac=0.5, alpha=8.0, lambda_=0.5), guess=(2.0, -1.0)),
}


def make_inputs(c):
    if c["kind"] == "gauss":
        a, b = synth.gaussian_scene(c["nx"], c["shift"], c["ny"])
        a, b = a[None], b[None]
    else:
        a, b = synth.lattice_scene(c["nx"], c["ny"], seed=c["seed"], nchan=c["nc"])
    u0 = v0 = None
    if "guess" in c:
        u0 = np.full((c["ny"], c["nx"]), c["guess"][0], np.float32)
        v0 = np.full((c["ny"], c["nx"]), c["guess"][1], np.float32)
    return a, b, u0, v0


if __name__ == "__main__":
    out = {}
    for name, c in CASES.items():
        a, b, u0, v0 = make_inputs(c)
        u, v, its = oo.flow(a, b, oo.FlowParams(**c["prm"]), u0=u0, v0=v0)
        out[name + "_img1"] = a
        out[name + "_img2"] = b
        out[name + "_u"] = u
        out[name + "_v"] = v
        out[name + "_its"] = np.int64(its)
        print(name, "its", its, "mean", u.mean(), v.mean())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_flow.npz"), **out)
