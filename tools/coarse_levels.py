#!/usr/bin/env python3
"""Where does a deep pyramid leave the truth?  Runs the CPU oracle (one flavour / dot-product schedule per call) on a lattice scene and
prints, for the K coarsest pyramid levels, the size of the flow after each linearisation and after the level; the process exits as soon
as level K - 1 is done (the fine levels, which cost the time, are never solved).  With --gpu the HIP path is traced the same way (tune
key trace_levels keeps the stage taps to those levels).  Used for SURVEY 8d's run R3 at 5000^2 (kiters 10: a 10 x 10 coarsest level that is
pure aliasing of the scene's 8 ... 256-pixel wavelengths), profiles/r4_parity_r2_r3_fullsize.txt.
usage: coarse_levels.py n seed kiters liters cgiters K [flavour=strict|fma|omp] [dot_threads] [--gpu] [--save file.npz]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from octane_amd import synth


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n, seed, kit, lit, cg, K = (int(x) for x in args[:6])
    flavour = args[6] if len(args) > 6 else "strict"
    dott = int(args[7]) if len(args) > 7 else 0
    gpu = "--gpu" in sys.argv
    save = sys.argv[sys.argv.index("--save") + 1] if "--save" in sys.argv else None
    if gpu:
        a, b = synth.lattice_scene(n, n, seed=seed, device="cuda")
        a, b = a.cpu().numpy(), b.cpu().numpy()
    else:
        a, b = synth.lattice_scene(n, n, seed=seed)
    out = {}

    def note(tag, k, gnc, l, arr):
        if tag in ("u", "v"):
            print(f"  L{k} gnc{gnc} l{l} |{tag}|max {np.abs(arr).max():12.4f}", flush=True)
        if tag in ("ulev", "vlev"):
            print(f"L{k} ({arr.shape[-1]} x {arr.shape[-2]}) done: |{tag[0]}|max {np.abs(arr).max():12.4f} mean {arr.mean():10.4f}", flush=True)
            out[f"{tag}{k}"] = arr.copy()

    def finish():
        if save:
            np.savez(save, **out)
        sys.stdout.flush()
        os._exit(0)

    if gpu:
        from octane_amd import capi
        capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
        pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=kit, liters=lit, cgiters=cg))
        pl.tune("trace_levels", K)
        store = {}
        pl.set_trace(store)
        u, v = pl.run_host(a, b)
        for key in sorted(store, key=lambda t: (t[1], t[2], t[3], t[0])):
            tag, k, gnc, l = key
            if k < K:
                note(tag, k, gnc, l, store[key][0])
        print(f"HIP path: whole pyramid done, {pl.last_iterations()} iterations, |u|max {np.abs(u).max():.3f} |v|max {np.abs(v).max():.3f}")
        finish()
    from oracle import oct_oracle as oo
    TR = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int)

    def cb(user, tag, k, gnc, l, data, nx, ny, npl):
        tag = tag.decode()
        if k >= K:
            finish()
        if tag in ("u", "v", "ulev", "vlev"):
            note(tag, k, gnc, l, np.ctypeslib.as_array(data, shape=(ny, nx)))

    keep = TR(cb)
    L = oo.lib(flavour)

    class Trace(C.Structure):
        _fields_ = [("cb", TR), ("user", C.c_void_p)]
    t = Trace(keep, None)
    u = np.zeros((n, n), np.float32); v = np.zeros((n, n), np.float32)
    p = oo.FlowParams(kiters=kit, liters=lit, cgiters=cg).c()
    L.oct_oracle_set_dot_schedule(dott)
    if flavour == "omp":
        oo.set_threads(oo.host_cpu_share())
    L.oct_oracle_vof(a[0] if a.ndim == 3 else a, b[0] if b.ndim == 3 else b, n, n, 1, u, v, C.byref(p), C.byref(t))
    finish()


if __name__ == "__main__":
    main()
