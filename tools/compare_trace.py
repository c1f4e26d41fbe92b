"""Debug aid: run the HIP path and the CPU oracle on one small pair with stage taps on both
and print, per stage, the first place they disagree.  (Uses oracle/ as the checker only.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octane_amd import capi, synth
from oracle import oct_oracle as oo

def rel(a, b):
    d = np.sqrt(((a.astype(np.float64) - b) ** 2).sum())
    n = np.sqrt((b.astype(np.float64) ** 2).sum())
    return d / n if n > 0 else d

def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    ny = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    kit = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    nc = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    a, b = synth.lattice_scene(nx, ny, seed=7, nchan=nc)
    prm = capi.FlowParams(kiters=kit)
    oprm = oo.FlowParams(kiters=kit)
    tr_o, tr_g = {}, {}
    uo, vo, its = oo.flow(a, b, oprm, trace=tr_o)
    pl = capi.Plan(nx, ny, nc, prm)
    pl.set_trace(tr_g)
    ug, vg = pl.run_host(a, b)
    print("oracle its", its, "gpu its", pl.last_iterations())
    for key in sorted(tr_g.keys(), key=lambda k: (k[1], k[2], k[3], k[0])):
        tag, k, gnc, l = key
        g = tr_g[key]
        if tag == "coef7":
            o = tr_o[("coef", k, gnc, l)]
            o7 = np.stack([o[0], o[1], o[2], o[5], o[6], o[7], o[8]])
            names = ["a1", "a2", "a4", "wx", "wy", "bu", "bv"]
            msg = " ".join(f"{n}:{rel(g[i], o7[i]):.2e}/{int((g[i] != o7[i]).sum())}" for i, n in enumerate(names))
            print(f"L{k} g{gnc} l{l} coef  {msg}")
        elif tag == "dx2":
            o = tr_o[("dx", k, gnc, l)][0]
            ou, ov = o[:, 0::2], o[:, 1::2]
            print(f"L{k} g{gnc} l{l} dx    u:{rel(g[0], ou):.2e} v:{rel(g[1], ov):.2e}")
        elif key in tr_o:
            o = tr_o[key]
            print(f"L{k} g{gnc} l{l} {tag:5s} rel {rel(g, o):.2e} nmismatch {int((g != o).sum())} of {g.size}")
    print("FINAL relL2:", np.sqrt((((ug - uo) ** 2 + (vg - vo) ** 2).sum()) / ((uo ** 2 + vo ** 2).sum())))
    print("mean flow gpu", ug.mean(), vg.mean(), "oracle", uo.mean(), vo.mean())

if __name__ == "__main__":
    main()
