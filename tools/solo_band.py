"""Round 5: the compute term of the multi-GPU scaling model, measured.

The pool gives one GPU per box, so an N-GPU solve of one frame cannot be timed.  This tool runs ONLY band b's launch sequence of an
N-band solve (octane_vof_solo_band_time, diagnostic library: the replicated set-ups and coarse levels, the band's assemblies, PCG
launches and flow updates, an event record at every phase boundary; the neighbours' rows and partial blocks static -- wrong flow,
right timeline) for N = 1, 2, 4, 8 and every distinct kind of band (first, inner, last), and prints the per-band pyramid time next to
the plain plan's.  An N-GPU pyramid cannot be faster than its slowest band's solo time; what comes on top (waiting at boundaries,
xGMI latency, peer copies) is what tools/tiled_model.py prices.

   OCTANE_LIB=octane_amd/liboctane_vof_diag.so python tools/solo_band.py [n=10848] [kiters=8] [liters=3] [cgiters=30] [bands=2,4,8] [min_band_pixels=0] [reps=3]

Output: one line per (N, band); profiles/r5_solo_band.txt."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OCTANE_LIB", os.path.join(ROOT, "octane_amd", "liboctane_vof_diag.so"))
import torch  # noqa: E402
from octane_amd import capi, synth  # noqa: E402


def solo(L, n, prm, nb, band, mbp, a, b, reps):
    ms = (C.c_double * reps)()
    its, pbytes, bnd = C.c_longlong(), C.c_longlong(), C.c_longlong()
    nbanded, rows = C.c_int(), C.c_int()
    lev = (C.c_double * 32)()
    rc = L.octane_vof_solo_band_time(n, n, 1, C.byref(prm.c()), nb, band, mbp, a.data_ptr(), b.data_ptr(), 1, reps, ms,
                                     C.byref(its), C.byref(pbytes), C.byref(bnd), C.byref(nbanded), C.byref(rows), lev)
    if rc != 0:
        raise capi.OctaneError(rc, "octane_vof_solo_band_time")
    solo.levels = [x for x in lev if x >= 0]
    return list(ms), its.value, pbytes.value, bnd.value, nbanded.value, rows.value


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10848
    kit = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    lit = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cg = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    bands = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "2,4,8").split(",")]
    mbp = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    L = capi.lib()
    assert os.path.basename(capi.LIB_PATH) == "liboctane_vof_diag.so", "the solo-band entry lives in the diagnostic library: OCTANE_LIB=octane_amd/liboctane_vof_diag.so"
    vp = C.c_void_p
    L.octane_vof_solo_band_time.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(capi.VofParams), C.c_int, C.c_int, C.c_longlong, vp, vp, C.c_int, C.c_int,
                                            C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    dev = torch.device("cuda:0")
    a, b = synth.lattice_scene(n, n, seed=20240615, device=dev)
    ou, ov = torch.empty(n, n, device=dev), torch.empty(n, n, device=dev)
    prm = capi.FlowParams(kiters=kit, liters=lit, cgiters=cg)
    expect = kit * 3 * lit * cg
    torch.cuda.synchronize()
    pl = capi.Plan(n, n, 1, prm)
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        pl.solve_device(a.data_ptr(), b.data_ptr(), ou.data_ptr(), ov.data_ptr())       # zero first guess
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    assert pl.last_iterations() == expect
    pl.close()
    t_plain = min(ts[1:])
    print(f"{n}x{n} kiters={kit} liters={lit} cgiters={cg}, banding threshold {mbp or 'default (4 Mi pixels)'}; plain plan {t_plain:.2f} ms per pyramid "
          f"(best of {reps}); solo times are best of {reps}", flush=True)
    # N = 1 through the band machinery (one band = the whole frame, nothing banded): what the band code itself costs
    ms, its, pb, bnd, nbd, rows = solo(L, n, prm, 1, 0, mbp, a, b, reps)
    print(f"N=1 band 0: {min(ms):8.2f} ms  (banded levels {nbd}, boundaries {bnd}, iterations {its}/{expect}); GPU ms per level, coarsest first: "
          f"{' '.join(f'{x:.2f}' for x in solo.levels)}", flush=True)
    for nb in bands:
        kinds = sorted({0, nb // 2, nb - 1})
        worst = 0.0
        for band in kinds:
            ms, its, pb, bnd, nbd, rows = solo(L, n, prm, nb, band, mbp, a, b, reps)
            ok = "ok" if its == expect else "ITERATION COUNT DIFFERS: not a measurement"
            worst = max(worst, min(ms))
            print(f"N={nb} band {band} ({'first' if band == 0 else 'last' if band == nb - 1 else 'inner'}, {rows} rows of the finest level): {min(ms):8.2f} ms "
                  f"(all reps {', '.join(f'{x:.2f}' for x in ms)}); banded levels {nbd}; phase boundaries {bnd}; peer copies {pb / 1e6:.1f} MB; "
                  f"iterations {its}/{expect} {ok}; GPU ms per level, coarsest first: {' '.join(f'{x:.2f}' for x in solo.levels)}", flush=True)
        print(f"N={nb}: slowest band {worst:.2f} ms -> compute-only speed-up {t_plain / worst:.2f} x, efficiency {t_plain / worst / nb:.3f} "
              f"(an upper bound: boundaries' waiting, xGMI latency and peer copies come on top)", flush=True)


if __name__ == "__main__":
    main()
