"""Times the row-band solve with virtual bands on one GPU against the plain plan (same frame, same parameters):
what the band bookkeeping, the exchanges and the host-side issue rate cost when no second device helps.
usage: python tools/tiled_virtual.py [n=5000] [kiters=8] [liters=3] [cgiters=30] [bands=1,2,4] [min_band_pixels=0]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from octane_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    kit = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    lit = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cg = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    bands = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "1,2,4").split(",")]
    mbp = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    dev = torch.device("cuda:0")
    a, b = synth.lattice_scene(n, n, seed=20240615, device=dev)
    u0 = torch.zeros(n, n, device=dev)
    v0 = torch.zeros(n, n, device=dev)
    ou, ov = torch.empty(n, n, device=dev), torch.empty(n, n, device=dev)
    prm = capi.FlowParams(kiters=kit, liters=lit, cgiters=cg)
    torch.cuda.synchronize()
    pl = capi.Plan(n, n, 1, prm)
    for _ in range(2):
        pl.run_device(a.data_ptr(), b.data_ptr(), ou.data_ptr(), ov.data_ptr())
        ou.zero_(); ov.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pl.run_device(a.data_ptr(), b.data_ptr(), ou.data_ptr(), ov.data_ptr())
    torch.cuda.synchronize()
    t_plain = time.perf_counter() - t0
    up, vp = ou.cpu().numpy().copy(), ov.cpu().numpy().copy()
    pl.close()
    print(f"plain plan           : {t_plain * 1e3:8.2f} ms  {n * n / t_plain / 1e6:7.2f} Mpix/s", flush=True)
    for nb in bands:
        tp = capi.TiledPlan(n, n, 1, prm, nbands=nb, devices=[0] * nb, min_band_pixels=mbp)
        tp.load_device(a.data_ptr(), b.data_ptr(), u0.data_ptr(), v0.data_ptr())
        tp.solve(); tp.wait()
        t0 = time.perf_counter()
        tp.solve()
        t_issue = time.perf_counter() - t0
        tp.wait()
        t = time.perf_counter() - t0
        tp.fetch_device(ou.data_ptr(), ov.data_ptr())
        ut, vt = ou.cpu().numpy(), ov.cpu().numpy()
        bad = np.where((~np.isfinite(ut)).any(axis=1))[0]
        if len(bad):
            print(f"   non-finite rows: {bad.min()}..{bad.max()} ({len(bad)} rows), iterations {tp.last_iterations()}", flush=True)
        num = ((ut.astype(np.float64) - up) ** 2 + (vt.astype(np.float64) - vp) ** 2).sum()
        den = (up.astype(np.float64) ** 2 + vp.astype(np.float64) ** 2).sum()
        print(f"{nb} virtual band(s)    : {t * 1e3:8.2f} ms  {n * n / t / 1e6:7.2f} Mpix/s  host issue {t_issue * 1e3:7.2f} ms  "
              f"banded levels {tp.banded_levels}  copies {tp.last_copies()}  relL2 vs plain {np.sqrt(num / den):.2e}", flush=True)
        tp.close()


if __name__ == "__main__":
    main()
