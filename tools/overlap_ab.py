"""Level setup on a side stream (OCTANE_TUNE_OVERLAP=1, the default) against everything on one stream: ms per pyramid and whether the flow has the same bits.
usage: overlap_ab.py [size kiters [nchan [lambdac]]] ..."""
import os; os.environ.setdefault("OCTANE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octane_amd", "liboctane_vof_diag.so"))  # the OCTANE_TUNE_* tuning variables exist in the diagnostic library only (round 5)
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octane_amd import capi, synth
cases = [(5000, 8, 1, 0.0), (2000, 6, 1, 0.0), (512, 4, 3, 0.5), (700, 5, 2, 0.0)]
for n, kit, nc, lc in cases:
    a, b = synth.lattice_scene(n, n, seed=20240615, nchan=nc, device="cuda")
    out = {}
    for ov in (0, 1):
        os.environ["OCTANE_TUNE_OVERLAP"] = str(ov)
        pl = capi.Plan(n, n, nc, capi.FlowParams(kiters=kit, lambdac=lc))
        u = torch.full((n, n), 0.25, device="cuda"); v = torch.full((n, n), -0.5, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        best = 1e9
        for rep in range(4):
            u.fill_(0.25); v.fill_(-0.5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), st)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        h = hashlib.sha1(u.cpu().numpy().tobytes()); h.update(v.cpu().numpy().tobytes())
        out[ov] = (best, h.hexdigest()[:12], pl.last_iterations())
        pl.close()
    print(f"{n}x{n}x{nc} kiters {kit} lambdac {lc}: one stream {out[0][0]*1e3:8.3f} ms, side stream {out[1][0]*1e3:8.3f} ms ({(out[1][0]/out[0][0]-1)*100:+.1f} %), "
          f"bits {'same' if out[0][1] == out[1][1] else 'DIFFER'} ({out[0][1]} / {out[1][1]}), iterations {out[0][2]} / {out[1][2]}", flush=True)
