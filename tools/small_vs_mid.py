import os; os.environ.setdefault("OCTANE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octane_amd", "liboctane_vof_diag.so"))  # the OCTANE_TUNE_* tuning variables exist in the diagnostic library only (round 5)
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from octane_amd import capi, synth
os.environ["OCTANE_TUNE_PERSIST_MINP"] = "1"
for n in (39, 45, 50, 55):
    a, b = synth.lattice_scene(n, n, seed=n, device="cuda")
    for smax in (3072, 1024):
        os.environ["OCTANE_TUNE_SMALL_MAX"] = str(smax)
        pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=1, liters=3, cgiters=30))
        u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), st)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), st)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        print(f"{n}x{n} ({n*n} px) small_max {smax}: {best*1e6:7.1f} us per one-level pyramid, {best*1e6/270:5.2f} us per iteration", flush=True)
        pl.close()
