"""Slots per thread of the persistent mid-level solve: time per PCG iteration of a one-level solve with sub-domains of at least 1 / 2 / 4 slots
of 8 rows (OCTANE_TUNE_PERSIST_MINP), and the distance of each result to the CPU oracle (the checker).   usage: mid_minp.py [size ...]"""
import os; os.environ.setdefault("OCTANE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octane_amd", "liboctane_vof_diag.so"))  # the OCTANE_TUNE_* tuning variables exist in the diagnostic library only (round 5)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from octane_amd import capi, synth
from oracle import oct_oracle as oo
sizes = [int(a) for a in sys.argv[1:]] or [63, 78, 125, 157, 250, 313, 500, 625]
oo.build(); oo.set_threads(oo.host_cpu_share())
for n in sizes:
    a, b = synth.lattice_scene(n, n, seed=n, device="cuda")
    prm = dict(kiters=1, liters=3, cgiters=30)
    uo, vo, its_o = oo.flow(a.cpu().numpy(), b.cpu().numpy(), oo.FlowParams(**prm), flavour="omp", dot_threads=oo.REF_GRID_THREADS)
    for minp in (4, 2, 1):
        os.environ["OCTANE_TUNE_PERSIST_MINP"] = str(minp)
        pl = capi.Plan(n, n, 1, capi.FlowParams(**prm))
        u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            u.zero_(); v.zero_(); pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), st)
        torch.cuda.synchronize()
        ug, vg = u.cpu().numpy(), v.cpu().numpy()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), st)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        its = pl.last_iterations()
        d = np.sqrt((((ug - uo) ** 2).sum() + ((vg - vo) ** 2).sum()) / ((uo ** 2).sum() + (vo ** 2).sum()))
        print(f"{n}x{n} min slots {minp}: {best * 1e6:7.1f} us per one-level pyramid ({its} iterations, oracle {its_o}), {best * 1e6 / max(1, its):5.2f} us per iteration incl. assembly; distance to the oracle {d:.2e}", flush=True)
        pl.close()
