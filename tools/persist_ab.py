"""Persistent mid-level solve on / off: ms per pyramid for small frames whose every level is 'mid' (timing A/B in one process).
usage: persist_ab.py [size kiters]..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octane_amd import capi, synth
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)

specs = [(int(a), int(b)) for a, b in zip(sys.argv[1::2], sys.argv[2::2])] or [(1250, 4), (2000, 6)]
for n, kit in specs:
    a, b = synth.lattice_scene(n, n, seed=5, device="cuda")
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for persist in (1, 0, 1, 0):
        pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=kit))
        pl.tune("persist", persist)
        for _ in range(2):
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            u.zero_(); v.zero_()
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
        torch.cuda.synchronize()
        res.setdefault(persist, []).append((time.perf_counter() - t0) / reps * 1e3)
        its = pl.last_iterations()
        pl.close()
    print(f"{n}x{n} kiters={kit}: persistent {min(res[1]):.2f} ms, one launch per iteration {min(res[0]):.2f} ms  ({its} iterations; {n * n / min(res[1]) / 1e3:.1f} vs {n * n / min(res[0]) / 1e3:.1f} Mpix/s)", flush=True)
