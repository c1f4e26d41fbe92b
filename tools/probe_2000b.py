"""2000x2000 finest level: does the working set fit the Infinity Cache when x is updated every launch (two p buffers) and
nothing is streamed past it?  Whole-pyramid timing (the probe holds defer_x fixed), several settings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octane_amd import capi, synth
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
n, kit = (int(sys.argv[1]) if len(sys.argv) > 1 else 2000), 6
a, b = synth.lattice_scene(n, n, seed=5, device="cuda")
u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, kn in (("default", {}), ("defer_x=0", dict(defer_x=0)), ("nt=0", dict(nt=0)), ("defer_x=0 nt=0", dict(defer_x=0, nt=0)),
                 ("stored q", dict(fused_q=0)), ("stored q defer_x=0 nt=0", dict(fused_q=0, defer_x=0, nt=0))):
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=kit))
    for k, val in kn.items():
        pl.tune(k, val)
    for _ in range(2):
        pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        u.zero_(); v.zero_()
        pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    pl.tune("fused_q", 1)
    pl.close()
    print(f"{n}x{n} {name:28s} {dt * 1e3:7.2f} ms  {n * n / dt / 1e6:6.1f} Mpix/s", flush=True)
