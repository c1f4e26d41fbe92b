"""Mean us per finest-level PCG launch of a real one-level 5000^2 solve, by kind: first GNC step (unit weights) / varying weights, with / without
x work (every launch of the profiled run is timed by its own event pair: octane_vof_plan_get_launch_times).  OCTANE_LIB selects a variant build."""
import sys, os, json
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/octane_amd") else os.getcwd())
import torch
from octane_amd import capi, synth
n = 5000
a, b = synth.lattice_scene(n, n, seed=20240615, device="cuda")
u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=1, liters=3, cgiters=30))
s = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    u.zero_(); v.zero_()
    pl.set_profiling(rep > 0)
    pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
    torch.cuda.synchronize()
lt = pl.launch_times()
acc = {}
for i, ms in enumerate(lt):
    k, g0 = i % 30, (i // 30) // 3 == 0
    if k == 0: continue
    key = ("unit" if g0 else "vary") + ("+x" if k % 2 == 0 else "")
    acc.setdefault(key, []).append(ms)
print("BYKIND", {k: round(sum(v) / len(v) * 1e3, 1) for k, v in acc.items()}, flush=True)
