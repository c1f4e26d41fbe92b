"""How far the recovered flow is from the analytic displacement of the lattice scene, for several sizes / parameter sets
(diagnostic: tells a diverging coarse-to-fine run from a kernel problem).  usage: check_truth.py n:kiters:liters:cgiters[:seed] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octane_amd import capi, synth

for spec in sys.argv[1:]:
    f = [int(x) for x in spec.split(":")]
    n, k, l, c = f[:4]
    seed = f[4] if len(f) > 4 else 20240616
    a, b = synth.lattice_scene(n, n, seed=seed, device="cuda")
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=k, liters=l, cgiters=c))
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    tr = {}
    torch.cuda.synchronize()
    pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    its = pl.last_iterations()
    pl.close()
    tu, tv = synth.true_lattice_flow(n, n, xp=torch)
    m = n // 8
    s = max(1, n // 700)
    eu = (u[m:-m:s, m:-m:s].double().cpu() - tu[m:-m:s, m:-m:s]).abs()
    ev = (v[m:-m:s, m:-m:s].double().cpu() - tv[m:-m:s, m:-m:s]).abs()
    print(f"{n}x{n} kiters={k} liters={l} cgiters={c} seed={seed}: iterations {its}, mean |u-tu| {eu.mean():.4f} |v-tv| {ev.mean():.4f}, "
          f"max {eu.max():.2f} {ev.max():.2f}, |u|max {float(u.abs().max()):.2f} |v|max {float(v.abs().max()):.2f}", flush=True)
    del a, b, u, v
