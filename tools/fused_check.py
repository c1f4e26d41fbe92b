"""Fused one-kernel-per-iteration PCG against the two-pass form on the same inputs: distance between the two results,
iteration counts, and time.  usage: python tools/fused_check.py [n=2000] [kiters=6] [liters=3] [cgiters=30]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from octane_amd import capi, synth  # noqa: E402
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lit = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cg = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device("cuda:0")
a, b = synth.lattice_scene(n, n, seed=20240615, device=dev)
prm = capi.FlowParams(kiters=kit, liters=lit, cgiters=cg)
res = {}
for fused in (0, 1):
    pl = capi.Plan(n, n, 1, prm)
    pl.tune("fused", fused)
    u = torch.zeros(n, n, device=dev); v = torch.zeros(n, n, device=dev)
    torch.cuda.synchronize()
    for rep in range(3):
        u.zero_(); v.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[fused] = (u.cpu().numpy(), v.cpu().numpy(), pl.last_iterations(), dt)
    pl.close()
    print(f"fused={fused}: {dt * 1e3:8.2f} ms  {n * n / dt / 1e6:7.2f} Mpix/s  iterations {res[fused][2]}", flush=True)
(u0, v0, i0, _), (u1, v1, i1, _) = res[0], res[1]
num = ((u1.astype(np.float64) - u0) ** 2 + (v1.astype(np.float64) - v0) ** 2).sum()
den = (u0.astype(np.float64) ** 2 + v0.astype(np.float64) ** 2).sum()
print(f"relL2 fused vs two-pass: {np.sqrt(num / den):.3e}; finite: {np.isfinite(u1).all() and np.isfinite(v1).all()}; iterations equal: {i0 == i1}")
d = np.hypot(u1 - u0, v1 - v0)
iy, ix = np.unravel_index(np.argmax(d), d.shape)
print(f"max |diff| {d.max():.3e} at (x={ix}, y={iy}); 99.9th percentile {np.percentile(d, 99.9):.3e}; median {np.median(d):.3e}; mean |flow| {np.hypot(u0, v0).mean():.3f}")
rows = d.max(axis=1); cols = d.max(axis=0)
print("rows with the largest diffs:", np.argsort(rows)[-6:][::-1].tolist(), " cols:", np.argsort(cols)[-6:][::-1].tolist())
