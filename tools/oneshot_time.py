"""Times the one-shot drop-in call (octane_vof_run: create plan, upload, solve, download, destroy -- what the C++ shim of
oct_variational_optical_flow does per image pair) against a reused plan, host buffers both.
usage: python tools/oneshot_time.py [n=5000] [kiters=8]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from octane_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
kit = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a, b = (x[0] for x in synth.lattice_scene(n, n, seed=20240615))
prm = capi.FlowParams(kiters=kit)
capi.flow(a[:256, :256].copy(), b[:256, :256].copy(), capi.FlowParams(kiters=3))      # runtime warm-up
for rep in range(2):
    t0 = time.perf_counter(); u, v = capi.flow(a, b, prm); t1 = time.perf_counter() - t0
    print(f"one-shot octane_vof_run          : {t1 * 1e3:8.1f} ms  {n * n / t1 / 1e6:7.2f} Mpix/s", flush=True)
t0 = time.perf_counter(); pl = capi.Plan(n, n, 1, prm); tc = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter(); u2, v2 = pl.run_host(a, b); t2 = time.perf_counter() - t0
    print(f"reused plan, host buffers        : {t2 * 1e3:8.1f} ms  {n * n / t2 / 1e6:7.2f} Mpix/s   (plan creation {tc * 1e3:.1f} ms)", flush=True)
t0 = time.perf_counter(); pl.close(); td = time.perf_counter() - t0
print(f"plan destroy {td * 1e3:.1f} ms; results equal: {np.array_equal(u, u2) and np.array_equal(v, v2)}")
