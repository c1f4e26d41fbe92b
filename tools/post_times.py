"""pix2uv on a 5000 x 5000 flow through the host-buffer entry: wall time of the call (copies included); run under
`rocprofv3 --kernel-trace` and tools/level_times.py for the kernel alone (4.5 ms)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from octane_amd import capi
n = 5000
rng = np.random.RandomState(1)
u = (2.0 * rng.randn(n, n)).astype(np.float32); v = (2.0 * rng.randn(n, n)).astype(np.float32)
nav = capi.Nav(pph=35786023.0, req=6378137.0, rpol=6356752.31414, lam0=-1.308996939, xScale=5.6e-05, xOffset=-0.14, yScale=-5.6e-05,
               yOffset=0.14, g2xOffset=-0.14, g2yOffset=0.14, nx=n, ny=n)
for i in range(3):
    t0 = time.perf_counter(); r = capi.pix2uv(nav, 7.1e8, 7.1e8 + 300.0, u, v); dt = time.perf_counter() - t0
    print("pix2uv 5000^2 host call: %.1f ms" % (dt * 1e3), flush=True)
