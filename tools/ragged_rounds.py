"""Does a ragged last round of tiles cost a round?  us per launch of the finest-level PCG kernel (stop test held open) on frames whose tile count
is an exact multiple of the 512-workgroup grid, just above one, and in between.  (It does not: time follows the pixel count; round 3.)"""
import sys, os
sys.path.insert(0, os.getcwd())
from octane_amd import capi
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
for nx, ny in ((5120, 4096), (5120, 4112), (5120, 4200), (5120, 4300), (5120, 4500), (5000, 5000), (2560, 2560), (2500, 2500), (2560, 2048)):
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(kiters=1, liters=1, cgiters=4))
    us = min(pl.probe(0, 41)[0] * 1e3 for _ in range(3))
    tiles = ((nx + 127) // 128) * ((ny + 15) // 16)
    print(f"{nx}x{ny}: {us:8.2f} us per launch, {tiles} tiles = {tiles / 512:.2f} rounds of 512, {us / (nx * ny) * 1e6:.3f} ps per pixel", flush=True)
    pl.close()
