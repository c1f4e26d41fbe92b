"""What a launch of the finest-level PCG kernel costs before it moves a byte: us per launch (probe: event pair per launch) on frames of
1 tile ... 16 rounds of 512 tiles.  The intercept of time against pixels is the fixed cost the 2000^2 and 2500^2 levels pay 270 times a pyramid.
usage: [OCTANE_LIB=octane_amd/variants/x.so] python tools/fixed_cost.py [WxH ...]"""
import sys, os
sys.path.insert(0, os.getcwd())
from octane_amd import capi
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
sizes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]] or [(128, 16), (1024, 64), (2048, 256), (2048, 512), (2048, 1024), (2048, 2048), (4096, 2048), (4096, 4096)]
tag = os.path.basename(os.environ.get("OCTANE_LIB", "product"))
for nx, ny in sizes:
    pl = capi.Plan(nx, ny, 1, capi.FlowParams(kiters=1, liters=1, cgiters=4))
    us = min(pl.probe(0, 41)[0] * 1e3 for _ in range(3))
    tiles = ((nx + 127) // 128) * ((ny + 15) // 16)
    print(f"{tag:12s} {nx}x{ny}: {us:8.2f} us per launch, {tiles} tiles = {tiles / 512:.2f} rounds of 512, {us / (nx * ny) * 1e6:.3f} ps per pixel", flush=True)
    pl.close()
