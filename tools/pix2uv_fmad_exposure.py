#!/usr/bin/env python3
"""how exposed is the bit-exactness claim of pix2uv to nvcc's default -fmad=true?  The reference's kernel
(ref src/oct_pix2uv_cuda.cu:27-171,196-197) is built by nvcc, which by default contracts a * b + c into one fused operation --
in float (the base position xi * xScale + xOffset, p2u:40-44) AND in double (the projection formulas).  No CUDA here, so it cannot be
measured on the reference; what CAN be counted is how many of the `short` outputs change when the SAME formulas are compiled with and
without contraction: oracle/pix2uv_oracle.c strict (-ffp-contract=off: what the HIP kernel and the tests use) against its
FMA-contracted build (gcc -mfma -ffp-contract=fast contracts every eligible float and double expression; nvcc's choice of which
products to fuse may differ in detail, so this is an estimate of the exposure, not the reference's answer).
Runs on the CPU (the oracle is the subject here, not the checker); the navigation cases are those of tests/test_gpu_pix2uv.py plus a
5000 x 5000 CONUS-like frame.  Output: profiles/r4_pix2uv_fmad_exposure.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oct_oracle as oo


def nav(**kw):
    n = oo.Nav()
    for k, v in kw.items():
        setattr(n, k, v)
    return n


def cases():
    geo = dict(pph=35786023.0, req=6378137.0, rpol=6356752.31414)
    rng = np.random.RandomState(0)
    nx, ny = 500, 300
    yield ("CONUS 2 km, window at (100, 50), random flow sigma 3 px, dt 300 s", nav(**geo, lam0=-75.0 * 3.14159265 / 180.0, xScale=5.6e-05, xOffset=-0.101332,
           yScale=-5.6e-05, yOffset=0.128212, g2xOffset=-0.101332, g2yOffset=0.128212, minX=100, minY=50, nx=nx, ny=ny),
           (rng.randn(ny, nx) * 3).astype(np.float32), (rng.randn(ny, nx) * 3).astype(np.float32), 0, 0.0, 300.0)
    nx = ny = 64
    yield ("sub-satellite 2 km, u = 1.5 px (the recorded reference answer, 983 cm/s)", nav(**geo, lam0=-75.0 * 3.14159265 / 180.0, xScale=5.6e-05, xOffset=-0.0018,
           yScale=-5.6e-05, yOffset=0.0018, g2xOffset=-0.0018, g2yOffset=0.0018, nx=nx, ny=ny),
           np.full((ny, nx), 1.5, np.float32), np.zeros((ny, nx), np.float32), 0, 0.0, 300.0)
    nx = ny = 340
    rng = np.random.RandomState(1)
    yield ("full disk 32 km incl. limb and space pixels, dt 600 s", nav(**geo, lam0=-1.308996939, xScale=8.96e-04, xOffset=-0.151872, yScale=-8.96e-04,
           yOffset=0.151872, g2xOffset=-0.151872, g2yOffset=0.151872, nx=nx, ny=ny),
           (rng.rand(ny, nx) * 4 - 2).astype(np.float32), (rng.rand(ny, nx) * 4 - 2).astype(np.float32), 0, 1000.0, 1600.0)
    nx, ny = 200, 120
    rng = np.random.RandomState(2)
    u = rng.randn(ny, nx).astype(np.float32); v = rng.randn(ny, nx).astype(np.float32)
    for lat1 in (90.0, 70.0):
        yield (f"polar stereographic, lat1 = {lat1:.0f}, dt 86400 s", nav(xScale=1000.0, xOffset=-100000.0, yScale=1000.0, yOffset=-60000.0, g2xOffset=-100000.0,
               g2yOffset=-60000.0, lat1=lat1, lon0=-45.0, R=6371228.0, nx=nx, ny=ny), u, v, 1, 0.0, 86400.0)
    yield ("mercator, dt 600 s", nav(xScale=2000.0, xOffset=-200000.0, yScale=2000.0, yOffset=1000000.0, g2xOffset=-200000.0, g2yOffset=1000000.0,
           lon1=-1.2, R=6371228.0, nx=nx, ny=ny), u, v, 2, 0.0, 600.0)
    nx = ny = 5000
    rng = np.random.RandomState(3)
    yield ("CONUS-like 5000 x 5000 at 1 km, smooth flow of 1-4 px + noise, dt 300 s", nav(**geo, lam0=-75.0 * 3.14159265 / 180.0, xScale=2.8e-05, xOffset=-0.07,
           yScale=-2.8e-05, yOffset=0.126, g2xOffset=-0.07, g2yOffset=0.126, nx=nx, ny=ny),
           (2.5 + 1.5 * np.sin(np.arange(ny)[:, None] / 800.0) + 0.3 * rng.randn(ny, nx)).astype(np.float32),
           (-1.0 + np.cos(np.arange(nx)[None, :] / 800.0) + 0.3 * rng.randn(ny, nx)).astype(np.float32), 0, 0.0, 300.0)


def run(flavour, n, t1, t2, u, v, mode):
    L = oo.lib(flavour)
    L.oct_oracle_pix2uv.argtypes = oo.lib().oct_oracle_pix2uv.argtypes
    L.oct_oracle_pix2uv.restype = oo.lib().oct_oracle_pix2uv.restype
    import ctypes as C
    sz = u.size
    out = [np.zeros(sz, np.int16) for _ in range(4)]
    dT = C.c_float()
    L.oct_oracle_pix2uv(C.byref(n), t1, t2, np.ascontiguousarray(u).ravel(), np.ascontiguousarray(v).ravel(), 0, mode, *out, C.byref(dT))
    return out


def main():
    tot = bad = 0
    print("pix2uv: oracle built strict (-ffp-contract=off) against the same source built FMA-contracted (gcc -mfma -ffp-contract=fast)")
    for name, n, u, v, mode, t1, t2 in cases():
        s = run("strict", n, t1, t2, u, v, mode)
        f = run("fma", n, t1, t2, u, v, mode)
        d = [int((a != b).sum()) for a, b in zip(s, f)]
        mx = [int(np.abs(a.astype(np.int32) - b).max()) for a, b in zip(s, f)]
        nz = int((s[0] != 0).sum())
        tot += 2 * u.size; bad += d[0] + d[1]
        print(f"  {name}: {u.size} pixels ({nz} navigated): U {d[0]} V {d[1]} U_raw {d[2]} V_raw {d[3]} shorts differ, max |difference| {max(mx)} (cm/s)")
    print(f"total: {bad} of {tot} navigated-wind shorts differ ({bad / tot:.2e}); the raw-displacement shorts (short)(100 * uPix) have no product-sum and never differ")


if __name__ == "__main__":
    main()
