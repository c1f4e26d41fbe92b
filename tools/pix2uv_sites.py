#!/usr/bin/env python3
"""Round 5: which multiply-add sites of the navigation decide the `short` winds?

nvcc builds the reference's kernel with -fmad=true (its default: ref src/Makefile:9,20,27 pass no -fmad=false), so every `a * b + c` of
ref src/oct_pix2uv_cuda.cu:13-25 (haversine), :40-44 / :99-103 (base and displaced position) and :105-118 (fixed-grid projection) MAY be a
single fused operation in the reference's binary.  oracle/pix2uv_oracle.c carries each of the 13 sites as an explicit fma() switch; this
script counts, on the navigation cases of tools/pix2uv_fmad_exposure.py (four fixed-grid / polar / mercator cases + the 5000 x 5000
frame), how many shorts each site moves ALONE, what the two float sites do together, what the eleven double sites do together, all
thirteen, and what gcc's own contraction (the oracle's "fma" flavour) gives -- so that "the reference CUDA path" reduces to a small
number of candidate outputs.  CPU only (the oracle is the subject).  Output: profiles/r5_pix2uv_sites.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from oracle import oct_oracle as oo
from pix2uv_fmad_exposure import cases


def shorts(n, t1, t2, u, v, mode, flavour="strict", sites=0):
    return oo.pix2uv(n, t1, t2, u, v, 0, mode, flavour=flavour, sites=sites)[:2]


def diff(a, b):
    return int((a[0] != b[0]).sum() + (a[1] != b[1]).sum()), int(max(np.abs(a[0].astype(np.int32) - b[0]).max(), np.abs(a[1].astype(np.int32) - b[1]).max()))


def main():
    oo.build()
    ns = len(oo.P2U_SITES)
    dbl = oo.P2U_ALL_SITES & ~oo.P2U_FLOAT_SITES
    print("pix2uv: navigated-wind shorts (U, V) that change against the strict two-rounding forms when a multiply-add site is ONE fused operation")
    print("sites: " + "; ".join(f"{k} = {nm}" for k, nm in enumerate(oo.P2U_SITES)))
    tot = {}
    for name, n, u, v, mode, t1, t2 in cases():
        base = shorts(n, t1, t2, u, v, mode)
        nav = int((base[0] != 0).sum())
        row = {}
        for k in range(ns):
            row[f"site {k}"] = diff(shorts(n, t1, t2, u, v, mode, sites=1 << k), base)
        fl = shorts(n, t1, t2, u, v, mode, sites=oo.P2U_FLOAT_SITES)
        al = shorts(n, t1, t2, u, v, mode, sites=oo.P2U_ALL_SITES)
        gc = shorts(n, t1, t2, u, v, mode, flavour="fma")
        row["float sites (0+1)"] = diff(fl, base)
        row["double sites (2..12)"] = diff(shorts(n, t1, t2, u, v, mode, sites=dbl), base)
        row["all 13"] = diff(al, base)
        row["gcc -ffp-contract=fast"] = diff(gc, base)
        row["all 13 vs float sites only"] = diff(al, fl)
        row["gcc contraction vs float sites only"] = diff(gc, fl)
        print(f"  {name}: {2 * u.size} shorts ({nav} pixels navigated)")
        for key, (cnt, mx) in row.items():
            if cnt or not key.startswith("site "):
                print(f"      {key:40s} {cnt:9d} differ, max |difference| {mx} cm/s")
            t = tot.setdefault(key, [0, 0]); t[0] += cnt; t[1] = max(t[1], mx)
        quiet = [k for k in range(ns) if row[f"site {k}"][0] == 0]
        print(f"      sites that move nothing alone: {quiet}")
    print("total over the cases:")
    for key, (cnt, mx) in tot.items():
        print(f"      {key:40s} {cnt:9d} differ, max |difference| {mx} cm/s")


if __name__ == "__main__":
    main()
