"""Synthetic image pairs shaped like the solver's real inputs.

There is no GOES data in the container and no network, so every configuration
in BASELINE.json runs on shape-matched synthetic pairs.  Values
are float32 in the 0..255 range the reference's calibration step produces
(ref src/oct_navcal_cuda.cu:93).

* ``gaussian_scene``  -- "S1": five Gaussian blobs, second image translated by a
  constant shift.  Formula from SURVEY.md 8(d); used for the 512x512 config and
  the recorded reference answers in BASELINE.md 2.
* ``lattice_scene``   -- "S2..S5": six octaves of cosine lattices with seeded
  phases, second image advected by a smooth non-constant displacement field.
  The scene is analytic, so the second image is evaluated exactly at the
  displaced coordinates (no resampling step).
"""
from __future__ import annotations

import math

import numpy as np

_CX = (.3, .7, .5, .2, .8)
_CY = (.3, .6, .8, .7, .2)
_SG = (.08, .12, .06, .10, .09)


def gaussian_scene(n: int, shift=(3.0, -2.0), ny: int | None = None):
    """S1 pair. Returns (img1, img2) float32 [ny, nx]; true flow == shift."""
    nx = n
    ny = n if ny is None else ny
    j, i = np.meshgrid(np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")

    def scene(x, y):
        acc = np.zeros_like(x)
        for cx, cy, s in zip(_CX, _CY, _SG):
            acc += np.exp(-(((x / nx) - cx) ** 2 + ((y / ny) - cy) ** 2) / (2 * s * s))
        return 127.5 * acc

    a = scene(i, j)
    b = scene(i - shift[0], j - shift[1])
    return a.astype(np.float32), b.astype(np.float32)


def true_lattice_flow(nx: int, ny: int, xp=np):
    """The displacement field lattice_scene advects by (u along x, v along y)."""
    j, i = xp.meshgrid(xp.arange(ny, dtype=xp.float64), xp.arange(nx, dtype=xp.float64), indexing="ij")
    u = 2.5 + 1.5 * xp.sin(2 * math.pi * j / ny)
    v = -1.0 + 1.0 * xp.cos(2 * math.pi * i / nx)
    return u, v


def lattice_scene(nx: int, ny: int, seed: int = 20240613, nchan: int = 1, device=None):
    """S2-style pair.  With ``device`` given (a torch device) the pair is built
    with torch on that device and returned as torch tensors [nchan, ny, nx];
    otherwise numpy arrays.  Same formula either way (libm/ocml ulp drift aside)."""
    rng = np.random.RandomState(seed)
    octaves = []
    for c in range(nchan):
        per = []
        for o in range(6):
            wavelength = 256.0 / (2 ** o)
            th = rng.uniform(0, math.pi)
            k = 2 * math.pi / wavelength
            per.append((1.0 / (o + 1), k * math.cos(th), k * math.sin(th), rng.uniform(0, 2 * math.pi),
                        k * math.sin(th + 0.7), k * math.cos(th + 0.7), rng.uniform(0, 2 * math.pi)))
        octaves.append(per)
    if device is not None:
        import torch
        xp = torch
        j, i = torch.meshgrid(torch.arange(ny, dtype=torch.float64, device=device),
                              torch.arange(nx, dtype=torch.float64, device=device), indexing="ij")
        u = 2.5 + 1.5 * torch.sin(2 * math.pi * j / ny)
        v = -1.0 + 1.0 * torch.cos(2 * math.pi * i / nx)
    else:
        xp = np
        j, i = np.meshgrid(np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
        u, v = true_lattice_flow(nx, ny)

    amp = sum(o[0] for o in octaves[0])

    def scene(per, x, y):
        acc = None
        for a, kx1, ky1, p1, kx2, ky2, p2 in per:
            t = a * xp.cos(kx1 * x + ky1 * y + p1) * xp.cos(kx2 * x - ky2 * y + p2)
            acc = t if acc is None else acc + t
        return (acc / amp * 0.5 + 0.5) * 255.0

    im1 = [scene(per, i, j) for per in octaves]
    im2 = [scene(per, i - u, j - v) for per in octaves]
    if device is not None:
        import torch
        return (torch.stack(im1).to(torch.float32).contiguous(), torch.stack(im2).to(torch.float32).contiguous())
    return np.stack(im1).astype(np.float32), np.stack(im2).astype(np.float32)


def interior_mean(f: np.ndarray, frac: float = 0.125):
    ny, nx = f.shape[-2:]
    my, mx = int(ny * frac), int(nx * frac)
    return float(np.asarray(f)[..., my:ny - my, mx:nx - mx].mean())
