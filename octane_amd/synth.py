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
* ``disc_scene``      -- "S6" (round 5): what the calibration step really hands the
  solver from a full-disk file (ref src/oct_navcal_cuda.cu:29-93): an Earth disc
  on a background of exact zeros, the limb taper between subpoint distances
  0.021 and 0.0212 rad^2, radiances that went through int16 counts (plateaus of
  equal values -> exactly-zero gradients), sensor noise, a saturated patch that
  moves with the flow.
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


def _lattice_octaves(rng, nchan: int):
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
    return octaves


def lattice_scene(nx: int, ny: int, seed: int = 20240613, nchan: int = 1, device=None):
    """S2-style pair.  With ``device`` given (a torch device) the pair is built
    with torch on that device and returned as torch tensors [nchan, ny, nx];
    otherwise numpy arrays.  Same formula either way (libm/ocml ulp drift aside)."""
    rng = np.random.RandomState(seed)
    octaves = _lattice_octaves(rng, nchan)
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

# ABI full disk on the GOES-R fixed grid: 10848 columns of 28 urad (ref src/oct_fileread.cc reads x / y as shorts with scale 2.8e-5 at
# 1 km), so the frame's half width is 0.151872 rad; band 13's count scale and the 0..255 normalisation range of
# ref src/oct_normalize_geo.cc:71-74.
DISC_HALF_WIDTH = 10848 * 2.8e-5 / 2
_RAD_SCALE, _RAD_OFFSET = 0.04572, -1.5726
_MININ, _MAXIN = -1.6443, 185.5699


def disc_mask(nx: int, ny: int, centre=(0.5, 0.5), span: float = 1.0):
    """sdsconst of ref src/oct_navcal_cuda.cu:31-33,81-91 for a frame whose nx columns cover `span` of the full disk's width, the
    sub-satellite point at (centre[0] * nx, centre[1] * ny): 1 inside, 0 beyond the limb threshold, the linear taper between (float32,
    as the reference's)."""
    dp = np.float32(2 * DISC_HALF_WIDTH * span / nx)
    xs = (np.arange(nx, dtype=np.float32) * dp + np.float32(-centre[0] * nx * float(dp))).astype(np.float64)
    ys = (np.arange(ny, dtype=np.float32) * -dp + np.float32(centre[1] * ny * float(dp))).astype(np.float64)
    dist = xs[None, :] ** 2 + ys[:, None] ** 2
    slope = np.float32(1. / (0.021 - 0.0212))
    icpt = np.float32(1. - 0.021 * float(slope))
    taper = (np.float64(slope) * dist + np.float64(icpt)).astype(np.float32)
    return np.where(dist < 0.021, np.float32(1), np.where(dist >= 0.0212, np.float32(0), taper)).astype(np.float32)


def disc_scene(nx: int, ny: int, seed: int = 20240613, nchan: int = 1, centre=(0.5, 0.5), span: float = 1.0,
               noise: float = 0.6, saturate: bool = True, device=None):
    """S6 pair: the lattice clouds of ``lattice_scene`` seen the way a GOES-R full-disk file hands them to the solver.  Per channel:
    radiance = 12-bit counts (mid-grey + lattice texture under a smooth envelope that leaves near-uniform "clear" regions + a hot
    blob that saturates at 4095 when ``saturate``) + ``noise`` counts of Gaussian sensor noise, rounded to int16 and clipped;
    image = sdsconst * normalise(counts * scale + offset) exactly as ref src/oct_navcal_cuda.cu:34,93 forms it (float32 count
    arithmetic, double normalisation, float32 store).  Clouds, envelope and blob move with ``true_lattice_flow``; the disc does not
    (geostationary).  Outside the disc both images are exactly 0.  Returns float32 [nchan, ny, nx] (torch tensors on ``device`` when
    given; the noise always comes from numpy's seeded generator so both forms see the same counts up to libm / ocml ulps)."""
    rng = np.random.RandomState(seed)
    octaves = _lattice_octaves(rng, nchan)
    mask = disc_mask(nx, ny, centre, span)
    nz = [[(noise * rng.randn(ny, nx)).astype(np.float32) if noise > 0 else None for _ in range(2)] for _ in range(nchan)]
    if device is not None:
        import torch
        xp = torch
        j, i = torch.meshgrid(torch.arange(ny, dtype=torch.float64, device=device),
                              torch.arange(nx, dtype=torch.float64, device=device), indexing="ij")
        u = 2.5 + 1.5 * torch.sin(2 * math.pi * j / ny)
        v = -1.0 + 1.0 * torch.cos(2 * math.pi * i / nx)
        mask_x = torch.from_numpy(mask).to(device)
        f32 = lambda t: t.to(torch.float32)
        f64 = lambda t: t.to(torch.float64)
        rint, clip = torch.round, torch.clamp
    else:
        xp = np
        j, i = np.meshgrid(np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
        u, v = true_lattice_flow(nx, ny)
        mask_x = mask
        f32 = lambda t: t.astype(np.float32)
        f64 = lambda t: t.astype(np.float64)
        rint, clip = np.rint, np.clip
    amp = sum(o[0] for o in octaves[0])
    bx, by, bs = 0.62 * nx, 0.41 * ny, 0.035 * max(nx, ny)              # the hot blob
    ex, ey = 2 * math.pi / (0.9 * nx), 2 * math.pi / (0.7 * ny)          # the envelope: ~one period over the frame

    def counts(per, x, y, n):
        acc = None
        for a, kx1, ky1, p1, kx2, ky2, p2 in per:
            t = a * xp.cos(kx1 * x + ky1 * y + p1) * xp.cos(kx2 * x - ky2 * y + p2)
            acc = t if acc is None else acc + t
        env = 0.02 + 0.98 * (0.5 + 0.5 * xp.cos(ex * x + 0.3) * xp.cos(ey * y - 0.8)) ** 2
        c = 1900.0 + 1600.0 * (acc / amp) * env
        if saturate:
            c = c + 3000.0 * xp.exp(-((x - bx) ** 2 + (y - by) ** 2) / (2 * bs * bs))
        if n is not None:
            c = c + (f64(torch.from_numpy(n).to(device)) if device is not None else n)
        return clip(rint(c), 0, 4095)

    def calibrated(cnt):
        dval = f32(cnt) * np.float32(_RAD_SCALE) + np.float32(_RAD_OFFSET)      # float dVal = data2 * radScale + radOffset
        minin, maxin = float(np.float32(_MININ)), float(np.float32(_MAXIN))
        val = ((f64(dval) - minin) / (maxin - minin)) * (255.0 - 0.0) + 0.0
        return f32(f64(mask_x) * val)

    im1 = [calibrated(counts(per, i, j, nz[c][0])) for c, per in enumerate(octaves)]
    im2 = [calibrated(counts(per, i - u, j - v, nz[c][1])) for c, per in enumerate(octaves)]
    if device is not None:
        return xp.stack(im1).contiguous(), xp.stack(im2).contiguous()
    return np.stack(im1), np.stack(im2)


def interior_mean(f: np.ndarray, frac: float = 0.125):
    ny, nx = f.shape[-2:]
    my, mx = int(ny * frac), int(nx * frac)
    return float(np.asarray(f)[..., my:ny - my, mx:nx - mx].mean())
