#!/usr/bin/env python3
"""The arithmetic behind DESIGN 7's predicted multi-GPU efficiencies of the row-band solve (PREDICTED: the pool gives one GPU per box).
Inputs are measurements of round 3 / 4 on one MI355X: the launch-time fit of the finest-level PCG kernel (10.7 us + 11.98 ps per pixel; the
10848^2 level itself: 1297 us), the stored-q kernel's 80 B/pixel against 61.3, the persistent solves' us per iteration, ~15 us per phase
boundary (event-ordered, measured between virtual bands and between processes), and a 10848^2 pyramid's 519 ms on one GPU.
usage: tiled_model.py [threshold_pixels=4194304] [boundary_us=15]"""
import sys

THRESH = int(sys.argv[1]) if len(sys.argv) > 1 else 4 << 20
BOUNDARY = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
N = 10848
ITER = 270                                     # PCG iterations per level (3 GNC x 3 linearisations x 30)
levels = [(N >> k) if k else N for k in range(8)]          # 10848, 5424, 2712, 1356, 678, 339, 170 (169.5 -> 170), 85
levels = [10848, 5424, 2712, 1356, 678, 339, 170, 85]
PERSIST_US = {1356: 17.0, 678: 6.5, 339: 4.2, 170: 3.3, 85: 3.1}      # replicated: one persistent launch per solve


def kernel_us(pixels, qform=True):
    per_px = 11.98e-6 * (1.0 if qform else 80.0 / 61.3)
    return 10.7 + per_px * pixels


def level_iter_us(n, bands):
    px = n * n
    if px == N * N and bands == 1:
        return 1297.0                          # measured at 10848^2 (the fit over-predicts there)
    if bands == 1 or px < THRESH:
        return PERSIST_US.get(n, kernel_us(px))            # replicated: the whole level on every device
    band_px = px / bands
    base = 1297.0 / bands + 10.7 * (1 - 1 / bands) if px == N * N else kernel_us(band_px, qform=band_px >= (2 << 20))
    return base + BOUNDARY


def pyramid_ms(bands):
    pcg = ITER * sum(level_iter_us(n, bands) for n in levels) * 1e-3
    # assembly (9 launches per level, 88 B/pixel at ~18 ps per pixel), flow updates, level set-up: 35 ms on one GPU; the banded
    # levels' share divides by the band count, the set-up (5 ms) and the replicated levels' share do not
    banded_share = sum(n * n for n in levels if bands > 1 and n * n >= THRESH) / sum(n * n for n in levels)
    rest = 5.0 + 30.0 * ((1 - banded_share) + banded_share / bands)
    return pcg + rest


def main():
    t1 = pyramid_ms(1)
    print(f"threshold {THRESH} pixels, boundary {BOUNDARY} us; one GPU: {t1:.0f} ms (measured 519)")
    for b in (2, 4, 8):
        t = pyramid_ms(b)
        print(f"{b} GPUs: {t:.0f} ms = {t1 / t:.2f} x, efficiency {t1 / t / b:.2f}")


if __name__ == "__main__":
    main()
