#!/usr/bin/env python3
"""The arithmetic behind DESIGN 7's multi-GPU figures for the row-band solve of ONE frame (BASELINE configs[3]), round 5.

The pool gives one GPU per box, so an N-GPU pyramid cannot be timed.  Since round 5 the COMPUTE term is measured, not fitted:
tools/solo_band.py runs only band b's launch sequence of an N-band solve on the one GPU (profiles/r5_solo_band.txt).  This script

  1. rebuilds that term from kernel-level measurements -- per level too: the residual per PCG iteration of a small band (+10 ... +17 us
     at N = 8) is what a boundary's own event record + stream wait costs a band whose kernel is shorter than ~60 us, which stand-alone
     launch timings do not contain -- (the launch-time fit 10.7 us + 11.98 ps / pixel of the finest-level PCG kernel, the
     stored-q kernel's 80 B/pixel against 61.3, the persistent solves' us per iteration, assembly at 18 ps / pixel, updates, set-up) and
     compares it with the measured solo times -- the reconciliation VERDICT r4 asked for (within 5 %);
  2. adds what the solo measurement cannot show, each term named: per phase boundary the event-ordered wait for the slowest neighbour
     (measured between virtual bands and between processes on one GPU: ~15 us of stream idle time; the solo run already contains the
     host's share, an event record and a satisfied wait), and the peer copies over xGMI (bytes counted by the solo run; one link, 50 GB/s
     effective of its ~64 GB/s per direction);
  3. prints the PREDICTED N-GPU pyramid time, speed-up and efficiency, and the Amdahl bound set by what does not shrink with N: the
     replicated levels, the set-up, and the fixed cost of a launch + boundary on every banded PCG iteration.

usage: tiled_model.py [boundary_us=15] [xgmi_gbs=50]"""
import sys

BOUNDARY = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
XGMI = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
ITER = 270                                     # PCG iterations per level (3 GNC x 3 linearisations x 30)
LIN = 9                                        # linearisations (assemblies, flow updates) per level
THRESH = 4 << 20

# measured, profiles/r5_solo_band.txt: plain plan and slowest band per N (ms), peer-copy MB of the busiest band, phase boundaries;
# "lev": GPU ms per level of an inner band, FINEST first (N = 1: the whole frame through the band machinery)
SOLO = {
    10848: {"plain": 516.8, "levels": [10848, 5424, 2712, 1356, 678, 339, 170, 85],
            2: (281.4, 621.3, 870), 4: (155.6, 461.4, 870), 8: (99.8, 382.3, 870),
            "lev": {1: [388.58, 98.49, 26.64, 4.98, 2.41, 1.66, 1.28, 1.36], 2: [198.97, 53.84, 16.96, 5.02, 2.40, 1.65, 1.28, 1.36],
                    4: [102.36, 29.17, 12.51, 5.04, 2.41, 1.66, 1.28, 1.36], 8: [56.28, 19.20, 11.50, 5.01, 2.40, 1.66, 1.29, 1.37]}},
    5000: {"plain": 118.1, "levels": [5000, 2500, 1250, 625, 313, 157, 79, 40],
           2: (71.8, 128.0, 580), 4: (46.2, 92.0, 580), 8: (37.4, 74.0, 580),
           "lev": {1: [85.08, 23.56, 4.39, 1.78, 1.14, 0.96, 0.97, 0.78], 2: [46.24, 15.47, 4.37, 1.77, 1.14, 0.96, 0.97, 0.78],
                   4: [24.98, 11.09, 4.40, 1.80, 1.15, 0.96, 0.97, 0.76], 8: [16.68, 10.29, 4.43, 1.79, 1.14, 0.97, 0.97, 0.77]}},
}
# us per PCG iteration of a REPLICATED level (persistent / single-workgroup solves; profiles/r5_kernel_trace_summary.md and r3's 10848^2 run)
PERSIST_US = {1356: 17.0, 1250: 14.6, 678: 6.5, 625: 5.8, 339: 4.2, 313: 3.7, 170: 3.3, 157: 2.9, 85: 3.1, 79: 2.8, 40: 2.2}


def kernel_us(pixels, qform=True):
    return 10.7 + 11.98e-6 * (1.0 if qform else 80.0 / 61.3) * pixels


def band_iter_us(n, bands):
    band_px = n * n / bands
    if n == 10848 and bands == 1:
        return 1297.0
    return kernel_us(band_px, qform=band_px >= (2 << 20))


def level_ms(n, lv, bands):
    """one level of the solo term from kernel-level numbers (ms)"""
    return compute_ms(n, [lv], bands, full_res_reads=False)


def compute_ms(n, levels, bands, full_res_reads=True):
    """the solo term from kernel-level numbers: banded levels shrink with the band, everything else is replicated"""
    t = 0.0
    for lv in levels:
        px = lv * lv
        banded = bands > 1 and px >= THRESH
        share = 1.0 / bands if banded else 1.0
        if banded or px >= (2 << 20):
            t += ITER * band_iter_us(lv, bands if banded else 1)                  # one launch per iteration
        else:
            t += ITER * PERSIST_US.get(lv, 3.0)
        t += LIN * (5.0 + 18.2e-6 * px * share)                                   # k_assemble
        t += LIN * (4.0 + 7.7e-6 * px * share) if px >= (2 << 20) else 0.0        # k_flow_update_fused (inside the persistent solve otherwise)
        t += 6.0 + 5.5e-6 * px                                                     # level set-up: blur, decimation, gradients, up-sampling (replicated)
    if full_res_reads:
        t += 4 * 2.0e-6 * n * n                                                    # full-resolution reads of the sampled blur, four fields
    return t * 1e-3


def main():
    for n, rec in SOLO.items():
        lv = rec["levels"]
        t1 = rec["plain"]
        print(f"{n} x {n}, R1 parameters; plain plan measured {t1:.1f} ms, from kernel-level numbers {compute_ms(n, lv, 1):.1f} ms")
        fixed = None
        for b in (2, 4, 8):
            solo, mb, nbnd = rec[b]
            model = compute_ms(n, lv, b)
            wait = nbnd * BOUNDARY * 1e-3
            copies = mb / 1e3 / XGMI * 1e3
            t = solo + wait + copies
            print(f"  N={b}: solo band measured {solo:6.1f} ms (kernel-level model {model:6.1f}, {100 * (model / solo - 1):+.1f} %); + {nbnd} boundaries x {BOUNDARY:.0f} us = {wait:4.1f} ms"
                  f" + {mb:.0f} MB of peer copies at {XGMI:.0f} GB/s = {copies:4.1f} ms  ->  PREDICTED {t:6.1f} ms = {t1 / t:.2f} x, efficiency {t1 / t / b:.2f}"
                  f"   (compute only: {t1 / solo:.2f} x, {t1 / solo / b:.2f})")
            banded = [x for x in lv if x * x >= THRESH]
            print("        per banded level, measured | kernel-level model (ms): " +
                  "; ".join(f"{x}^2 {rec['lev'][b][i]:.1f} | {level_ms(n, x, b):.1f} ({(rec['lev'][b][i] - level_ms(n, x, b)) * 1e3 / ITER:+.0f} us per iteration)"
                            for i, x in enumerate(banded)))
        # Amdahl: what does not shrink with N (from the kernel-level model): replicated levels + set-up + per-iteration fixed costs of the banded levels
        nb_levels = sum(1 for x in lv if x * x >= THRESH)
        rep = compute_ms(n, [x for x in lv if x * x < THRESH], 1) + sum(6.0 + 5.5e-6 * x * x for x in lv if x * x >= THRESH) * 1e-3
        fixed = rep + nb_levels * ITER * (10.7 + BOUNDARY) * 1e-3
        par = t1 - rep - nb_levels * ITER * 10.7e-3
        print(f"  does not shrink with N: replicated levels and set-up {rep:.1f} ms + {nb_levels} banded levels x {ITER} iterations x (10.7 us launch + {BOUNDARY:.0f} us boundary) = "
              f"{fixed:.1f} ms; shrinks: {par:.1f} ms  ->  bound {t1 / (fixed + par / 4):.2f} x on 4 GPUs ({t1 / (fixed + par / 4) / 4:.2f}), "
              f"{t1 / (fixed + par / 8):.2f} x on 8 ({t1 / (fixed + par / 8) / 8:.2f})")


if __name__ == "__main__":
    main()
