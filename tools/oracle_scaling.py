#!/usr/bin/env python3
"""How the OpenMP oracle's throughput depends on its thread count, and whether two oracle runs side by side on half the cores each beat one
after the other on all of them (the question behind the number of prefetch workers of tests/test_gpu_fullsize.py).  CPU only.
usage: oracle_scaling.py [n=2000] [kiters=4]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octane_amd import synth  # noqa: E402
from oracle import oct_oracle as oo  # noqa: E402  (the checker; this tool times it)


def run(a, b, prm, threads):
    oo.set_threads(threads)
    t = time.time()
    oo.flow(a, b, oo.FlowParams(**prm), flavour="omp", dot_threads=oo.REF_GRID_THREADS)
    return time.time() - t


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    prm = dict(kiters=int(sys.argv[2]) if len(sys.argv) > 2 else 4, liters=1, cgiters=30)
    oo.build()
    cores = oo.host_cpu_share()
    a, b = synth.lattice_scene(n, n, seed=5)
    a2, b2 = synth.lattice_scene(n, n, seed=6)
    run(a, b, dict(prm, cgiters=2), cores)                      # warm-up
    for th in sorted({cores, cores - 2, cores // 2, cores // 4} - {0}):
        print(f"{n}x{n} {prm}: {th:2d} threads {run(a, b, prm, th):6.1f} s", flush=True)
    t = time.time()
    run(a, b, prm, cores - 2); run(a2, b2, prm, cores - 2)
    seq = time.time() - t
    res = {}

    def work(key, x, y, th):
        res[key] = run(x, y, prm, th)
    half = (cores - 2) // 2
    ts = [threading.Thread(target=work, args=(0, a, b, half)), threading.Thread(target=work, args=(1, a2, b2, half))]
    t = time.time()
    for x in ts:
        x.start()
    for x in ts:
        x.join()
    par = time.time() - t
    print(f"two cases one after the other on {cores - 2} threads: {seq:.1f} s; side by side on {half} threads each: {par:.1f} s ({res[0]:.1f} / {res[1]:.1f})")


if __name__ == "__main__":
    main()
