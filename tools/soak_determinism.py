#!/usr/bin/env python3
"""Run-to-run determinism at full size: N consecutive solves of the headline configuration (5000^2, kiters 8, liters 3, cgiters 30) on one plan, the
flow hashed on the device after each; every hash has to equal the first (the reductions are fixed-order fp64 folds; the reference's float atomicAdd
reductions, ref .cu:151-186, are order-dependent).  Also reports device memory before / after (no growth) and the spread of the step times.
usage: soak_determinism.py [n=5000] [solves=100]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from octane_amd import capi, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    solves = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    a, b = synth.lattice_scene(n, n, seed=20240615, device="cuda")
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=8, liters=3, cgiters=30))
    s = torch.cuda.current_stream().cuda_stream
    free0 = torch.cuda.mem_get_info()[0]
    hashes, times = [], []
    for i in range(solves):
        u.zero_(); v.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        its = pl.last_iterations()
        assert its > 0 and (i == 0 or its == its0), (its, i)
        its0 = its
        # an order-sensitive device-side hash of the bit patterns (int64 wrap-around arithmetic)
        w = torch.arange(1, n * n + 1, device="cuda", dtype=torch.int64)
        h = int((u.view(torch.int32).flatten().to(torch.int64) * w).sum() ^ ((v.view(torch.int32).flatten().to(torch.int64) * (w + 7)).sum() << 1))
        del w
        hashes.append(h)
    free1 = torch.cuda.mem_get_info()[0]
    same = sum(1 for h in hashes if h == hashes[0])
    times_ms = sorted(t * 1e3 for t in times[1:])
    print(f"{solves} solves of {n}x{n} (kiters 8, liters 3, cgiters 30; {its0} PCG iterations each): {same} / {solves} flows bit-identical to the first (hash {hashes[0] & 0xffffffffffff:012x}); "
          f"step time min / median / max {times_ms[0]:.2f} / {times_ms[len(times_ms) // 2]:.2f} / {times_ms[-1]:.2f} ms; "
          f"free device memory {free0 / 2**30:.2f} -> {free1 / 2**30:.2f} GiB")
    pl.close()
    sys.exit(0 if same == solves else 1)


if __name__ == "__main__":
    main()
