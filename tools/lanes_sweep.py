"""Throughput of L concurrent lanes (plans on their own streams, one host thread each) on pairs of n x n, kiters levels:
what overlapping one pair's latency-bound coarse levels with another's bandwidth-bound fine levels is worth at a given
frame size.   usage: python tools/lanes_sweep.py n kiters lanes [pairs per lane = 3]"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from octane_amd import capi, synth  # noqa: E402

n, kit, lanes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
a, b = synth.lattice_scene(n, n, seed=3, device=dev)
prm = capi.FlowParams(kiters=kit)
plans = [capi.Plan(n, n, 1, prm) for _ in range(lanes)]
outs = [(torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)) for _ in range(lanes)]
torch.cuda.synchronize()


def work(ln, count):
    u, v = outs[ln]
    for _ in range(count):
        plans[ln].solve_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), 0, 0, capi.STREAM_OWN)
    plans[ln].wait()


def run(count):
    th = [threading.Thread(target=work, args=(ln, count)) for ln in range(lanes)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


run(1)
dt = run(reps)
print(f"{n}x{n} kiters={kit} lanes={lanes}: {lanes * reps * n * n / dt / 1e6:.1f} Mpix/s ({dt / reps * 1e3:.1f} ms per round of {lanes})", flush=True)
