"""Two plans on two streams (the lanes of the batch workload) against one plan run twice: wall time, and -- from a
rocprofv3 kernel trace of this script -- how much the two lanes' kernels overlap (tools/lanes_overlap.py).
usage: python tools/lanes_probe.py [n=2000] [kiters=6] [pairs per lane=2] [threads=0|1]"""
import sys
import threading
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from octane_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
use_threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
a, b = synth.lattice_scene(n, n, seed=3, device=dev)
prm = capi.FlowParams(kiters=kit)
plans = [capi.Plan(n, n, 1, prm) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = [(torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)) for _ in range(2)]
torch.cuda.synchronize()


def lane(i, count):
    for _ in range(count):
        with torch.cuda.stream(streams[i]):
            outs[i][0].zero_(); outs[i][1].zero_()
            plans[i].run_device(a.data_ptr(), b.data_ptr(), outs[i][0].data_ptr(), outs[i][1].data_ptr(), streams[i].cuda_stream)


lane(0, 1); lane(1, 1); torch.cuda.synchronize()
t0 = time.perf_counter(); lane(0, 2 * reps); torch.cuda.synchronize(); t_one = time.perf_counter() - t0
t0 = time.perf_counter()
if use_threads:
    th = [threading.Thread(target=lane, args=(i, reps)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
else:
    for _ in range(reps):
        lane(0, 1); lane(1, 1)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); t_two = time.perf_counter() - t0
print(f"{2 * reps} pyramids of {n}x{n} kiters={kit}: one lane {t_one * 1e3:.1f} ms, two lanes {t_two * 1e3:.1f} ms "
      f"(host issue {t_issue * 1e3:.1f} ms, {'two host threads' if use_threads else 'one host thread'})")
