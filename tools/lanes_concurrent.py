"""Round 5: two lanes of 2000 x 2000 pairs (BASELINE configs[4]'s shape, kiters 6) with their persistent mid-level
solves (i) serialised per device and uncapped -- the product's arrangement --, (ii) launched CONCURRENTLY, each capped at half the CUs,
(iii) concurrently and uncapped, (iv) serialised and capped at half.  Same pairs, flows compared bit for bit with arrangement (i).
   python tools/lanes_concurrent.py [n=2000] [kiters=6] [pairs per lane=8]"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from octane_amd import capi, synth  # noqa: E402
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lanes = 2
dev = torch.device("cuda:0")
pairs = [synth.lattice_scene(n, n, seed=3 + i, device=dev) for i in range(lanes)]
prm = capi.FlowParams(kiters=kit)
expect = kit * 3 * 3 * 30
ref = None
for name, chain, maxg in (("serialised, uncapped (product)", 1, 0), ("concurrent, capped at 128", 0, 128), ("concurrent, uncapped", 0, 0), ("serialised, capped at 128", 1, 128)):
    plans = [capi.Plan(n, n, 1, prm) for _ in range(lanes)]
    for pl in plans:
        pl.tune("persist_chain", chain)
        if maxg:
            pl.tune("persist_max_g", maxg)
    outs = [(torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)) for _ in range(lanes)]
    torch.cuda.synchronize()

    def work(ln, count):
        a, b = pairs[ln]
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
    best = min(run(reps) for _ in range(3))
    its = [pl.last_iterations() for pl in plans]
    flows = [(o[0].cpu(), o[1].cpu()) for o in outs]
    if ref is None:
        ref = flows
    same = all(torch.equal(f[0], r[0]) and torch.equal(f[1], r[1]) for f, r in zip(flows, ref))
    print(f"{n}x{n} kiters={kit}, two lanes, persistent solves {name:32s}: {lanes * reps * n * n / best / 1e6:7.1f} Mpix/s "
          f"({best / reps * 1e3:6.2f} ms per round of two pairs; iterations {its}, expected {expect}; flows == arrangement (i): {same})", flush=True)
    for pl in plans:
        pl.close()
