"""One rank of the multi-process row-band test (tests/test_gpu_tiled_mp.py starts `world` of these; rank r runs on GPU
r mod (visible GPUs): all share GPU 0 on a one-GPU box, real peers over HIP IPC wherever there are more).  Rendezvous and the IPC-handle all-gather go through torch.distributed (gloo, 127.0.0.1)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octane_amd import capi, synth  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nx, ny, kit, lit, cg, minpix = (int(x) for x in sys.argv[1:7])
    hint = len(sys.argv) > 7 and sys.argv[7] == "hint"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)

    def all_gather(blob):
        out = [None] * world
        dist.all_gather_object(out, blob)
        return out

    a, b = synth.lattice_scene(nx, ny, seed=41)
    u0 = v0 = None
    dev = capi.band_devices(world)[rank]
    prm = capi.FlowParams(kiters=kit, liters=lit, cgiters=cg, device=dev)
    if hint:
        rng = np.random.RandomState(4)
        u0 = (2.0 + 0.2 * rng.randn(ny, nx)).astype(np.float32)
        v0 = (-1.0 + 0.2 * rng.randn(ny, nx)).astype(np.float32)
        prm = capi.FlowParams(kiters=kit, liters=lit, cgiters=cg, lambdac=0.4, device=dev)
    # "xchg": register torch.distributed (gloo here: host-staged) as the library's collective transport; which transport then RUNS
    # is the self-check's choice, or OCTANE_TILED_TRANSPORT's
    exchange = None
    if os.environ.get("OCTANE_TEST_EXCHANGE") == "1":
        from octane_amd import exchange as xch
        torch.cuda.set_device(dev)
        exchange = xch.TorchExchange(torch.device("cuda", dev))
    try:
        mp = capi.MpPlan(nx, ny, 1, prm, rank, world, f"/octane_test_{os.environ['MASTER_PORT']}", all_gather, min_band_pixels=minpix,
                         exchange=exchange)
    except capi.OctaneError as e:
        print(f"MP_CREATE_FAILED rank={rank} msg={str(e)!r}", flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3)
    banded = mp.banded_levels
    if len(sys.argv) > 7 and sys.argv[7] == "loop":
        # dead-peer drill: solve again and again until the library reports that the group is dead (the test kills another
        # rank meanwhile); the survivor must get an error -- not hang --, a second call must fail at once, close must return
        import time
        print("MP_LOOP_RUNNING", flush=True)
        t_err = None
        try:
            for _ in range(100000):
                t_call = time.perf_counter()
                mp.run_host(a, b, u0, v0)
        except capi.OctaneError as e:
            t_err = time.perf_counter() - t_call
            msg = str(e)
        t2 = time.perf_counter()
        second_failed = False
        try:
            mp.run_host(a, b, u0, v0)
        except capi.OctaneError:
            second_failed = True
        t2 = time.perf_counter() - t2
        t3 = time.perf_counter()
        mp.close()
        t3 = time.perf_counter() - t3
        print(f"MP_DEAD_RESULT error_after={t_err:.2f}s second_call_failed={second_failed} in {t2:.3f}s close={t3:.3f}s msg={msg!r}", flush=True)
        os._exit(0 if (t_err is not None and second_failed and t2 < 1.0 and t3 < 5.0) else 1)     # no gloo teardown with a dead peer
    if len(sys.argv) > 7 and sys.argv[7] == "fault":
        # abandoned-solve drill: the persistent solve of a replicated level gives up on every rank (test hook); every
        # rank's octane_vof_mp_run has to return an error -- not success with an invalid flow --, last_iterations() is -2, and the
        # next run, hook off, is good again
        tune = capi.Plan(64, 64, 1, capi.FlowParams(kiters=1, device=dev))
        u_ok, v_ok = mp.run_host(a, b, u0, v0)
        its_ok = mp.last_iterations()
        tune.tune("persist_fault", 1)
        failed = False
        try:
            mp.run_host(a, b, u0, v0)
        except capi.OctaneError as e:
            failed = True
            msg = str(e)
        its_bad = mp.last_iterations()
        tune.tune("persist_fault", 0)
        u2, v2 = mp.run_host(a, b, u0, v0)
        same = True if rank != 0 else bool(np.array_equal(u2, u_ok) and np.array_equal(v2, v_ok))
        ok = failed and its_bad == -2 and mp.last_iterations() == its_ok and same
        print(f"MP_FAULT_RESULT rank={rank} failed={failed} its_bad={its_bad} its={mp.last_iterations()}/{its_ok} same={same} ok={ok}"
              + (f" msg={msg!r}" if failed else ""), flush=True)
        tune.close(); mp.close()
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    for rep in range(2):                                   # twice: the protocol must be re-enterable
        u, v = mp.run_host(a, b, u0, v0)
    ok = True
    if rank == 0:
        pl = capi.Plan(nx, ny, 1, prm)                          # the plain plan on this rank's device
        up, vp = pl.run_host(a, b, u0, v0)
        ip = pl.last_iterations()
        pl.close()
        d = float(np.sqrt((((u - up).astype(np.float64)) ** 2 + ((v - vp).astype(np.float64)) ** 2).sum() /
                          ((up.astype(np.float64)) ** 2 + (vp.astype(np.float64)) ** 2).sum()))
        ok = np.isfinite(u).all() and d < 2e-5 and mp.last_iterations() == ip
        import json
        import zlib
        d_or = -1.0
        if nx * ny <= 400000:                                   # the oracle too, where it takes a moment
            from oracle import oct_oracle as oo
            uo, vo, _ = oo.flow(a, b, oo.FlowParams(kiters=kit, liters=lit, cgiters=cg, lambdac=prm.lambdac), u0=u0, v0=v0,
                                 flavour="omp", dot_threads=oo.REF_GRID_THREADS)
            d_or = float(np.sqrt((((u - uo).astype(np.float64)) ** 2 + ((v - vo).astype(np.float64)) ** 2).sum() /
                                 ((uo.astype(np.float64)) ** 2 + (vo.astype(np.float64)) ** 2).sum()))
            ok = ok and d_or < 2e-5
        info = mp.transport_info()
        if exchange is not None:
            info["exchange_calls"] = exchange.calls
        print(f"MP_RESULT devices={capi.band_devices(world)} banded={banded} relL2={d:.3e} oracle={d_or:.3e} its={mp.last_iterations()}/{ip} ok={ok} "
              f"crc={zlib.crc32(u.tobytes() + v.tobytes()):08x} info={json.dumps(info)}", flush=True)
    mp.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
