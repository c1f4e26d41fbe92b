#!/usr/bin/env python3
"""bench.py -- Mpix/s of the full coarse-to-fine solve (BASELINE.json's metric).

A "step" is one whole oct_variational_optical_flow call on one image pair: pyramid build, and per
level 3 GNC x liters assemblies, each followed by cgiters PCG iterations -- with the pair already
resident in HBM when the timed region starts.  Workload at every N: BASELINE.json configs[2]
("R1"): 5000x5000, kiters=8, liters=3, cgiters=30, one channel, synthetic lattice
scene.  With N ranks each rank solves its own independent pair on its own GPU (the path shards
over pairs: no collective in the data path; weak scaling); value = N * pixels / max-over-ranks
time.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (python -m torch.distributed.run,
fresh child processes, before this process has touched the GPU) and exits with their code; under torchrun (the
driver's launch) the ranks come from the environment and --gpus has to agree with WORLD_SIZE.

The JSON line also carries
  value_drop_in -- SURVEY 8d's primary metric, the drop-in call: one oct_variational_optical_flow-shaped call on the caller's pageable
                   HOST buffers without a first guess (H2D + all levels + D2H through octane_vof_solve); never `value`;
  value_with_transfers -- the same call in its other forms (with a first guess, pinned buffers);
  secondary_multi_gpu -- N > 1 only: after the pair headline rank 0 measures BASELINE.json configs[3] (one 10848^2 frame, one row band
                   per rank, with its transport block and parity against the plain plan) and configs[4] (64 pairs of 2000^2 sharded)
                   as bounded, non-fatal side legs -- CHILD jobs of N fresh ranks each, so that not even a GPU fault in a leg can
                   cost the headline;
  placement_trials -- min / median / max over the candidate arenas the plan timed when it was created (the headline is a
                   best-of-n-placements figure, EXPERIMENTS.md 8).  A trial is a few PCG launches with the stop test held open
                   (varying weights, x work every second launch: 64 B/pixel), each timed by an event pair;
  roofline      -- dominant kernel (the fused, q-recomputing PCG iteration at the finest level, k_pcg_fused_q_dma): its
                   algorithmic bytes per launch (r p a1 a2 a4 wx wy read, r p written = 52 B/pixel; every second launch
                   + x and the p before last read, x written = 76; mean 64, and 8 less in the first GNC step, whose
                   weights are the constant -1: 61.33 B/pixel over a pyramid's finest-level launches; 80 / 72 for the
                   stored-q kernel of smaller levels) / its mean duration from HIP events on the launch stream.  SURVEY
                   8d's 116 B/pixel (two passes, seven coefficient planes, q stored) does not describe this kernel --
                   the line carries that pricing too, as frac_at_survey_bytes;
  cpu_baseline  -- the CPU oracle ("port", OpenMP over the host cores) timed on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The batch workload keeps four streams busy; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by
# default) and two streams on one queue serialise.  Must be in the environment before the runtime initialises, i.e.
# before torch touches the GPU (the library asks for the same when it is loaded first).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PASS_A_BYTES_PER_PIXEL = 36 + 16   # reads r(2) p(2) a1 a2 a4 wx wy, writes p(2) q(2)  -- DESIGN.md
PASS_A_BYTES_PER_PIXEL_GNC0 = 28 + 16   # first GNC step (a third of the launches): wx == wy == -1, the planes are not read
PASS_B_BYTES_PER_PIXEL = 40 + 16   # reads x(2) r(2) p(2) q(2) a1 a4, writes x(2) r(2)
# one fused kernel per iteration: reads r q p (24) + a1 a2 a4 wx wy (20), writes r p q (24) = 68 B/pixel; x is updated by every second
# launch only, which then also reads x and the p before last (16) and writes x (8): 92.  Mean 80.
FUSED_BYTES_PER_PIXEL = 80
FUSED_BYTES_PER_PIXEL_GNC0 = 72        # first GNC step: wx == wy == -1, not read
FUSED_Q_BYTES_PER_PIXEL = 64           # k_pcg_fused_q_dma / k_pcg_fused_q (levels of >= 2 * 2^20 pixels): q = A p is formed again, neither written nor read
FUSED_Q_BYTES_PER_PIXEL_GNC0 = 56
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
TILED_SEED = 20240615              # (with 20240616 the 85 x 85 coarsest level of a 10848^2 / 8-level pyramid runs away at R1's iteration counts)
TILED_PARITY_BAR = 2e-5            # banded vs plain solve of the same frame: two groupings of the same fp64 partial sums


def device_state(torch, dev):
    """What this box's memory system delivers right now: boxes of this pool come in a fast and a slow state (the same binary's
    finest-level PCG launch takes 0.33 or 0.42 ms), so every record carries a plain 1 GiB device copy timed in the same process."""
    import time as _t
    prop = torch.cuda.get_device_properties(dev)
    n = 1 << 28                                       # 1 GiB of floats read + 1 GiB written per copy
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    a.fill_(1.0)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    t0 = _t.perf_counter()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    dt = (_t.perf_counter() - t0) / reps
    del a, b
    torch.cuda.empty_cache()
    where = None
    try:    # which GPU of which node this is (best effort: the pool's boxes differ, see above)
        import shutil, subprocess
        smi = shutil.which("rocm-smi")
        # under rocprofv3 every child inherits the profiler's preload (which initialises the GPU) and rocm-smi's `#!/usr/bin/env python3`
        # is then an exec from a GPU-initialised process, which this pool's boxes refuse: no query there, and no env hop anywhere
        if smi is None or "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
            raise RuntimeError("no rocm-smi query under a profiler")
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
        txt = subprocess.run([sys.executable, os.path.realpath(smi), "--showbus", "--showproductname", "--showmemorypartition", "--showcomputepartition"],
                             capture_output=True, text=True, timeout=20, env=env).stdout
        keep = [ln.split(":", 1)[1].strip().replace("\t", " ") for ln in txt.splitlines()
                if ln.startswith("GPU[0]") and any(k in ln for k in ("PCI Bus", "Node ID", "GUID", "Partition"))]
        where = "; ".join(keep) or None
    except Exception:
        pass
    return {"name": prop.name, "compute_units": prop.multi_processor_count, "where": where, "copy_1gib_gbs": round(2 * n * 4 / dt / 1e9, 1),
            "note": "torch device-to-device copy of 1 GiB (read + write), a yardstick for the box's memory state, not a roofline"}


def flow_distance(torch, got, want):
    """relative L2 distance of two flow fields held on a device"""
    num = ((got[0] - want[0]).double() ** 2).sum() + ((got[1] - want[1]).double() ** 2).sum()
    den = (want[0].double() ** 2).sum() + (want[1].double() ** 2).sum()
    return float(torch.sqrt(num / den))


def pair_lanes(args, capi, shard, synth, torch, dist, world, rank, dev, n, prm):
    """The pair workload with several pairs in flight per GPU (--lanes L): one pair's latency-bound coarse levels run under
    another's bandwidth-bound fine levels.  Not the headline configuration (that is one pair at a time); a step is one
    pair per lane."""
    import threading
    lanes = args.lanes
    pairs = [synth.lattice_scene(n, n, seed=20240613 + 2 + rank + 31 * ln, device=dev) for ln in range(lanes)]
    plans = [capi.Plan(n, n, 1, prm) for _ in range(lanes)]
    for pl in plans:
        if lanes > 2:
            pl.set_lane_mode(1)      # beside more than one other lane only the tiny levels keep the persistent solve (octane_vof_batch_run does the same)
        elif lanes == 2:
            pl.set_lane_mode(2)      # two lanes: persistent solves concurrent, each on half the CUs (octane_vof_batch_run does the same)
    outs = [(torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)) for _ in range(lanes)]

    def lane_work(ln, count):
        a, b = pairs[ln]
        u, v = outs[ln]
        for _ in range(count):
            plans[ln].solve_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), 0, 0, capi.STREAM_OWN)
        plans[ln].wait()

    def run(count):
        th = [threading.Thread(target=lane_work, args=(ln, count)) for ln in range(lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(max(1, args.warmup))
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev if world > 1 else None)
    if rank == 0:
        print(json.dumps({"metric": "Mpix/s (full pyramid) at %dx%d, %d pairs in flight per GPU" % (n, n, lanes),
                          "value": round(shard.whole_job_mpix(world * lanes * n * n, args.steps, elapsed), 3), "unit": "Mpix/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"{lanes} concurrent {n}x{n} pairs per GPU, kiters={args.kiters} liters={args.liters} "
                                                 f"cgiters={args.cgiters}; a step is one pair per lane (not the headline configuration)",
                                     "sharding": "independent pairs, no data-path collective"},
                          "roofline": None, "cpu_baseline": None}), flush=True)
    for pl in plans:
        pl.close()
    if world > 1:
        dist.destroy_process_group()


def batch64_leg(args, capi, shard, synth, torch, dist, world, rank, local, dev):
    """BASELINE.json configs[4]: 64 independent 2000x2000 pairs (kiters=6) over the ranks, pair b on rank b % world;
    each GPU runs two lanes (two plans, each on its private stream, one host thread each) so one pair's
    latency-bound coarse levels overlap the others' bandwidth-bound fine levels.  Strong scaling: the work is fixed
    at 64 pairs per step."""
    n, npairs = 2000, 64
    # round 1 (no persistent solves): three or four lanes, one per hardware queue; round 2: two lanes with the persistent solves uncapped
    # (188.6 against 182.4 / 177.3 Mpix/s with three / four lanes on the same box)
    lanes = int(os.environ.get("OCTANE_BENCH_LANES", "2"))     # octane_vof_batch_run's choice (vof_plan.hip): two lanes, persistent solves uncapped
    prm = capi.FlowParams(kiters=6, liters=args.liters, cgiters=args.cgiters, device=local)
    mine = shard.pairs_for_rank(npairs, rank, world)
    # four distinct resident pairs per rank stand in for its share (inputs stay in HBM; values do not matter for time)
    pool = [synth.lattice_scene(n, n, seed=20240613 + 4 + 97 * rank + i, device=dev) for i in range(4)]
    plans = [capi.Plan(n, n, 1, prm) for _ in range(lanes)]
    for pl in plans:
        if os.environ.get("OCTANE_BENCH_LANE_MODE", "1") != "0":
            if lanes > 2:
                pl.set_lane_mode(1)      # as octane_vof_batch_run does for more than two lanes
            elif lanes == 2:
                pl.set_lane_mode(2)      # ... and for two: persistent solves concurrent, each capped at half the CUs (round 5: +9 %)
    outs = [(torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)) for _ in range(lanes)]

    # Each lane runs on its plan's private stream: those sit on different hardware queues, so one pair's latency-bound
    # coarse levels really do overlap the others' fine levels (+48 % with three lanes); two streams of torch's pool may
    # share a hardware queue (ROCm spreads streams over GPU_MAX_HW_QUEUES = 4 queues) and then never overlap -- which
    # is what an earlier version of this workload measured as "lanes do not help".
    import threading

    def lane_work(ln):       # one host thread per lane: ctypes drops the GIL while a pyramid's ~3300 launches are issued
        for j, _b in enumerate(mine):
            if j % lanes != ln:
                continue
            a, b = pool[j % len(pool)]
            u, v = outs[ln]
            plans[ln].solve_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), stream=capi.STREAM_OWN)   # zero first guess

    def step():
        th = [threading.Thread(target=lane_work, args=(ln,)) for ln in range(lanes)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    def barrier():
        for p in plans:
            p.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev if os.environ.get("OCTANE_BENCH_BACKEND", "nccl") == "nccl" else None)
    for p in plans:
        p.close()
    if rank != 0:
        return None
    return {"metric": "Mpix/s (full pyramid), batch of 64 pairs of 2000x2000", "value": round(npairs * n * n * args.steps / elapsed / 1e6, 3),
            "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"64 independent 2000x2000 pairs, kiters=6 liters={args.liters} cgiters={args.cgiters} "
                                   f"(BASELINE.json configs[4]); pair b on rank b % {world}, {lanes} lanes per GPU",
                       "sharding": "independent pairs, no data-path collective"},
            "roofline": None, "cpu_baseline": None}


def batch64(args, capi, shard, synth, torch, dist, world, rank, local, dev):
    out = batch64_leg(args, capi, shard, synth, torch, dist, world, rank, local, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def tiled(args, capi, synth, torch):
    """BASELINE.json configs[3]: ONE frame (--size, default 10848) solved by --bands row bands, band b on device
    b % (visible devices).  A single process drives all bands (octane_vof_tiled_*: one host thread per band, peer reads
    and event ordering between devices).  Which transport carries what crosses bands is the library's first-contact self-check's
    choice (include/octane_vof.h: inplace -> inplace without LDS-DMA from the neighbour -> copy); the frame is then solved once by the
    plain plan and once as bands BEFORE anything is timed, and if the banded flow is not the plain plan's (2e-5, equal iteration
    counts) the bench falls back to the next transport itself instead of failing -- the JSON line says which transport ran and why.
    On a one-GPU box the bands are virtual ranks sharing device 0: that run checks the whole mechanism but says nothing about
    multi-GPU speed (the bands' persistent kernels then queue behind each other)."""
    n = args.size
    ndev = capi.lib().octane_device_count()
    devices = [b % ndev for b in range(args.bands)]
    dev = torch.device("cuda", devices[0])
    torch.cuda.set_device(dev)
    a, b = synth.lattice_scene(n, n, seed=TILED_SEED, device=dev)
    z = torch.zeros(n, n, device=dev)
    prm = capi.FlowParams(kiters=args.kiters, liters=args.liters, cgiters=args.cgiters)
    torch.cuda.synchronize()
    # the plain plan's answer on band 0's device first: the banded solve is checked against it before anything is timed
    pu, pv = torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=args.kiters, liters=args.liters, cgiters=args.cgiters, device=devices[0]))
    pl.run_device(a.data_ptr(), b.data_ptr(), pu.data_ptr(), pv.data_ptr())
    torch.cuda.synchronize()
    plain_its = pl.last_iterations()
    pl.close()
    peers = sorted({(devices[i], devices[i + 1]) for i in range(len(devices) - 1) if devices[i] != devices[i + 1]})
    # candidates: what the environment / the self-check chose, then the copy transport (kernels read local memory only)
    tries, attempts, tp = [os.environ.get("OCTANE_TILED_TRANSPORT")] + ([] if os.environ.get("OCTANE_TILED_TRANSPORT") == "copy" else ["copy"]), [], None
    for forced in tries:
        if forced:
            os.environ["OCTANE_TILED_TRANSPORT"] = forced
        try:
            tp = capi.TiledPlan(n, n, 1, prm, nbands=args.bands, devices=devices)
            info = tp.transport_info()
            print(f"bench.py tiled preflight: {args.bands} bands on devices {devices} ({ndev} visible); transport '{info['transport_used']}'"
                  f"{'' if info['q_dma'] else ' without LDS-DMA from the neighbouring band'} (self-check: {info['selfcheck']}; peer access "
                  f"{'ok' if info['peer_ok'] else 'NOT available'}); "
                  + (f"peer pairs {peers}" if peers else "all bands share one device: VIRTUAL bands, no xGMI traffic"), file=sys.stderr, flush=True)
            tp.load_device(a.data_ptr(), b.data_ptr(), z.data_ptr(), z.data_ptr())     # the pair resident on every band's device
            tp.solve()
            ou, ov = torch.empty(n, n, device=dev), torch.empty(n, n, device=dev)
            tp.fetch_device(ou.data_ptr(), ov.data_ptr())
            torch.cuda.synchronize()
            parity = flow_distance(torch, (ou, ov), (pu, pv))
            parity_its = tp.last_iterations()
            del ou, ov
            ok = bool(parity <= TILED_PARITY_BAR and plain_its == parity_its)
            attempts.append({"transport": info["transport_used"], "q_dma": info["q_dma"], "rel_l2": parity, "iterations_banded": parity_its, "ok": ok})
        except capi.OctaneError as e:
            attempts.append({"transport": forced or "auto", "error": str(e), "ok": False})
            ok = False
        if ok:
            break
        print(f"bench.py tiled: the banded solve under transport {attempts[-1]['transport']} is not the plain plan's "
              f"({attempts[-1].get('rel_l2', attempts[-1].get('error'))}); falling back", file=sys.stderr, flush=True)
        if tp is not None:
            tp.close(); tp = None
    if not attempts[-1]["ok"]:
        print(json.dumps({"metric": "Mpix/s (full pyramid) at %dx%d, one frame as row bands" % (n, n), "value": None, "n_gpus": len(set(devices)),
                          "error": "no transport reproduces the plain plan", "attempts": attempts}), flush=True)
        raise SystemExit(3)
    for _ in range(max(0, args.warmup - 1)):
        tp.solve()
    tp.wait()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tp.solve()
    tp.wait()
    elapsed = time.perf_counter() - t0
    iters, expect = tp.last_iterations(), args.kiters * 3 * args.liters * args.cgiters
    if iters < 0 or (iters != expect and not args.allow_early_exit):
        print(f"bench.py tiled: the timed solves ran {iters} PCG iterations per pyramid, expected {expect}", file=sys.stderr)
        raise SystemExit(4)
    ngpu = len(set(devices))
    out = {"metric": "Mpix/s (full pyramid) at %dx%d, one frame as row bands" % (n, n),
           "value": round(n * n * args.steps / elapsed / 1e6, 3), "unit": "Mpix/s", "n_gpus": ngpu, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{n}x{n} pair, kiters={args.kiters} liters={args.liters} cgiters={args.cgiters} nchan=1 alpha=5 "
                                  f"lambda=1 (BASELINE.json configs[3]), {iters} PCG iterations per pyramid (expected {expect}); "
                                  f"{args.bands} row bands on {ngpu} device(s)" + (" -- VIRTUAL bands sharing one GPU" if ngpu < args.bands else ""),
                      "sharding": f"row bands of the {tp.banded_levels} finest level(s), coarser levels replicated; per PCG iteration "
                                  "one event-ordered phase boundary, partial sums and a few rows per inner edge read in "
                                  "place from the neighbouring band (or copied: see transport)",
                      "device_bytes_per_band": tp.device_bytes, "devices": devices, "ranks": 1, "backend": None,
                      "peer_pairs": [list(p) for p in peers]},
           "transport": dict(tp.transport_info(), attempts=attempts, bench_fell_back=len(attempts) > 1),
           "parity_vs_plain": {"rel_l2": attempts[-1]["rel_l2"], "bar": TILED_PARITY_BAR, "iterations_plain": plain_its,
                               "iterations_banded": attempts[-1]["iterations_banded"], "ok": bool(attempts[-1]["ok"])},
           "roofline": None, "cpu_baseline": None}
    tp.close()
    print(json.dumps(out), flush=True)


def tiled_mp_leg(args, capi, shard, synth, torch, dist, world, rank, local, dev):
    """Returns (the JSON object on rank 0 / None elsewhere, exit code: 0 ok, 3 no transport reproduces the plain plan, 4 the timed solves
    are not valid).  Collective: every rank walks the same control flow whatever happens.

    BASELINE.json configs[3] under the one-process-per-GPU launch (torchrun): rank r owns row band r of the ONE frame
    (octane_vof_mp_*).  torch.distributed carries the rendezvous, the handle all-gather, the timing -- and, registered with the
    library as its collective transport (octane_amd/exchange.py: backend nccl = RCCL over xGMI on device buffers, gloo staged through
    the host), the bands' exchange itself whenever HIP IPC mappings are not available or the first-contact self-check finds the
    in-place and copy transports wanting.  As in the thread form the banded flow is compared with the plain plan's before anything
    is timed, and the bench falls back (copy, then collective) by itself instead of failing.  Every rank holds the whole pair."""
    from octane_amd import exchange as xch
    n = args.size
    t_leg = time.perf_counter()

    def note(msg):      # progress on stderr: a leg that hangs on a node has to say where (rank 0 speaks for all)
        if rank == 0:
            print(f"bench.py tiled [{time.perf_counter() - t_leg:6.1f} s] {msg}", file=sys.stderr, flush=True)
    a, b = synth.lattice_scene(n, n, seed=TILED_SEED, device=dev)          # same seed on every rank: the same frame
    u = torch.zeros(n, n, device=dev)
    v = torch.zeros(n, n, device=dev)
    prm = capi.FlowParams(kiters=args.kiters, liters=args.liters, cgiters=args.cgiters, device=local)
    backend = str(dist.get_backend())

    def all_gather(blob):
        out = [None] * world
        dist.all_gather_object(out, blob)
        return out

    torch.cuda.synchronize()
    plain_its = None
    if rank == 0:    # the plain plan's answer on rank 0's device: the banded solve is checked against it before anything is timed
        pu, pv = torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)
        pl = capi.Plan(n, n, 1, prm)
        pl.run_device(a.data_ptr(), b.data_ptr(), pu.data_ptr(), pv.data_ptr())
        torch.cuda.synchronize()
        plain_its = pl.last_iterations()
        pl.close()
    note(f"{n}x{n} frame on every rank, plain plan solved on rank 0 ({plain_its} iterations)")
    dist.barrier()
    ex = xch.TorchExchange(dev)
    first = os.environ.get("OCTANE_TILED_TRANSPORT")
    tries = [first] + [t for t in ("copy", "collective") if t != first]
    attempts, mp = [], None
    for forced in tries:
        if forced:
            os.environ["OCTANE_TILED_TRANSPORT"] = forced
        nonce = [os.urandom(6).hex() if rank == 0 else None]       # a name no earlier (crashed) run can have left behind
        dist.broadcast_object_list(nonce, src=0)
        verdict = [None]
        try:
            mp = capi.MpPlan(n, n, 1, prm, rank, world, "/octane_bench_%s" % nonce[0], all_gather, exchange=ex)
            note(f"band plans created and connected (transport {mp.transport_info()['transport_used']}, self-check {mp.transport_info()['selfcheck']})")
            mp.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr())
            torch.cuda.synchronize()
            note("first banded solve done")
            info = mp.transport_info()
            if rank == 0:
                parity = flow_distance(torch, (u, v), (pu, pv))
                its = mp.last_iterations()
                verdict = [{"transport": info["transport_used"], "q_dma": info["q_dma"], "rel_l2": parity, "iterations_banded": its,
                            "ok": bool(parity <= TILED_PARITY_BAR and its == plain_its)}]
            mine_ok = True
        except capi.OctaneError as e:
            if rank == 0:
                verdict = [{"transport": forced or "auto", "error": str(e), "ok": False}]
            mine_ok = False
        oks = all_gather(mine_ok)
        dist.broadcast_object_list(verdict, src=0)
        verdict[0]["ok"] = bool(verdict[0]["ok"] and all(oks))
        attempts.append(verdict[0])
        if verdict[0]["ok"]:
            break
        if rank == 0:
            print(f"bench.py tiled: the banded solve under transport {verdict[0]['transport']} is not the plain plan's "
                  f"({verdict[0].get('rel_l2', verdict[0].get('error'))}); falling back", file=sys.stderr, flush=True)
        if mp is not None:
            mp.close(); mp = None
    if not attempts[-1]["ok"]:
        dist.barrier()
        return ({"metric": "Mpix/s (full pyramid) at %dx%d, one frame as row bands" % (n, n), "value": None, "n_gpus": world,
                 "error": "no transport reproduces the plain plan", "attempts": attempts} if rank == 0 else None), 3
    if rank == 0:
        del pu, pv
        info = mp.transport_info()
        print(f"bench.py tiled preflight: {world} ranks, backend {backend}; transport '{info['transport_used']}' (self-check: {info['selfcheck']}; "
              f"IPC mappings {'ok' if info['peer_ok'] else 'NOT available'}; {info['devices']} distinct device(s)); collective library: {ex.name}",
              file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(0, args.warmup - 1)):
        mp.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr())
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mp.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr())     # blocking and collective
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev if os.environ.get("OCTANE_BENCH_BACKEND", "nccl") == "nccl" else None)
    note(f"{args.steps} timed solves done: {elapsed * 1e3 / args.steps:.1f} ms each")
    # the verdict on the TIMED solves is formed on rank 0 before anything is printed and shared with every rank: a run whose last
    # solve was abandoned (-2) or ran another number of iterations than the configuration names reports no value and fails on all
    # ranks, as the thread form (tiled) does
    iters, expect = mp.last_iterations(), args.kiters * 3 * args.liters * args.cgiters
    timed = [None]
    if rank == 0:
        timed = [{"iterations": iters, "expected": expect,
                  "ok": bool(iters >= 0 and (iters == expect or args.allow_early_exit))}]
    dist.broadcast_object_list(timed, src=0)
    out = None
    if rank == 0:
        workload = (f"{n}x{n} pair, kiters={args.kiters} liters={args.liters} cgiters={args.cgiters} nchan=1 alpha=5 "
                    f"lambda=1 (BASELINE.json configs[3]), {iters} PCG iterations per pyramid (expected {expect}); "
                    f"one row band per rank, {world} ranks")
        if not timed[0]["ok"]:
            print(f"bench.py tiled: the timed solves ran {iters} PCG iterations per pyramid, expected {expect}"
                  + (" (a persistent solve was abandoned)" if iters < 0 else ""), file=sys.stderr)
            out = {"metric": "Mpix/s (full pyramid) at %dx%d, one frame as row bands" % (n, n), "value": None, "n_gpus": world,
                   "error": "the timed solves are not valid: %d PCG iterations per pyramid, expected %d" % (iters, expect),
                   "config": {"workload": workload}, "attempts": attempts}
        else:
            out = {"metric": "Mpix/s (full pyramid) at %dx%d, one frame as row bands" % (n, n),
                   "value": round(n * n * args.steps / elapsed / 1e6, 3), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": workload,
                              "sharding": f"row bands of the {mp.banded_levels} finest level(s), coarser levels replicated; what crosses ranks per PCG "
                                          "iteration (partial sums, a few rows per inner edge) is read through HIP IPC mappings, copied through them, "
                                          "or moved by torch.distributed: see transport",
                              "ranks": world, "backend": backend},
                   "transport": dict(mp.transport_info(), attempts=attempts, bench_fell_back=len(attempts) > 1, exchange_calls=ex.calls),
                   "parity_vs_plain": {"rel_l2": attempts[-1]["rel_l2"], "bar": TILED_PARITY_BAR, "iterations_plain": plain_its,
                                       "iterations_banded": attempts[-1]["iterations_banded"], "ok": bool(attempts[-1]["ok"])},
                   "roofline": None, "cpu_baseline": None}
    mp.close()
    dist.barrier()
    return out, (0 if timed[0]["ok"] else 4)


def tiled_mp(args, capi, shard, synth, torch, dist, world, rank, local, dev):
    """--workload tiled under torchrun: the leg above, printed; a run whose banded solve is not the plain plan's under any transport
    (3) or whose timed solves are not valid (4) reports no value and fails on all ranks, as the thread form (tiled) does."""
    out, code = tiled_mp_leg(args, capi, shard, synth, torch, dist, world, rank, local, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    if code:
        raise SystemExit(code)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(ngpus):
    """--gpus N without a launcher: start N fresh ranks (one per GPU) and exit with their code.  Nothing in this process
    has touched the GPU yet (torch.cuda.device_count() does not initialise it), and the ranks are child processes, not
    an exec of this one."""
    import subprocess
    import torch
    visible = torch.cuda.device_count()
    one_device = os.environ.get("OCTANE_BENCH_ONE_DEVICE") == "1"       # rehearsal: all ranks share GPU 0
    if visible < ngpus and not one_device:
        print(f"bench.py: --gpus {ngpus} but only {visible} GPU(s) visible (set OCTANE_BENCH_ONE_DEVICE=1 with "
              f"OCTANE_BENCH_BACKEND=gloo to rehearse the N-rank control flow on one GPU)", file=sys.stderr)
        raise SystemExit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def run_name(n, kiters, liters, cgiters):
    """Which of SURVEY 8d's named runs a parameter set is (they differ in work per pixel by 2.8x; never substitute silently)."""
    if (kiters, liters, cgiters) == (8, 3, 30):
        return "SURVEY 8d run R1" + (" = BASELINE.json configs[2]" if n == 5000 else "")
    if (kiters, liters, cgiters) == (8, 10, 10):
        return "SURVEY 8d run R2 (300 PCG iterations per level, 240 warps; API-only, cgiters is not a CLI flag)"
    if (kiters, liters, cgiters) == (10, 10, 30):
        return "SURVEY 8d run R3 (the metric string's '300 warps': kiters * 3 * liters = 300)"
    if (n, kiters, liters, cgiters) == (2000, 6, 3, 30):
        return "BASELINE.json configs[1]"
    return "not one of SURVEY 8d's named runs"


SECONDARY_RUNS = (("R2", 8, 10, 10), ("R3", 10, 10, 30))   # R2 = 300 PCG iterations per level; R3 = the metric string's "300 warps"


def secondary_runs(args, capi, torch, a, b, u, v, n, local, steps=3, warmup=1):
    """SURVEY 8d's other two named runs on the same resident pair, after the R1 headline: the metric string of BASELINE.json
    ("full pyramid, 300 warps") reads as R3 (kiters * 3 * liters = 300 assemblies), its config line ("8 levels x 300 SOR iters") as
    R2 (liters * 3 * cgiters = 300 PCG iterations per level).  `steps` timed solves each, one plan per run (own placement trials).
    R3's coarse solves stop by the tolerance test (ref .cu:1131: with 300 linearisations per level they converge), so its
    iteration count is reported next to the cap rather than asserted equal -- the --allow-early-exit semantics; an abandoned
    persistent solve (-2) still fails.  Ref: src/main.cc:82-86,142-144."""
    out = {}
    stream = torch.cuda.current_stream().cuda_stream
    for name, kiters, liters, cgiters in SECONDARY_RUNS:
        prm = capi.FlowParams(kiters=kiters, liters=liters, cgiters=cgiters, device=local)
        plan = capi.Plan(n, n, 1, prm)
        try:
            def step():
                u.zero_(); v.zero_()
                plan.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            iters, cap = plan.last_iterations(), kiters * 3 * liters * cgiters
            if iters < 0:
                print(f"bench.py: secondary run {name}: a persistent solve was abandoned ({iters})", file=sys.stderr)
                raise SystemExit(4)
            if iters != cap and name != "R3":
                print(f"bench.py: secondary run {name} ran {iters} PCG iterations per pyramid, expected {cap}", file=sys.stderr)
                raise SystemExit(4)
            out[name] = {"value": round(n * n * steps / dt / 1e6, 3), "unit": "Mpix/s", "ms_per_step": round(dt * 1e3 / steps, 3),
                         "steps": steps, "warmup": warmup,
                         "config": {"workload": f"{n}x{n} pair, kiters={kiters} liters={liters} cgiters={cgiters} nchan=1 alpha=5 lambda=1 "
                                                f"({run_name(n, kiters, liters, cgiters)})"},
                         "assemblies_per_pyramid": kiters * 3 * liters, "pcg_iterations_per_pyramid": iters, "pcg_iteration_cap": cap,
                         "early_exit": ("allowed: coarse solves stop by the tolerance test (ref .cu:1131)" if name == "R3" else "not allowed: count asserted"),
                         "inputs": "resident in HBM (same pair as the headline)"}
        finally:
            plan.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs = ranks (one process per GPU).  N > 1 without a launcher "
                    "starts the N ranks itself; under torchrun it has to equal WORLD_SIZE.  Default: WORLD_SIZE, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=None, help="frame edge: 5000 (pair), 10848 (tiled) unless given")
    ap.add_argument("--kiters", type=int, default=8)
    ap.add_argument("--liters", type=int, default=3)
    ap.add_argument("--cgiters", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-transfers", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip SURVEY 8d's runs R2 / R3 (3 steps each after an R1 headline)")
    ap.add_argument("--workload", default="pair", choices=["pair", "batch64", "tiled"],
                    help="pair (default): one --size pair per GPU; batch64: BASELINE.json configs[4], 64 pairs of "
                         "2000x2000 (kiters=6) shared by all ranks, two concurrent lanes per GPU (OCTANE_BENCH_LANES overrides); tiled: BASELINE.json "
                         "configs[3], one --size frame as row bands: --bands bands driven by a single process, or -- "
                         "under torchrun -- one band per rank")
    ap.add_argument("--bands", type=int, default=4, help="row bands of the tiled workload")
    ap.add_argument("--nchan", type=int, default=1, choices=[1, 2, 3], help="pair workload only: channels of the synthetic pair (the reference's loop "
                    "handles 1 ... 3 alike, ref .cu:749-829; BASELINE's configurations are single-channel: not the headline with 2 or 3)")
    ap.add_argument("--lanes", type=int, default=1, help="pair workload only: this many pairs in flight per GPU, each on its own plan, "
                    "stream and host thread (a step is then one pair per lane); 1 = the headline configuration")
    ap.add_argument("--cpu-sample", type=int, default=0, help="0 (default): the CPU baseline is the workload's own frame and pyramid with one "
                    "linearisation per GNC step, scaled by 1 / liters; M > 0: rounds 1-3's M x M sample with at most 4 levels")
    ap.add_argument("--allow-early-exit", action="store_true", help="do not fail when solves stop early by the tolerance test "
                    "(fewer PCG iterations than kiters * 3 * liters * cgiters); an abandoned persistent solve always fails the run")
    args = ap.parse_args()
    if args.size is None:
        args.size = 10848 if args.workload == "tiled" else 5000
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus is not None and args.gpus > 1:
            self_launch(args.gpus)                      # does not return
        args.gpus = 1
    else:
        if args.gpus is not None and args.gpus != int(env_world):
            print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={env_world}", file=sys.stderr)
            raise SystemExit(2)
        args.gpus = int(env_world)

    import torch
    import torch.distributed as dist
    from octane_amd import capi, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs (not used by the driver): OCTANE_BENCH_BACKEND=gloo and OCTANE_BENCH_ONE_DEVICE=1 let several
    # ranks share one GPU so that the N>1 control flow can be exercised on a one-GPU box.
    backend = os.environ.get("OCTANE_BENCH_BACKEND", "nccl")
    if os.environ.get("OCTANE_BENCH_ONE_DEVICE") == "1":
        local = 0
        # ranks of a rehearsal share one GPU: their persistent PCG solves (one workgroup per CU, all resident at once) cannot be
        # serialised across processes, so each rank may only hold its share of the CUs
        os.environ.setdefault("OCTANE_TUNE_PERSIST_MAXG", str(max(1, 256 // max(1, world))))
    if world > 1:   # one process per GPU over RCCL; only the barrier and the max-over-ranks time use it
        if backend == "nccl":
            shard.init_from_env("nccl", device_id=torch.device("cuda", local))
        else:
            shard.init_from_env(backend)
    ndev = capi.lib().octane_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if local >= ndev:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but {ndev} are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.workload == "tiled":
        if world > 1:      # torchrun: one band per rank
            return tiled_mp(args, capi, shard, synth, torch, dist, world, rank, local, dev)
        return tiled(args, capi, synth, torch)
    if args.workload == "batch64":
        return batch64(args, capi, shard, synth, torch, dist, world, rank, local, dev)
    n = args.size
    prm = capi.FlowParams(kiters=args.kiters, liters=args.liters, cgiters=args.cgiters, device=local)
    if args.lanes > 1:
        return pair_lanes(args, capi, shard, synth, torch, dist, world, rank, dev, n, prm)
    a, b = synth.lattice_scene(n, n, seed=20240613 + 2 + rank, nchan=args.nchan, device=dev)
    u = torch.zeros(n, n, device=dev)
    v = torch.zeros(n, n, device=dev)
    plan = capi.Plan(n, n, args.nchan, prm)
    if args.nchan != 1:          # the host-buffer, secondary and CPU legs are defined on the single-channel configurations
        args.no_transfers = args.no_secondary = args.no_cpu_baseline = True
    tr = sorted(t for t in plan.placement_trials() if t > 0)
    trials_ms = {"n": len(tr), "min_ms": round(tr[0], 4), "median_ms": round(tr[len(tr) // 2], 4), "max_ms": round(tr[-1], 4),
                 "what": "ms per launch of the trial PCG iterations on each candidate arena (stop test held open, varying weights, x work "
                         "in every second launch: 64 B/pixel against 61.3 in a solve's mix; an event pair per launch since round 3): "
                         "comparable among the candidates; the solve's own mean launch time is roofline.avg_launch_ms"} if tr else None
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        u.zero_(); v.zero_()           # zero first guess, as oct_optical_flow.cc:38-48
        plan.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=dev if backend == "nccl" else None)
    iters = plan.last_iterations()
    expect = args.kiters * 3 * args.liters * args.cgiters
    if iters < 0 or (iters != expect and not args.allow_early_exit):
        # -2: a persistent mid-level solve of the last timed step was abandoned (its flow is not valid); fewer than expected: solves
        # stopped by the tolerance test, i.e. less work than the configuration names.  Either way the time is not a measurement.
        print(f"bench.py: rank {rank}: the timed steps ran {iters} PCG iterations per pyramid, expected {expect}"
              + (" (a persistent solve was abandoned: GPU shared with another process?)" if iters == -2 else ""), file=sys.stderr)
        raise SystemExit(4)
    ms_per_step = elapsed * 1e3 / args.steps
    value = shard.whole_job_mpix(world * n * n, args.steps, elapsed)

    # per-kernel durations of the finest level: one extra, untimed, profiled step (HIP events are
    # recorded on the launch stream around every finest-level launch)
    roof = None
    if rank == 0:
        plan.set_profiling(True)
        step()
        torch.cuda.synchronize()
        pr = plan.profile()
        plan.set_profiling(False)
        a_ms = pr.pass_a_ms / max(1, pr.pass_a_launches)
        b_ms = pr.pass_b_ms / max(1, pr.pass_b_launches)
        unit_w = True      # (the product library no longer reads the OCTANE_TUNE_* tuning variables, round 5: the defaults are what runs)
        fused = pr.pass_b_launches == 0          # one kernel per PCG iteration (the default); its launches are timed as "pass A"
        if fused:
            # mean algorithmic bytes of a finest-level launch: 80 B/px, 72 in the first of the three GNC steps
            imm = 0                                # (x is updated by every second launch)
            qform = n * n >= (2 << 20)
            b_all, b_gnc0 = ((FUSED_Q_BYTES_PER_PIXEL, FUSED_Q_BYTES_PER_PIXEL_GNC0) if qform
                             else (FUSED_BYTES_PER_PIXEL, FUSED_BYTES_PER_PIXEL_GNC0))
            bpp = ((b_gnc0 + 2 * b_all) / 3.0 if unit_w else b_all) + imm
            qname = "k_pcg_fused_q_dma" if os.environ.get("OCTANE_TUNE_Q_DMA", "1") != "0" else "k_pcg_fused_q"   # LDS-DMA staging is the default
            dom, dms = (qname if qform else "k_pcg_fused"), a_ms
            iter_ms = a_ms
        else:
            bpp_a = (PASS_A_BYTES_PER_PIXEL_GNC0 + 2 * PASS_A_BYTES_PER_PIXEL) / 3.0 if unit_w else PASS_A_BYTES_PER_PIXEL
            if pr.pass_a_ms >= pr.pass_b_ms:
                dom, dms, bpp = "k_pcg_pass_a", a_ms, bpp_a
            else:
                dom, dms, bpp = "k_pcg_pass_b", b_ms, PASS_B_BYTES_PER_PIXEL
            iter_ms = a_ms + b_ms
        # HBM-side traffic of that kernel from the PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE cannot be
        # collected inside this run; tools/profile_round.sh + tools/summarize_rocprof.py produce the file)
        traffic = traffic_source = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            tk = "k_pcg_fused" if fused else dom          # the summariser files every fused instance under one name
            if tj.get(tk, {}).get("size") == n and tj[tk].get("kernel", dom) == dom:
                # the counters describe a kernel TEXT: if the kernel's sources have changed since they were collected the figure is stale
                # and is not reported ("goes stale silently the next time the kernel text changes")
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from summarize_rocprof import kernel_source_sha1
                same_text = tj.get("kernel_source_sha1") == kernel_source_sha1()
                if same_text:
                    traffic = tj[tk]["read_bytes"] + tj[tk]["write_bytes"]
                traffic_source = ("profiles/traffic.json @ %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this kernel, whose "
                                  "source text is unchanged since; NOT measured in this run)" % tj.get("commit", "unknown commit")) if same_text else \
                                 ("profiles/traffic.json @ %s is STALE: the kernel's sources have changed since those counter passes; not reported"
                                  % tj.get("commit", "unknown commit"))
        except (OSError, ValueError):
            pass
        # the four kinds of finest-level launch, each priced on its own bytes (q-recomputing kernel; launch k of a solve: k = 0 forms no
        # q and reads no p; even k > 0 applies two x updates -- x and the p before last in, x out; the first GNC step reads no weights)
        by_kind = None
        lt = plan.launch_times()
        per_solve = args.cgiters
        if fused and qform and per_solve >= 4 and len(lt) == 3 * args.liters * per_solve:
            acc = {}
            for i, ms_i in enumerate(lt):
                k, gnc0 = i % per_solve, (i // per_solve) // args.liters == 0
                if k == 0:
                    continue                                   # the first launch of a solve is a kind of its own (no stencil, no p)
                xwork = (k % 2 == 0)
                b_px = (52 + (24 if xwork else 0)) - (8 if (gnc0 and unit_w) else 0)
                key = ("first GNC step (weights -1, not read)" if (gnc0 and unit_w) else "varying weights") + (", with x work" if xwork else ", without x work")
                ent = acc.setdefault(key, [0.0, 0, b_px]); ent[0] += ms_i; ent[1] += 1
            by_kind = {key: {"launches": c, "bytes_per_pixel": b_px, "avg_launch_ms": round(t / c, 4),
                             "frac": round(b_px * n * n / (t / c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for key, (t, c, b_px) in acc.items()}
        achieved = bpp * n * n / (dms * 1e-3) / 1e9
        iter_gbs = 116 * n * n / (iter_ms * 1e-3) / 1e9
        survey_bpp = 116 if fused else (60 if dom == "k_pcg_pass_a" else 56)
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                # `traffic` sits on the L2's fabric side and counts Infinity-Cache hits; the DRAM-side split the round-5 review asked for does
                # not exist on gfx950: TCC_EA0_RDREQ_DRAM / _WRREQ_DRAM equal TCC_EA0_RDREQ / _WRREQ to the last request ("DRAM" = the request's
                # destination class, not a miss of the memory-side cache), profiles/r6_dram_counters.txt
                "traffic_dram": None,
                "avg_launch_ms": round(dms, 4), "bytes_per_launch": int(round(bpp * n * n)), "by_kind_of_launch": by_kind,
                "pcg_iteration_ms": round(iter_ms, 4),
                # one whole PCG iteration on SURVEY 8(d)'s accounting (116 B/px: pass A with seven coefficient planes + pass B).
                # This implementation moves 64 (fused kernel, five planes, x every second launch, q formed twice instead of stored;
                # 80 on levels below 2 Mi pixels, where q is stored) -- the figure above counts those, the stricter one.
                # the same launch priced with the bytes rocprofv3 counted on the L2's fabric side (profiles/traffic.json;
                # Infinity-Cache hits included): what "rocprof achieved GB/s against the roofline" reads
                "rocprof_traffic_gbs": round(traffic / (dms * 1e-3) / 1e9, 1) if traffic else None,
                "rocprof_traffic_frac": round(traffic / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "pcg_iteration_gbs_at_116B_per_pixel": round(iter_gbs, 1),
                "frac_at_survey_bytes": round(survey_bpp * n * n / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "assemble_ms": round(pr.assemble_ms / max(1, pr.assemble_launches), 4),
                "setup_ms_all_levels": round(pr.setup_ms, 3), "profiled_step_ms": round(pr.total_ms, 2)}

    # SURVEY 8d's primary metric: the whole call on HOST buffers (H2D + all levels + D2H), as the C++ shim of
    # oct_variational_optical_flow costs per pair -- pageable memory (what the reference's caller has) and pinned memory.
    transfers = None
    if rank == 0 and world == 1 and not args.no_transfers:
        import numpy as np
        ha, hb = a.cpu().numpy(), b.cpu().numpy()
        transfers = {}
        for kind in ("pageable", "pinned"):
            if kind == "pinned":
                ta, tb = torch.from_numpy(ha).pin_memory(), torch.from_numpy(hb).pin_memory()
                tu, tv = torch.zeros(n, n).pin_memory(), torch.zeros(n, n).pin_memory()
                xa, xb, xu, xv = ta.numpy(), tb.numpy(), tu.numpy(), tv.numpy()
            else:
                xa, xb, xu, xv = ha, hb, np.zeros((n, n), np.float32), np.zeros((n, n), np.float32)
            best = best0 = None
            for rep in range(3):           # the first call creates the cached plan (with its placement trials); the best of the next two counts
                xu[:] = 0; xv[:] = 0
                t1 = time.perf_counter()
                capi.flow_inplace(xa, xb, xu, xv, prm)
                dt = time.perf_counter() - t1
                if rep > 0:
                    best = dt if best is None else min(best, dt)
            for rep in range(2):           # the same call told that there is no first guess (octane_vof_solve, u0 = v0 = NULL: zeros are not uploaded)
                t1 = time.perf_counter()
                capi.flow_into(xa, xb, xu, xv, prm)
                dt = time.perf_counter() - t1
                best0 = dt if best0 is None else min(best0, dt)
            transfers[kind] = {"ms": round(best * 1e3, 2), "mpix_s": round(n * n / best / 1e6, 2),
                               "no_first_guess": {"ms": round(best0 * 1e3, 2), "mpix_s": round(n * n / best0 / 1e6, 2),
                                                  "what": "octane_vof_solve with u0 = v0 = NULL, what oct_optical_flow() calls without -firstguess: "
                                                          "4 instead of 6 PCIe transfers"}}
        capi.release_cache()

    # SURVEY 8d's other named runs (R2, R3) ride along with an R1 headline: the metric string's own configuration is R3
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and (args.kiters, args.liters, args.cgiters) == (8, 3, 30):
        plan.close()
        try:        # the headline line must not be lost to a failure of a side leg: it is reported in its place
            secondary = secondary_runs(args, capi, torch, a, b, u, v, n, local)
        except (Exception, SystemExit) as e:
            secondary = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py: the secondary runs failed: {secondary['error']}", file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oct_oracle as oo     # the checker, timed as the CPU baseline ("port")
        # The sample is the WORKLOAD'S OWN frame and pyramid with ONE linearisation per GNC step instead of `liters`: a pyramid's work
        # is kiters x 3 x liters x (one assembly + cgiters PCG iterations) + the level setup, i.e. linear in liters, so the whole
        # configuration costs `liters` times the sample less (liters - 1) level setups (< 1 % of the oracle's time; not subtracted:
        # the figure understates the CPU by that much).  Rounds 1-3 sampled a 3072^2 frame, which flattered the CPU by 54 %: at 5000^2 the
        # oracle's CSR matrix (284 B/pixel, 7 GB) no longer fits the caches that sample enjoyed.  --cpu-sample M keeps the old form.
        if not args.cpu_sample and n * n > 36_000_000:
            args.cpu_sample = 3072          # a full-disk frame would take the oracle minutes even at liters = 1
        m = args.cpu_sample if args.cpu_sample else n
        ck = args.kiters if not args.cpu_sample else min(args.kiters, 4)
        cl = 1 if not args.cpu_sample else args.liters
        ca, cb = synth.lattice_scene(m, m, seed=20240613 + 2)
        t1 = time.perf_counter()
        # OpenMP build of the oracle (bit-identical to the scalar one) under the reference's launch-geometry
        # dot-product schedule, on all the host cores this process may use
        oo.set_threads(oo.host_cpu_share())
        cores = oo.num_threads("omp")
        _, _, cits = oo.flow(ca, cb, oo.FlowParams(kiters=ck, liters=cl, cgiters=args.cgiters),
                             flavour="omp", dot_threads=oo.REF_GRID_THREADS)
        ct = time.perf_counter() - t1
        lev_sum = lambda k: sum(0.25 ** i for i in range(k))
        scale = (lev_sum(ck) / lev_sum(args.kiters)) * (cl / args.liters)
        cpu_mpix = m * m / ct / 1e6 * scale
        cpu = {"value": round(cpu_mpix, 5), "unit": "Mpix/s", "cores": cores, "kind": "port",
               "sample": f"{m}x{m} lattice pair, kiters={ck} liters={cl} cgiters={args.cgiters} "
                         f"({cits} PCG iterations) in {ct:.1f} s on {cores} host cores (OpenMP); Mpix/s scaled by {scale:.4f} "
                         + (f"(the configuration runs liters={args.liters} linearisations per GNC step, the sample {cl}: work is linear in liters)"
                            if not args.cpu_sample else f"(level-pixel sums of {ck} vs {args.kiters} levels)")}
        # the same oracle on the WHOLE configuration, measured once on a GPU box's host cores (69 s: too long for every bench run)
        try:
            if (n, args.kiters, args.liters, args.cgiters) == (5000, 8, 3, 30):
                cpu["full_config_measured_once"] = json.load(open(os.path.join(ROOT, "profiles", "r3_cpu_baseline_r1.json")))
        except (OSError, ValueError):
            pass

    # N > 1: what the collective library saw (world size after init_process_group, a sum all-reduce of ones, every rank's device uuid):
    # "RCCL saw N ranks on N distinct GPUs" can be read off the line alone.  Collective, after the timed region.
    census = shard.rank_census(backend, device=dev, local=local) if world > 1 else None
    out = None
    if rank == 0:
        dstate = device_state(torch, dev)
        if roof is not None and dstate.get("copy_1gib_gbs"):
            # "vs (i) 8 TB/s nominal and (ii) a measured device copy figure from the same run": the dominant kernel's
            # algorithmic bytes per second over what a plain 1 GiB device copy (1 read + 1 write stream) delivered in this process
            roof["measured_copy_gbs"] = dstate["copy_1gib_gbs"]
            roof["frac_of_measured_copy"] = round(roof["achieved"] / dstate["copy_1gib_gbs"], 4)
        drop_in = None
        if transfers:       # the call the reference's host code makes (pageable buffers, no first guess): octane_vof_solve as oct_optical_flow() calls it
            drop_in = transfers["pageable"]["no_first_guess"]["mpix_s"]
        out = {"metric": "Mpix/s (full pyramid, pair resident in HBM) at %dx%d" % (n, n), "value": round(value, 3), "unit": "Mpix/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": f"{n}x{n} pair, kiters={args.kiters} liters={args.liters} cgiters={args.cgiters} nchan={args.nchan} "
                                      f"alpha=5 lambda=1 ({run_name(n, args.kiters, args.liters, args.cgiters) if args.nchan == 1 else 'R1-shaped, NOT a BASELINE configuration: ' + str(args.nchan) + ' channels'}), "
                                      f"{iters} PCG iterations per pyramid (expected {expect}), one pair per GPU",
                          "sharding": "independent pairs, no data-path collective"},
               # the drop-in call: one oct_variational_optical_flow-shaped call on the caller's pageable host buffers without a first guess
               # (H2D + all levels + D2H through octane_vof_solve, what the oct_optical_flow() shim costs per pair): SURVEY 8d's primary
               # metric, never `value` (which is device-resident, as the bench contract asks)
               "value_drop_in": drop_in,
               # the same call in its other forms (with a first guess uploaded, pinned buffers)
               "value_with_transfers": transfers,
               # SURVEY 8d's runs R2 and R3 (the metric string's "300 warps") on the same pair, 3 timed steps each; the headline stays R1
               "secondary": secondary,
               # the plan kept the fastest of these candidate arenas: best-of-n placement
               "placement_trials": trials_ms,
               "device": dstate,
               "roofline": roof, "cpu_baseline": cpu}
        if census is not None:
            out["ranks"] = census
    # N > 1: after the timed pair headline rank 0 measures the two multi-GPU configurations of BASELINE.json -- configs[3] (one full-disk
    # frame, one row band per rank: the first time RCCL / IPC mappings see N real devices) and configs[4] (64 pairs sharded) -- as
    # bounded, NON-FATAL side legs in CHILD jobs of N fresh ranks each: whatever happens in them -- an error, a hang, a GPU fault that
    # kills a rank --, the headline line is printed and this job leaves with exit code 0.
    if world > 1 and not args.no_secondary and args.nchan == 1 and (args.kiters, args.liters, args.cgiters) == (8, 3, 30):
        plan.close()
        del a, b, u, v
        torch.cuda.empty_cache()
        side = multi_gpu_legs(dist, world, rank)
        if rank == 0:
            out["secondary_multi_gpu"] = side
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def side_leg_commands(world):
    """The child jobs of an N-rank default run: this script under torch.distributed.run with N fresh ranks, one job per configuration."""
    size3 = os.environ.get("OCTANE_BENCH_SECONDARY_TILED_SIZE", "10848")
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1"]
    me = os.path.abspath(__file__)
    return (
        ("configs3_one_frame_as_row_bands", launch + ["--master-port", str(_free_port()), me, "--gpus", str(world), "--workload", "tiled",
                                                      "--size", size3, "--steps", "2", "--warmup", "2"], 240.0),
        ("configs4_batch_of_64_pairs", launch + ["--master-port", str(_free_port()), me, "--gpus", str(world), "--workload", "batch64",
                                                 "--steps", "2", "--warmup", "1"], 150.0),
    )


def multi_gpu_legs(dist, world, rank, legs=None):
    """The side legs of an N-rank default run.  Rank 0 starts each leg as a CHILD job (N fresh ranks of this script on the same GPUs,
    which the parents have emptied and now leave idle), bounded by a time-out (the child's whole process group is killed), and takes
    the last JSON line of its stdout; the other parent ranks wait on the rendezvous store -- a CPU-side wait, no collective kernel spins
    on the GPUs the children are using.  A leg that fails, hangs or crashes is REPORTED in `secondary_multi_gpu` and costs nothing else:
    the parents never share a process, a communicator or a GPU context with it.  Returns the dictionary on rank 0, None elsewhere."""
    import signal
    import subprocess
    store = None
    try:
        from torch.distributed import distributed_c10d
        store = distributed_c10d._get_default_store()
    except Exception:       # (a torch without that accessor: fall back to a barrier, which on nccl spins a kernel -- correct, only less tidy)
        store = None
    result = None
    if rank == 0:
        result = {}
        legs = legs if legs is not None else side_leg_commands(world)
        drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID",
                "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "NCCL_ASYNC_ERROR_HANDLING")
        env = {k: v for k, v in os.environ.items() if k not in drop}
        for name, cmd, limit in legs:
            t0 = time.perf_counter()
            leg = {}
            try:
                p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
                try:
                    so, se = p.communicate(timeout=limit)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, signal.SIGKILL)
                    so, se = p.communicate()
                    leg["error"] = f"the leg did not finish within {limit:.0f} s and was killed"
                lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
                if lines:
                    try:
                        leg.update(json.loads(lines[-1]))
                    except ValueError:
                        leg.setdefault("error", "the leg's JSON line could not be parsed")
                elif "error" not in leg:
                    leg["error"] = "the leg printed no JSON line"
                leg["exit_code"] = p.returncode
                if p.returncode != 0 or "error" in leg:
                    leg["stderr_tail"] = [ln for ln in (se or "").splitlines() if "amdgpu.ids" not in ln and "socket.cpp" not in ln][-8:]
                progress = [ln for ln in (se or "").splitlines() if ln.startswith("bench.py tiled [")]
                if progress:
                    leg["progress"] = progress[-6:]
            except Exception as e:      # the launcher itself could not be started
                leg = {"error": f"{type(e).__name__}: {e}"}
            leg["leg_seconds"] = round(time.perf_counter() - t0, 1)
            result[name] = leg
            print(f"bench.py: side leg {name}: {'value ' + str(leg.get('value')) if leg.get('value') is not None else leg.get('error', 'no value')} "
                  f"({leg['leg_seconds']} s, exit code {leg.get('exit_code')})", file=sys.stderr, flush=True)
        if store is not None:
            store.set("octane_side_legs_done", "1")
    if store is not None:
        if rank != 0:
            import datetime
            try:
                store.wait(["octane_side_legs_done"], datetime.timedelta(seconds=600))
            except Exception:
                pass
    else:
        dist.barrier()
    return result


if __name__ == "__main__":
    main()
