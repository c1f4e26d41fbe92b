"""The one-shot drop-in call (octane_vof_run, what the C++ shim of oct_variational_optical_flow makes per image pair)
keeps its plan between calls of the same shape.  Caching must never change results."""
import os
import threading

import numpy as np
import pytest

from octane_amd import synth

pytestmark = pytest.mark.gpu


def test_cached_one_shot_calls_equal_fresh_plans(capi):
    a, b = synth.lattice_scene(220, 160, seed=3)
    c, d = synth.lattice_scene(220, 160, seed=4)
    e, f = synth.lattice_scene(130, 90, seed=5)
    prm = capi.FlowParams(kiters=3)
    capi.release_cache()
    want = {}
    for key, (x, y, p) in dict(ab=(a, b, prm), cd=(c, d, prm), ef=(e, f, prm), ab2=(a, b, capi.FlowParams(kiters=3, alpha=8.0))).items():
        pl = capi.Plan(x.shape[-1], x.shape[-2], 1, p)
        want[key] = pl.run_host(x, y)
        pl.close()
    seq = [("ab", a, b, prm), ("cd", c, d, prm), ("ab", a, b, prm),            # same shape: plan reused
           ("ef", e, f, prm), ("ab", a, b, prm),                              # shape change and back: plan replaced
           ("ab2", a, b, capi.FlowParams(kiters=3, alpha=8.0)), ("ab", a, b, prm)]   # parameter change
    for key, x, y, p in seq:
        u, v = capi.flow(x, y, p)
        assert np.array_equal(u, want[key][0]) and np.array_equal(v, want[key][1]), key
    capi.release_cache()
    capi.release_cache()                                                       # idempotent
    os.environ["OCTANE_VOF_CACHE"] = "0"
    try:
        u, v = capi.flow(a, b, prm)
    finally:
        del os.environ["OCTANE_VOF_CACHE"]
    assert np.array_equal(u, want["ab"][0]) and np.array_equal(v, want["ab"][1])


def test_concurrent_one_shot_calls(capi):
    """Two host threads inside octane_vof_run at once: one holds the cached plan, the other gets a private one."""
    a, b = synth.lattice_scene(300, 200, seed=8)
    prm = capi.FlowParams(kiters=3, liters=2)
    pl = capi.Plan(300, 200, 1, prm)
    want = pl.run_host(a, b)
    pl.close()
    out = [None] * 4

    def work(i):
        out[i] = capi.flow(a, b, prm)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for o in out:
        assert o is not None and np.array_equal(o[0], want[0]) and np.array_equal(o[1], want[1])
    capi.release_cache()


def test_two_lanes_on_their_own_streams(capi):
    """octane_vof_plan_solve on OCTANE_STREAM_OWN: two plans side by side on one GPU (the lanes of the batch workload),
    zero first guess by passing none, results where the caller wants them."""
    import torch
    dev = torch.device("cuda:0")
    n = 384
    pairs = [synth.lattice_scene(n, n, seed=s, device=dev) for s in (1, 2)]
    prm = capi.FlowParams(kiters=4, liters=2)
    want = []
    for a, b in pairs:
        pl = capi.Plan(n, n, 1, prm)
        want.append(pl.run_host(a.cpu().numpy(), b.cpu().numpy()))
        pl.close()
    plans = [capi.Plan(n, n, 1, prm) for _ in range(2)]
    outs = [(torch.full((n, n), 7.0, device=dev), torch.full((n, n), -7.0, device=dev)) for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(3):
        for i, (a, b) in enumerate(pairs):
            plans[i].solve_device(a.data_ptr(), b.data_ptr(), outs[i][0].data_ptr(), outs[i][1].data_ptr(), stream=capi.STREAM_OWN)
    for p in plans:
        p.wait()
    for i in range(2):
        assert np.array_equal(outs[i][0].cpu().numpy(), want[i][0]) and np.array_equal(outs[i][1].cpu().numpy(), want[i][1])
    # a first guess in separate buffers
    u0 = torch.full((n, n), 1.5, device=dev); v0 = torch.full((n, n), -0.5, device=dev)
    torch.cuda.synchronize()
    plans[0].solve_device(pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), outs[0][0].data_ptr(), outs[0][1].data_ptr(),
                          u0.data_ptr(), v0.data_ptr(), stream=capi.STREAM_OWN)
    plans[0].wait()
    ref = capi.Plan(n, n, 1, prm)
    wu, wv = ref.run_host(pairs[0][0].cpu().numpy(), pairs[0][1].cpu().numpy(), u0.cpu().numpy(), v0.cpu().numpy())
    ref.close()
    assert np.array_equal(outs[0][0].cpu().numpy(), wu) and np.array_equal(outs[0][1].cpu().numpy(), wv)
    assert float(u0[0, 0]) == 1.5                       # the first guess is not written
    for p in plans:
        p.close()


def test_one_shot_call_without_a_first_guess_equals_the_call_with_zeros(capi):
    """octane_vof_solve(u0 = v0 = NULL) -- what the shim of oct_optical_flow() calls when there is no -firstguess (ref
    src/oct_optical_flow.cc:38-48 zero-fills uPix / vPix) -- does not upload the zeros; the flow is the in-out call's bit for bit, and the
    output buffers' previous contents do not matter."""
    a, b = synth.lattice_scene(260, 190, seed=8)
    prm = capi.FlowParams(kiters=3, liters=2, cgiters=12)
    want = capi.flow(a, b, prm)
    u = np.full((190, 260), np.nan, np.float32)
    v = np.full((190, 260), 7.0, np.float32)
    capi.flow_into(a, b, u, v, prm)
    assert np.array_equal(u, want[0]) and np.array_equal(v, want[1])
    capi.release_cache()
