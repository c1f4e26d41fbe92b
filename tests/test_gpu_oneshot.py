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
