"""Patch-matching flow (-sosm) on the GPU, through the C-ABI: bit-exact against outputs of the reference itself
(tests/golden/ref_sosm.npz, generated from the reference's own C++ compiled where it lies) and against the pinned
CPU restatement on larger frames."""
import os

import numpy as np
import pytest

from octane_amd import synth
from test_oracle_sosm import CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_sosm.npz"))


@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_reference_outputs_bitwise(capi, golden, name):
    rad, srad = (int(x) for x in golden[name + "_prm"])
    u0 = golden[name + "_u0"] if name + "_u0" in golden else None
    v0 = golden[name + "_v0"] if name + "_v0" in golden else None
    u, v = capi.sosm(golden[name + "_a"], golden[name + "_b"], rad, srad, u0, v0)
    assert np.array_equal(u, golden[name + "_u"]), f"u: {(u != golden[name + '_u']).sum()} pixels differ"
    assert np.array_equal(v, golden[name + "_v"]), f"v: {(v != golden[name + '_v']).sum()} pixels differ"


@pytest.mark.parametrize("nx,ny,rad,srad,guess", [(1000, 700, 2, 2, False), (701, 333, 2, 2, True), (400, 300, 3, 4, True),
                                                    (257, 129, 1, 1, False)])
def test_gpu_matches_pinned_restatement(capi, oracle, nx, ny, rad, srad, guess):
    a, b = (x[0] for x in synth.lattice_scene(nx, ny, seed=nx + ny))
    u0 = v0 = None
    if guess:
        rng = np.random.RandomState(3)
        u0 = (2.5 + 2 * rng.randn(ny, nx)).astype(np.float32)
        v0 = (-1.0 + 2 * rng.randn(ny, nx)).astype(np.float32)
    ug, vg = capi.sosm(a, b, rad, srad, u0, v0)
    uo, vo = oracle.sosm(a, b, rad, srad, u0, v0, flavour="omp")
    assert np.array_equal(ug, uo), f"u: {(ug != uo).sum()} of {ug.size} differ"
    assert np.array_equal(vg, vo), f"v: {(vg != vo).sum()} of {vg.size} differ"


def test_sosm_tracks_the_lattice_scene_within_its_search_window(capi):
    """Physical sanity: on the textured lattice scene the winning offsets follow the true displacement where it lies
    inside the +-2 pixel search window (v ~ -1 +- 1), and saturate at the window edge where it does not (u ~ 2.5 +- 1.5)."""
    nx, ny = 600, 400
    a, b = (x[0] for x in synth.lattice_scene(nx, ny, seed=9))
    u, v = capi.sosm(a, b)
    tu, tv = synth.true_lattice_flow(nx, ny)
    m = (slice(40, -40), slice(40, -40))
    inside = (np.abs(tv[m]) < 1.7)
    assert np.median(np.abs(v[m][inside] - tv[m][inside])) < 0.25
    assert np.median(np.abs(u[m] - np.clip(tu[m], -2.0, 2.0))) < 0.6
