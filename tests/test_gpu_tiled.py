"""One frame solved as row bands (BASELINE.json configs[3], SURVEY.md 8e "spatial tiles of one frame"), through the
C-ABI (octane_vof_tiled_*).  Every test takes its devices from capi.band_devices(nbands): with two or more GPUs visible
the bands go round-robin over REAL devices (adjacent bands on different devices: in-kernel peer reads, LDS-DMA from the
neighbour's planes, cross-device events, peer copies over xGMI), no edit needed; a one-GPU box runs the bands as virtual
ranks on device 0 -- the kernels, the band bookkeeping, the halo / partial exchanges and their ordering are exactly the
ones several devices would run; only the copies are device-local instead of peer copies.

The banded solve computes the same global PCG as the plain plan -- same operator, same recurrences, same stop test
-- and differs only in the summation order of the dot products, so it must sit as close to the plain result as
two valid reduction orders do (~1e-6) and meet the same oracle bar."""
import numpy as np
import pytest

from conftest import rel_l2
from octane_amd import synth

pytestmark = pytest.mark.gpu

ORDER_BAR = 2e-5     # two reduction orders of the same solve (measured ~1e-6; sensitive parameter sets excluded)


def _plain(capi, a, b, prm, u0=None, v0=None):
    nc, ny, nx = (1,) + a.shape if a.ndim == 2 else a.shape
    pl = capi.Plan(nx, ny, nc, capi.FlowParams(**prm))
    u, v = pl.run_host(a, b, u0, v0)
    its = pl.last_iterations()
    pl.close()
    return u, v, its


def _tiled(capi, a, b, prm, nbands, min_band_pixels=1, u0=None, v0=None, devices=None):
    nc, ny, nx = (1,) + a.shape if a.ndim == 2 else a.shape
    tp = capi.TiledPlan(nx, ny, nc, capi.FlowParams(**prm), nbands=nbands, devices=devices or capi.band_devices(nbands),
                        min_band_pixels=min_band_pixels)
    u, v = tp.run_host(a, b, u0, v0)
    info = dict(its=tp.last_iterations(), banded=tp.banded_levels, copies=tp.last_copies(),
                rows=[[tp.band_rows(k, bnd) for bnd in range(nbands)] for k in range(prm.get("kiters", 4))])
    tp.close()
    return u, v, info


def test_one_band_is_the_plain_plan(capi):
    a, b = synth.lattice_scene(200, 150, seed=4)
    prm = dict(kiters=3)
    up, vp, ip = _plain(capi, a, b, prm)
    ut, vt, info = _tiled(capi, a, b, prm, 1)
    assert info["banded"] == 0 and info["copies"] == 0 and info["its"] == ip
    assert np.array_equal(up, ut) and np.array_equal(vp, vt)


def test_one_band_alone_on_its_device_runs_the_placement_trials_of_its_shared_allocation(capi):
    """Round 5: a band plan keeps the planes its neighbours read in an allocation of their own (vof_plan.hpp), and a band that has its
    device to itself -- every band of a real multi-GPU node -- times up to eight candidates of THAT allocation at creation.  On a
    one-GPU box only a single-band plan takes that path: 4.4 Mpixel (trials start at 4 Mi pixels), flow bit-equal to the plain plan's."""
    nx, ny = 2300, 1900
    a, b = synth.lattice_scene(nx, ny, seed=6)
    prm = dict(kiters=2, liters=1, cgiters=7)
    up, vp, ip = _plain(capi, a, b, prm)
    ut, vt, info = _tiled(capi, a, b, prm, 1)
    assert info["banded"] == 0 and info["its"] == ip
    assert np.array_equal(up, ut) and np.array_equal(vp, vt)


@pytest.mark.parametrize("nbands", [2, 3, 4])
def test_bands_match_plain_plan_and_oracle(capi, oracle, nbands):
    nx, ny = 300, 420
    a, b = synth.lattice_scene(nx, ny, seed=31)
    prm = dict(kiters=3, liters=2, cgiters=20)
    up, vp, ip = _plain(capi, a, b, prm)
    ut, vt, info = _tiled(capi, a, b, prm, nbands)
    assert np.isfinite(ut).all() and np.isfinite(vt).all()
    # levels are 75x105, 150x210, 300x420: a band needs 32 rows, so 4 bands leave the coarsest level replicated
    assert info["banded"] == (2 if nbands == 4 else 3)
    assert info["its"] == ip and info["copies"] > 0
    for k, lev in enumerate(info["rows"]):
        if lev[0][0]:                                    # banded: contiguous cover, aligned inner edges
            assert lev[0][1] == 0
            for lo, hi in zip(lev[:-1], lev[1:]):
                assert lo[2] == hi[1] and lo[2] % 32 == 0
    d = rel_l2(ut, vt, up, vp)
    assert d < ORDER_BAR, f"banded vs plain: {d:.3e}"
    uo, vo, io = oracle.flow(a, b, oracle.FlowParams(**prm), dot_threads=oracle.REF_GRID_THREADS)
    assert io == info["its"]
    assert rel_l2(ut, vt, uo, vo) < ORDER_BAR


def test_bands_with_first_guess_hint_and_two_channels(capi, oracle):
    """lambdac != 0 reads the decimated first guess in the assembly; two channels go through the channel-0
    decimation quirk; the first guess also seeds the coarsest level."""
    nx, ny = 260, 200
    a, b = synth.lattice_scene(nx, ny, seed=8, nchan=2)
    rng = np.random.RandomState(3)
    u0 = (2.0 + 0.2 * rng.randn(ny, nx)).astype(np.float32)
    v0 = (-1.0 + 0.2 * rng.randn(ny, nx)).astype(np.float32)
    prm = dict(kiters=2, liters=2, cgiters=15, lambdac=0.4)
    up, vp, ip = _plain(capi, a, b, prm, u0, v0)
    ut, vt, info = _tiled(capi, a, b, prm, 3, u0=u0, v0=v0)
    assert info["banded"] == 2 and info["its"] == ip
    assert rel_l2(ut, vt, up, vp) < ORDER_BAR
    uo, vo, _ = oracle.flow(a, b, oracle.FlowParams(**prm), u0=u0, v0=v0, dot_threads=oracle.REF_GRID_THREADS)
    assert rel_l2(ut, vt, uo, vo) < ORDER_BAR


def test_wide_frame_many_tiles_per_band(capi):
    """Several 128-column tiles per band row and several tile rows per band: every workgroup of the persistent
    kernels walks more than one tile, halo rows are stored by many workgroups."""
    nx, ny = 1700, 1100
    a, b = synth.lattice_scene(nx, ny, seed=12)
    prm = dict(kiters=2, liters=1, cgiters=12)
    up, vp, ip = _plain(capi, a, b, prm)
    ut, vt, info = _tiled(capi, a, b, prm, 4)
    assert info["banded"] == 2 and info["its"] == ip
    assert rel_l2(ut, vt, up, vp) < ORDER_BAR


def test_small_levels_stay_replicated_by_default_threshold(capi):
    """With the default threshold (4 Mpixel since round 4, 12 before) nothing at 2000 x 1500 is banded: every band solves every level
    and band 0's result is the plain plan's, bit for bit."""
    nx, ny = 2000, 1500
    a, b = synth.lattice_scene(nx, ny, seed=5)
    prm = dict(kiters=3, liters=1, cgiters=6)
    up, vp, _ = _plain(capi, a, b, prm)
    ut, vt, info = _tiled(capi, a, b, prm, 2, min_band_pixels=0)
    assert info["banded"] == 0 and info["copies"] == 0
    assert np.array_equal(up, ut) and np.array_equal(vp, vt)


def test_mixed_replicated_and_banded_levels(capi):
    """Threshold between the levels: the coarse ones replicated, the two finest banded -- the hand-over (every band
    holds the whole coarse flow; the banded level all-gathers its bands at its end) in both directions."""
    nx, ny = 1200, 900
    a, b = synth.lattice_scene(nx, ny, seed=21)
    prm = dict(kiters=4, liters=1, cgiters=10)
    up, vp, ip = _plain(capi, a, b, prm)
    ut, vt, info = _tiled(capi, a, b, prm, 3, min_band_pixels=200_000)   # 150x113, 300x225 replicated
    assert info["banded"] == 2 and info["its"] == ip
    assert rel_l2(ut, vt, up, vp) < ORDER_BAR


def test_device_resident_inputs_and_repeated_solves(capi):
    """load (dense device buffers) / solve / fetch, twice on the same plan: same bits as the host-buffer run."""
    import torch
    nx, ny = 640, 512
    a, b = synth.lattice_scene(nx, ny, seed=2)
    prm = dict(kiters=3, liters=1, cgiters=10)
    uh, vh, info = _tiled(capi, a, b, prm, 2)
    dev = torch.device("cuda:0")
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    tu, tv = torch.zeros(ny, nx, device=dev), torch.zeros(ny, nx, device=dev)
    ou, ov = torch.empty(ny, nx, device=dev), torch.empty(ny, nx, device=dev)
    torch.cuda.synchronize()
    tp = capi.TiledPlan(nx, ny, 1, capi.FlowParams(**prm), nbands=2, devices=capi.band_devices(2), min_band_pixels=1)
    tp.load_device(ta.data_ptr(), tb.data_ptr(), tu.data_ptr(), tv.data_ptr())
    for _ in range(2):
        tp.solve()
        tp.fetch_device(ou.data_ptr(), ov.data_ptr())
        assert tp.last_iterations() == info["its"]
        assert np.array_equal(ou.cpu().numpy(), uh) and np.array_equal(ov.cpu().numpy(), vh)
    tp.close()


def test_full_disk_quarter_scale_four_bands(capi):
    """5424 x 5424 (the second level of a 10848^2 full-disk pyramid; 29 Mpixel, above the default banding
    threshold), four bands, default threshold (4 Mpixel since round 4): the finest level and the 2712^2 one (7.4 Mpixel: bands of
    1.8 Mpixel on the stored-q kernel) are banded, the rest replicated.  Checked against the plain plan on the same inputs."""
    import torch
    n = 5424
    dev = torch.device("cuda:0")
    a, b = synth.lattice_scene(n, n, seed=20240613 + 3, device=dev)
    z = torch.zeros(n, n, device=dev)
    ou, ov = torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)     # in-out: zero first guess
    prm = capi.FlowParams(kiters=6, liters=1, cgiters=8)
    torch.cuda.synchronize()
    pl = capi.Plan(n, n, 1, prm)
    pl.run_device(a.data_ptr(), b.data_ptr(), ou.data_ptr(), ov.data_ptr())
    torch.cuda.synchronize()
    up, vp, ip = ou.cpu().numpy(), ov.cpu().numpy(), pl.last_iterations()
    pl.close()
    tp = capi.TiledPlan(n, n, 1, prm, nbands=4, devices=capi.band_devices(4))
    assert tp.banded_levels == 2
    banded, y0, y1 = tp.band_rows(5, 2)
    assert banded and (y0, y1) == (2720, 4064)
    tp.load_device(a.data_ptr(), b.data_ptr(), z.data_ptr(), z.data_ptr())
    tp.solve()
    tp.fetch_device(ou.data_ptr(), ov.data_ptr())
    ut, vt = ou.cpu().numpy(), ov.cpu().numpy()
    assert tp.last_iterations() == ip
    tp.close()
    assert np.isfinite(ut).all() and np.isfinite(vt).all()
    assert rel_l2(ut, vt, up, vp) < ORDER_BAR


def test_config3_full_disk_10848_four_bands_equals_plain_plan(capi):
    """BASELINE.json configs[3] at full size and full work: a 10848 x 10848 full-disk pair, SURVEY 8d's R1 parameters
    (kiters 8, liters 3, cgiters 30 -> 2160 PCG iterations), split four ways through octane_vof_tiled_* (virtual bands
    on the one device a test box has: same kernels, same band bookkeeping, same exchange protocol as four devices),
    against the plain plan on the same inputs.  u / v are in-out with a zero first guess (ref .cu:1213).  The banded
    solve is the same global PCG with another grouping of the fp64 partial sums: expected bit-identical, asserted
    <= 2e-5 relative L2 with equal iteration counts."""
    import torch
    n = 10848
    dev = torch.device("cuda:0")
    # seed: with 20240616 the coarsest (85 x 85) level of this pyramid runs away at R1's iteration counts (flows of 130 px;
    # plain and banded solves agree on that bit for bit, and 7 levels or liters 1 / cgiters 10 recover the truth) -- the
    # reference's scheme has no safeguard against an aliased coarse level.  This scene converges, so the truth is checked too.
    a, b = synth.lattice_scene(n, n, seed=20240615, device=dev)
    z = torch.zeros(n, n, device=dev)
    ou, ov = torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev)
    prm = capi.FlowParams(kiters=8, liters=3, cgiters=30)
    torch.cuda.synchronize()
    pl = capi.Plan(n, n, 1, prm)
    pl.run_device(a.data_ptr(), b.data_ptr(), ou.data_ptr(), ov.data_ptr())
    torch.cuda.synchronize()
    up, vp, ip = ou.clone(), ov.clone(), pl.last_iterations()
    pl.close()
    assert ip == 8 * 3 * 3 * 30
    tp = capi.TiledPlan(n, n, 1, prm, nbands=4, devices=capi.band_devices(4))
    nbanded = tp.banded_levels
    assert nbanded == 3                                   # 10848^2, 5424^2 and 2712^2 are above the default 4 Mpixel threshold (round 4; 12 before)
    tp.load_device(a.data_ptr(), b.data_ptr(), z.data_ptr(), z.data_ptr())
    tp.solve()
    tp.fetch_device(ou.data_ptr(), ov.data_ptr())
    torch.cuda.synchronize()
    its = tp.last_iterations()
    tp.close()
    assert bool(torch.isfinite(ou).all()) and bool(torch.isfinite(ov).all())
    num = ((ou - up).double() ** 2).sum() + ((ov - vp).double() ** 2).sum()
    den = (up.double() ** 2).sum() + (vp.double() ** 2).sum()
    d = float(torch.sqrt(num / den))
    ndiff = int((ou != up).sum() + (ov != vp).sum())
    tu, tv = synth.true_lattice_flow(n, n, xp=torch)
    m = n // 8
    eu = float((ou[m:-m:16, m:-m:16].double().cpu() - tu[m:-m:16, m:-m:16]).abs().mean())
    ev = float((ov[m:-m:16, m:-m:16].double().cpu() - tv[m:-m:16, m:-m:16]).abs().mean())
    print(f"PARITY case=config3_full_disk_4_bands {n}x{n}: banded vs plain relL2 {d:.3e} ({ndiff} values differ), "
          f"iterations plain/banded {ip}/{its}, banded levels {nbanded}, mean |flow - truth| {eu:.4f}, {ev:.4f} px")
    assert its == ip
    assert d <= ORDER_BAR
    assert eu < 0.05 and ev < 0.05


@pytest.mark.parametrize("nbands", [2, 3])
def test_bands_above_four_megapixels_recompute_q(capi, oracle, nbands):
    """Bands of 2 * 2^20 pixels and more run the q-recomputing fused kernel: nothing is stored on halo rows, a band reads r on
    the row beyond its edge and p on the two rows beyond it from the neighbour's planes.  The coarser level of the same
    plan keeps the stored-q form, so both exchange schemes run in one solve.  Checked against the plain plan (same
    kernels, one band) and, on a single level, against the oracle."""
    nx = 2432
    ny = 1792 * nbands                       # 4.36 Mpixel per band at the finest level
    a, b = synth.lattice_scene(nx, ny, seed=61)
    prm = dict(kiters=2, liters=1, cgiters=9)
    up, vp, ip = _plain(capi, a, b, prm)
    ut, vt, info = _tiled(capi, a, b, prm, nbands)
    assert info["banded"] == 2 and info["its"] == ip
    assert np.isfinite(ut).all()
    assert rel_l2(ut, vt, up, vp) < ORDER_BAR
    if nbands == 2:
        prm1 = dict(kiters=1, liters=1, cgiters=5)
        ut, vt, info = _tiled(capi, a, b, prm1, nbands)
        uo, vo, _ = oracle.flow(a, b, oracle.FlowParams(**prm1), flavour="omp", dot_threads=oracle.REF_GRID_THREADS)
        assert rel_l2(ut, vt, uo, vo) < ORDER_BAR


@pytest.mark.parametrize("nx,ny,nbands,prm", [
    (300, 420, 3, dict(kiters=3, liters=2, cgiters=20)),          # small bands: the stored-q form, one row of q per edge
    (2432, 3584, 2, dict(kiters=2, liters=1, cgiters=9)),         # bands of 4.4 Mpixel: the q-recomputing form, r and two rows of p per edge, wy
    (1700, 1100, 4, dict(kiters=2, liters=1, cgiters=12)),
])
def test_copy_transport_equals_in_place_reads_bit_for_bit(capi, nx, ny, nbands, prm):
    """OCTANE_TILED_TRANSPORT=copy: the rows beyond a band's edges and the other bands' partial sums are pulled into local memory
    by stream-ordered copies and every kernel reads local memory only -- the fallback that tells a protocol error from a
    visibility problem of in-kernel peer reads.  Same arithmetic, same fold order: the flow has to be the in-place transport's
    bit for bit, with more copies issued."""
    import os
    a, b = synth.lattice_scene(nx, ny, seed=29)
    ui, vi, info_i = _tiled(capi, a, b, prm, nbands)
    os.environ["OCTANE_TILED_TRANSPORT"] = "copy"
    try:
        uc, vc, info_c = _tiled(capi, a, b, prm, nbands)
    finally:
        del os.environ["OCTANE_TILED_TRANSPORT"]
    print(f"TRANSPORT {nx}x{ny} {nbands} bands: copies in place {info_i['copies']}, copy transport {info_c['copies']}; "
          f"{int((ui != uc).sum() + (vi != vc).sum())} values differ")
    assert info_c["banded"] == info_i["banded"] >= 1 and info_c["its"] == info_i["its"]
    assert info_c["copies"] > info_i["copies"]
    assert np.array_equal(ui, uc) and np.array_equal(vi, vc)


def test_two_real_peers_in_place_and_copy_transports_agree(capi):
    """Needs two GPUs (skips on a one-GPU box): two bands on devices 0 and 1, both transports -- the consuming kernel
    dereferencing the neighbour's memory over xGMI (peer global_load_lds included: bands of 4.4 Mpixel run the LDS-DMA
    q-recomputing kernel) and the stream-ordered peer copies -- against each other bit for bit and against the plain
    plan on device 0; then the same bands as virtual bands on device 0: real peers must not change a bit."""
    import os
    if capi.lib().octane_device_count() < 2:
        pytest.skip("one GPU visible: the row bands of this box are virtual bands (covered by every other test here)")
    nx, ny, prm = 2432, 3584, dict(kiters=2, liters=1, cgiters=9)
    a, b = synth.lattice_scene(nx, ny, seed=29)
    up, vp, ip = _plain(capi, a, b, prm)
    ui, vi, info_i = _tiled(capi, a, b, prm, 2, devices=[0, 1])
    os.environ["OCTANE_TILED_TRANSPORT"] = "copy"
    try:
        uc, vc, info_c = _tiled(capi, a, b, prm, 2, devices=[0, 1])
    finally:
        del os.environ["OCTANE_TILED_TRANSPORT"]
    uv, vv, info_v = _tiled(capi, a, b, prm, 2, devices=[0, 0])
    print(f"REAL-PEERS {nx}x{ny}: in place vs copy {int((ui != uc).sum() + (vi != vc).sum())} values differ, "
          f"real vs virtual bands {int((ui != uv).sum() + (vi != vv).sum())}, vs plain relL2 {rel_l2(ui, vi, up, vp):.2e}, "
          f"iterations {info_i['its']}/{info_c['its']}/{info_v['its']}/{ip}")
    assert info_i["banded"] == info_c["banded"] == 2 and info_i["its"] == info_c["its"] == info_v["its"] == ip
    assert np.array_equal(ui, uc) and np.array_equal(vi, vc)
    assert np.array_equal(ui, uv) and np.array_equal(vi, vv)
    assert rel_l2(ui, vi, up, vp) < ORDER_BAR


def test_first_contact_selfcheck_runs_at_creation_and_keeps_in_place_reads(capi):
    """octane_vof_tiled_create solves a small two-level frame on the plan's own devices under each candidate
    transport and keeps the first that reproduces the plain plan (include/octane_vof.h).  On a box where in-place reads work that is
    the first candidate: small bands check the stored-q kernel's peer reads, bands of 2 Mpixel and more also the LDS-DMA from the
    neighbouring band (the self-check's frame is then 2048 x 1024 per band)."""
    small = capi.TiledPlan(300, 420, 1, capi.FlowParams(kiters=2), nbands=3, devices=capi.band_devices(3), min_band_pixels=1)
    large = capi.TiledPlan(2432, 3584, 1, capi.FlowParams(kiters=2, liters=1, cgiters=5), nbands=2, devices=capi.band_devices(2), min_band_pixels=1)
    try:
        for tp, nb in ((small, 3), (large, 2)):
            info = tp.transport_info()
            print("SELFCHECK", info)
            assert info["transport_used"] == "inplace" and info["q_dma"] and not info["forced"]
            assert info["selfcheck"] == "first candidate passed" and info["candidates_tried"] == 1
            assert info["bands"] == nb and info["peer_ok"] and 0 <= info["check_rel_l2"][0] <= ORDER_BAR
    finally:
        small.close(); large.close()


def test_forced_transport_and_switched_off_selfcheck_are_reported(capi, monkeypatch):
    monkeypatch.setenv("OCTANE_TILED_TRANSPORT", "copy")
    tp = capi.TiledPlan(300, 420, 1, capi.FlowParams(kiters=2), nbands=2, devices=capi.band_devices(2), min_band_pixels=1)
    info = tp.transport_info(); tp.close()
    assert info["transport_used"] == "copy" and info["forced"] and info["selfcheck"] == "not run"
    monkeypatch.delenv("OCTANE_TILED_TRANSPORT")
    monkeypatch.setenv("OCTANE_TILED_SELFCHECK", "0")
    tp = capi.TiledPlan(300, 420, 1, capi.FlowParams(kiters=2), nbands=2, devices=capi.band_devices(2), min_band_pixels=1)
    info = tp.transport_info(); tp.close()
    assert info["transport_used"] == "inplace" and not info["forced"] and info["selfcheck"] == "not run"
    monkeypatch.setenv("OCTANE_TILED_TRANSPORT", "collective")
    with pytest.raises(capi.OctaneError):          # the collective transport is the process form's (octane_vof_mp_*)
        capi.TiledPlan(300, 420, 1, capi.FlowParams(kiters=2), nbands=2, devices=capi.band_devices(2), min_band_pixels=1)


def _drill(capi, bits, nx, ny, nbands, prm=(2, 1, 6)):
    import json
    import os
    import subprocess
    import sys
    if not os.path.exists(capi.DIAG_LIB_PATH):
        pytest.skip("diagnostic library not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tiled_diag_worker.py"), str(nx), str(ny), str(nbands), *map(str, prm)],
                       env=dict(os.environ, OCTANE_LIB=capi.DIAG_LIB_PATH, OCTANE_TEST_BREAK_TRANSPORT=str(bits)),
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("TILED_RESULT ")]
    assert r.returncode == 0 and line, r.stdout + r.stderr
    return json.loads(line[0][len("TILED_RESULT "):]), r.stderr


def test_selfcheck_downgrades_in_place_reads_to_copies_when_they_return_wrong_sums(capi):
    """The drill of the automatic downgrade (diagnostic library's fault hook, bit 0: a band folds its own partial block in place of
    another band's -- what stale peer reads would look like): the self-check's first candidate fails, the copy transport passes, the
    plan uses it, says so once on stderr, and the solve of the real frame is right."""
    res, err = _drill(capi, 1, 300, 420, 3)
    print("DRILL-1", res, err.strip().splitlines()[-1:] )
    assert res["created"] and res["info"]["transport_used"] == "copy" and res["info"]["selfcheck"] == "downgraded"
    assert res["info"]["candidates_tried"] == 2 and res["info"]["check_rel_l2"][0] > ORDER_BAR >= res["info"]["check_rel_l2"][1]
    assert res["rel_l2"] < ORDER_BAR and res["its"] == res["its_plain"] and res["banded"] >= 1
    assert "self-check" in err and "copy" in err


def test_selfcheck_drops_lds_dma_from_the_neighbour_before_it_gives_up_in_place_reads(capi):
    """Fault hook bit 1: the rows the LDS-DMA kernel fetches from beyond a band's lower edge come from the wrong place.  Bands of
    2 Mpixel: the self-check runs its large frame, candidate 1 (in place, LDS-DMA) fails, candidate 2 -- in place with the
    register-staged q-recomputing kernel -- passes; the real frame is then solved that way and is right."""
    res, err = _drill(capi, 2, 2048, 2112, 2, prm=(1, 1, 6))
    print("DRILL-2", res, err.strip().splitlines()[-1:])
    assert res["created"] and res["info"]["transport_used"] == "inplace" and not res["info"]["q_dma"]
    assert res["info"]["selfcheck"] == "downgraded" and res["info"]["candidates_tried"] == 2
    assert res["rel_l2"] < ORDER_BAR and res["its"] == res["its_plain"]


def test_selfcheck_that_finds_no_working_transport_fails_creation_with_a_reason(capi):
    """Bits 0 and 2: in-place reads AND the copy transport are broken; there is no third transport in the thread form.  Creation must
    fail (not return a plan that solves wrongly) and the error must say which candidates were tried."""
    res, err = _drill(capi, 5, 300, 420, 2)
    assert not res["created"] and "self-check" in res["msg"] and "inplace" in res["msg"] and "copy" in res["msg"]
