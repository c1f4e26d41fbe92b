"""Known-answer and self-consistency tests of the navigation-side oracles (pix2uv, navcal, uv2pix, srsal) on CPU:
the forward fixed-grid projection of uv2pix inverts navcal's pixel -> lat/lon, pix2uv inverts uv2pix, the
normalisation and the limb taper have closed forms, and the bilateral filter preserves constants."""
import numpy as np
import pytest


def _grid(oracle, nx, ny, xoff=-0.03, yoff=0.09):
    kw = dict(xScale=5.6e-05, xOffset=xoff, yScale=-5.6e-05, yOffset=yoff, radScale=0.05, radOffset=-1.5,
              rpol=6356752.31414, req=6378137.0, H=42164160.0, lam0=-1.308996939, fk1=10803.3, fk2=1392.74, bc1=0.0755,
              bc2=0.99975, kap1=0.0015839, maxin=185.5699, minin=-1.6443, maxout=255.0, minout=0.0, cal=0, donav=1,
              minx=0, maxx=nx, miny=0, maxy=ny)
    return kw, np.arange(nx, dtype=np.int16), np.arange(ny, dtype=np.int16)


def _nav(oracle, nx, ny, xoff=-0.03, yoff=0.09):
    return oracle.Nav(pph=35786023.0, req=6378137.0, rpol=6356752.31414, lam0=-1.308996939, xScale=5.6e-05, xOffset=xoff,
                      yScale=-5.6e-05, yOffset=yoff, g2xOffset=xoff, g2yOffset=yoff, nx=nx, ny=ny)


def test_navcal_normalisation_closed_form(oracle):
    nx, ny = 40, 30
    kw, x, y = _grid(oracle, nx, ny)
    data2 = np.random.RandomState(0).randint(0, 4000, (ny, nx)).astype(np.int16)
    data3, lat, lon, d2s, xs, ys = oracle.navcal(data2, x, y, oracle.NavcalParams(**kw))
    rad = np.float32(data2) * np.float32(0.05) + np.float32(-1.5)
    want = (rad.astype(np.float64) - np.float32(-1.6443)) / (np.float32(185.5699) - np.float32(-1.6443)) * 255.0
    np.testing.assert_allclose(data3, want, rtol=1e-6)
    assert np.array_equal(d2s, data2) and np.array_equal(xs, x) and np.array_equal(ys, y)
    assert 20 < lat.mean() < 40 and -100 < lon.mean() < -80           # CONUS-ish for this grid


def test_navcal_limb_taper(oracle):
    """sds < 0.021 -> 1, >= 0.0212 -> 0, linear in between (ref src/oct_navcal_cuda.cu:81-91)."""
    nx, ny = 400, 1
    kw, x, y = _grid(oracle, nx, ny, xoff=0.1440, yoff=0.0)
    kw.update(xScale=1e-05, donav=0)
    data2 = np.full((ny, nx), 2000, np.int16)
    data3 = oracle.navcal(data2, x, y, oracle.NavcalParams(**kw))[0][0]
    xv = (x.astype(np.float32) * np.float32(1e-05) + np.float32(0.1440)).astype(np.float64)
    sds = xv * xv
    full = data3[sds < 0.021]
    assert np.allclose(full, full[0]) and full[0] > 100
    assert not data3[sds >= 0.0212].any()
    mid = (sds >= 0.021) & (sds < 0.0212)
    assert mid.sum() > 20 and (np.diff(data3[mid]) < 0).all()         # monotone ramp down


def test_uv2pix_inverts_navcal_and_pix2uv_inverts_uv2pix(oracle):
    nx, ny = 120, 90
    kw, x, y = _grid(oracle, nx, ny)
    lat, lon = oracle.navcal(np.zeros((ny, nx), np.int16), x, y, oracle.NavcalParams(**kw))[1:3]
    nav = _nav(oracle, nx, ny)
    z = np.zeros((ny, nx), np.float32)
    u0, v0 = oracle.uv2pix(nav, 0.0, 300.0, z, z, lat, lon, x, y)      # zero wind: stays on its pixel
    assert np.abs(u0).max() < 2e-2 and np.abs(v0).max() < 2e-2         # float lat/lon is worth ~1 m of the 2 km pixel
    rng = np.random.RandomState(2)
    u = (rng.randn(ny, nx) * 8).astype(np.float32); v = (rng.randn(ny, nx) * 8).astype(np.float32)
    pu, pv = oracle.uv2pix(nav, 0.0, 300.0, u, v, lat, lon, x, y)
    ur, vr = oracle.pix2uv(nav, 0.0, 300.0, pu, pv)[:2]
    assert np.abs(ur / 100.0 - u).mean() < 0.3 and np.abs(vr / 100.0 - v).mean() < 0.3      # m/s


def test_uv2pix_guard_and_pix2uv_guard(oracle):
    nx, ny = 8, 6
    nav = _nav(oracle, nx, ny)
    nav.g2yOffset = 0.0901
    z = np.ones((ny, nx), np.float32)
    u, v = oracle.uv2pix(nav, 0.0, 60.0, z, z, z, z, np.arange(nx, dtype=np.int16), np.arange(ny, dtype=np.int16))
    assert not u.any() and not v.any()
    out = oracle.pix2uv(nav, 0.0, 60.0, z, z)
    assert out[5] == 1 and not out[0].any()


def test_srsal_known_answers(oracle):
    ny, nx = 45, 50
    c = np.zeros((ny, nx), np.float32)
    u = np.full((ny, nx), 3.0, np.float32); v = np.full((ny, nx), -2.0, np.float32)
    uo, vo = oracle.srsal(u, v, c)
    assert np.allclose(uo, 3.0, atol=1e-6) and np.allclose(vo, -2.0, atol=1e-6)
    # a guide-image edge of 200 units (10 sigma of the range kernel) keeps the two sides from mixing
    c[:, 25:] = 200.0
    u[:, 25:] = 7.0
    uo, _ = oracle.srsal(u, v, c)
    assert np.allclose(uo[:, :25], 3.0, atol=1e-6) and np.allclose(uo[:, 25:], 7.0, atol=1e-6)
    # without the edge in the guide it is a plain Gaussian blur across the step
    uo2, _ = oracle.srsal(u, v, np.zeros_like(c))
    assert 3.0 < uo2[20, 24] < uo2[20, 25] < 7.0


def test_projected_navigation_known_answers(oracle):
    """Mercator: lon = x/R + lon0 and the Gudermannian latitude; polar (orthographic about lat1 = 90 deg): the pole
    itself, the great-circle distance c = asin(rho/R) => lat = 90 - c, and no inverse outside the disc."""
    R = 6371228.0
    x = np.array([-300, 0, 300], np.int16)
    y = np.array([0, 250, -250], np.int16)
    d = np.zeros((3, 3), np.float32)
    m = oracle.proj_navcal(d, x, y, oracle.ProjNavcalParams(xScale=10000.0, xOffset=0.0, yScale=10000.0, yOffset=0.0, lon0=-100.0,
                                                             lat1=0.0, R=R, donav=1, mode=2, minx=0, maxx=3, miny=0, maxy=3))
    xv, yv = x.astype(np.float64) * 1e4, y.astype(np.float64) * 1e4
    assert np.allclose(m[2][0], np.degrees(xv / R) - 100.0, atol=2e-5)
    assert np.allclose(m[1][:, 0], np.degrees(np.pi / 2 - 2 * np.arctan(np.exp(-yv / R))), atol=2e-5)
    p = oracle.proj_navcal(d, x, y, oracle.ProjNavcalParams(xScale=10000.0, xOffset=0.0, yScale=10000.0, yOffset=0.0, lon0=-45.0,
                                                             lat1=90.0, R=R, donav=1, mode=1, minx=0, maxx=3, miny=0, maxy=3))
    lat, lon = p[1], p[2]
    assert abs(lat[0, 1] - 90.0) < 1e-4                                    # x = y = 0: the pole (rho == 0 branch)
    rho = np.hypot(xv[None, :], yv[:, None])
    want = 90.0 - np.degrees(np.arcsin(rho / R))
    off = rho > 0
    assert np.allclose(lat[off], want[off], atol=2e-3)                     # float lat1 = pi/2 costs ~1e-3 deg
    far = oracle.proj_navcal(d, np.array([700, 0, 0], np.int16), y, oracle.ProjNavcalParams(
        xScale=10000.0, xOffset=0.0, yScale=10000.0, yOffset=0.0, lon0=-45.0, lat1=90.0, R=R, donav=1, mode=1, minx=0, maxx=3, miny=0, maxy=3))
    assert np.isnan(far[1][0, 0])                                          # 7000 km from the pole on the plane: off the disc
    assert not p[3].any() and np.array_equal(p[4], x) and np.array_equal(p[5], y)


def test_pix2uv_fma_site_switches(oracle):
    """Round 5: every multiply-add of the navigation that nvcc's -fmad=true may fuse is an explicit switch of the oracle
    (oracle/pix2uv_oracle.c).  No site fused = the strict build's shorts (what every earlier test pins); the two float sites of the base
    position fused = what the compiler-contracted flavour gives (gcc fuses the double sites too: they move nothing here); each float site
    alone moves some shorts, by 1 cm/s; the eleven double sites together move none on this frame."""
    nx, ny = 300, 200
    nav = _nav(oracle, nx, ny)
    rng = np.random.RandomState(4)
    u = (rng.randn(ny, nx) * 3).astype(np.float32)
    v = (rng.randn(ny, nx) * 3).astype(np.float32)
    assert oracle.lib().oct_oracle_pix2uv_nsites() == len(oracle.P2U_SITES)
    base = oracle.pix2uv(nav, 0.0, 300.0, u, v)
    same = oracle.pix2uv(nav, 0.0, 300.0, u, v, sites=0)
    fl = oracle.pix2uv(nav, 0.0, 300.0, u, v, sites=oracle.P2U_FLOAT_SITES)
    al = oracle.pix2uv(nav, 0.0, 300.0, u, v, sites=oracle.P2U_ALL_SITES)
    dbl = oracle.pix2uv(nav, 0.0, 300.0, u, v, sites=oracle.P2U_ALL_SITES & ~oracle.P2U_FLOAT_SITES)
    gcc = oracle.pix2uv(nav, 0.0, 300.0, u, v, flavour="fma")

    def diff(a, b):
        return int((a[0] != b[0]).sum() + (a[1] != b[1]).sum())
    assert diff(same, base) == 0 and diff(dbl, base) == 0
    assert 0 < diff(fl, base) < 0.06 * 2 * nx * ny
    assert diff(al, fl) == 0
    from conftest import SANITIZE
    if not SANITIZE:        # (which products the compiler fuses by itself depends on its flags: the -O1 sanitizer build leaves some of the float ones alone)
        assert diff(gcc, fl) == 0
    for k in (0, 1):
        one = oracle.pix2uv(nav, 0.0, 300.0, u, v, sites=1 << k)
        assert diff(one, base) > 0
        assert max(np.abs(one[0].astype(int) - base[0]).max(), np.abs(one[1].astype(int) - base[1]).max()) == 1
    assert np.array_equal(oracle.pix2uv(nav, 0.0, 300.0, u, v)[0], base[0])        # the switch does not stick
