"""uv2pix and srsal: the optional steps before and after the flow solver.  GPU vs CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nav(cls, nx, ny):
    return cls(pph=35786023.0, req=6378137.0, rpol=6356752.31414, lam0=-1.308996939, xScale=5.6e-05, xOffset=-0.03,
               yScale=-5.6e-05, yOffset=0.09, g2xOffset=-0.03, g2yOffset=0.09, nx=nx, ny=ny)


def _latlon(capi, nx, ny):
    """lat/lon of the CONUS-like grid through navcal (donav=1), as oct_goesread produces them."""
    kw = dict(xScale=5.6e-05, xOffset=-0.03, yScale=-5.6e-05, yOffset=0.09, radScale=1.0, radOffset=0.0,
              rpol=6356752.31414, req=6378137.0, H=42164160.0, lam0=-1.308996939, fk1=1, fk2=1, bc1=0, bc2=1, kap1=1,
              maxin=1.0, minin=0.0, maxout=255.0, minout=0.0, cal=0, donav=1, minx=0, maxx=nx, miny=0, maxy=ny)
    x, y = np.arange(nx, dtype=np.int16), np.arange(ny, dtype=np.int16)
    out = capi.navcal(np.zeros((ny, nx), np.int16), x, y, capi.NavcalParams(**kw))
    return out[1], out[2], x, y


def test_uv2pix_matches_oracle_and_inverts_pix2uv(capi, oracle):
    nx, ny = 300, 200
    lat, lon, gx, gy = _latlon(capi, nx, ny)
    rng = np.random.RandomState(0)
    u = (rng.randn(ny, nx) * 10).astype(np.float32)          # m/s
    v = (rng.randn(ny, nx) * 10).astype(np.float32)
    nav, nav_o = _nav(capi.Nav, nx, ny), _nav(oracle.Nav, nx, ny)
    gu, gv = capi.uv2pix(nav, 0.0, 300.0, u, v, lat, lon, gx, gy)
    ou, ov = oracle.uv2pix(nav_o, 0.0, 300.0, u, v, lat, lon, gx, gy)
    for g, o in ((gu, ou), (gv, ov)):
        ulp = np.abs(g.view(np.int32).astype(np.int64) - o.view(np.int32))
        assert ulp.max() <= 2 and (ulp == 0).mean() > 0.99, (ulp.max(), (ulp == 0).mean())
    # 10 m/s for 300 s on a 2 km grid is 1.5 px; and pix2uv of the result gives the winds back (cm/s)
    ok = np.isfinite(lat) & np.isfinite(lon)                  # the frame's corner looks past the limb
    ok[:2] = ok[-2:] = False; ok[:, :2] = ok[:, -2:] = False
    assert ok.mean() > 0.9
    assert 0.5 < np.abs(gu[ok]).mean() < 3.0
    back = capi.pix2uv(nav, 0.0, 300.0, gu, gv)
    good = ok & (back[0] != 0)                                 # pix2uv zeroes pixels beyond its own limb threshold
    assert np.abs(back[0][good] / 100.0 - u[good]).mean() < 0.5 and np.abs(back[1][good] / 100.0 - v[good]).mean() < 0.5


def test_uv2pix_sector_moved_gives_zeros(capi):
    nx, ny = 20, 10
    nav = _nav(capi.Nav, nx, ny)
    nav.g2xOffset = -0.0301
    z = np.zeros((ny, nx), np.float32)
    gu, gv = capi.uv2pix(nav, 0.0, 60.0, z + 5, z - 5, z, z, np.arange(nx, dtype=np.int16), np.arange(ny, dtype=np.int16))
    assert not gu.any() and not gv.any()


@pytest.mark.parametrize("nx,ny", [(150, 90), (64, 37), (33, 70)])
def test_srsal_matches_oracle(capi, oracle, nx, ny):
    rng = np.random.RandomState(1)
    u = (rng.randn(ny, nx) * 2 + 3).astype(np.float32)
    v = (rng.randn(ny, nx) * 2 - 1).astype(np.float32)
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    cth = (8000 + 40 * np.sin(i / 9.0) * np.cos(j / 7.0) + rng.randn(ny, nx) * 5).astype(np.float32)   # guide image
    gu, gv = capi.srsal(u, v, cth)
    ou, ov = oracle.srsal(u, v, cth)
    for g, o in ((gu, ou), (gv, ov)):
        ulp = np.abs(g.view(np.int32).astype(np.int64) - o.view(np.int32))
        assert ulp.max() <= 1 and (ulp == 0).mean() > 0.995, (ulp.max(), (ulp == 0).mean())
    assert gu.std() < u.std() * 0.5        # it does smooth


def test_srsal_constant_guide_is_a_gaussian_blur(capi):
    """With a flat guide image the range weight is exp(0) = 1: a constant flow stays constant."""
    u = np.full((50, 60), 2.5, np.float32); v = np.full((50, 60), -1.25, np.float32)
    gu, gv = capi.srsal(u, v, np.zeros((50, 60), np.float32))
    assert np.allclose(gu, 2.5, atol=1e-6) and np.allclose(gv, -1.25, atol=1e-6)
