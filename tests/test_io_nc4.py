"""The file layer of the reference's CLI on nc4lite (HDF5): GOES-R L1b in, outfile.nc out.
"parity unpinned": the reference reads/writes through netcdf-cxx4, which does not exist in this image, and ships no
files.  CPU tests: what is written can be read back, carries the names / types / attributes the reference's writer
emits and the NetCDF-4 dimension-scale conventions.  GPU test: the `octane` command line end to end against the same
steps made through the library."""
import os
import subprocess

import numpy as np
import pytest

from octane_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "build")
HDF5_ROOT = os.environ.get("HDF5_ROOT", "/opt/conda")
LIBD = os.path.join(ROOT, "octane_amd")

# the reader's lam0: the file's float attribute times a double DTOR, rounded back to float (ref fr:179-182)
LAM0 = np.float32(np.float64(np.float32(-75.0)) * (3.14159265359 / 180.0))

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(HDF5_ROOT, "include", "hdf5_hl.h")),
                                reason="no HDF5 with the high-level library on this machine")


@pytest.fixture(scope="module")
def io_demo(capi):
    capi.lib()
    csrc = os.path.join(ROOT, "octane_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-s", "-f", "Makefile.host"])
    subprocess.check_call(["make", "-C", csrc, "-s", "-f", "Makefile.io"])
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "io_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I", os.path.join(csrc, "io"),
                           os.path.join(ROOT, "tests", "cpp", "io_demo.cpp"), "-o", out, "-L", LIBD, "-loctane_io", "-loctane_host",
                           "-loctane_vof", "-Wl,-rpath," + LIBD, "-Wl,-rpath-link," + os.path.join(HDF5_ROOT, "lib")])
    return out


def _dump(io_demo, path):
    out = subprocess.check_output([io_demo, "--dump", str(path)]).decode()
    d = {}
    for line in out.strip().splitlines():
        name, typ, shape, atts = line.split("|", 3)
        d[name] = dict(type=typ, shape=shape, atts=dict(a.split("=", 1) for a in atts.split(";") if "=" in a))
    return d


def _read(io_demo, path, var, typ, tmp):
    out = tmp / (var + ".bin")
    subprocess.check_call([io_demo, "--read", str(path), var, typ, str(out)])
    return np.fromfile(out, dtype={"short": np.int16, "int": np.int32, "float": np.float32, "double": np.float64}[typ])


def _counts(nx, ny, seed):
    a, b = synth.lattice_scene(nx, ny, seed=seed)
    return [np.clip(np.round(x[0] * 14.0 + 200.0), 0, 4094).astype(np.int16) for x in (a, b)]     # 12-bit C13-like counts


def test_goes_lookalike_round_trip(io_demo, tmp_path):
    nx, ny = 96, 64
    c1, _ = _counts(nx, ny, 3)
    raw = tmp_path / "rad.bin"
    c1.tofile(raw)
    f = tmp_path / "g1.nc"
    subprocess.check_call([io_demo, "--make-goes", str(f), str(nx), str(ny), str(raw), "7.1e8", "13", "-0.031332", "0.081212"])
    d = _dump(io_demo, f)
    assert d["Rad"]["type"] == "i2" and d["Rad"]["shape"] == f"{ny}x{nx}"
    assert abs(float(d["Rad"]["atts"]["scale_factor"]) - 0.04572892) < 1e-8      # stored as float32
    # NetCDF-4 dimension conventions: coordinate variables are dimension scales, Rad carries a DIMENSION_LIST,
    # a dimension without a variable has the placeholder name
    assert d["x"]["atts"]["CLASS"] == "DIMENSION_SCALE" and d["y"]["atts"]["CLASS"] == "DIMENSION_SCALE"
    assert d["x"]["atts"]["_Netcdf4Dimid"] == "1" and d["y"]["atts"]["_Netcdf4Dimid"] == "0"
    assert "DIMENSION_LIST" in d["Rad"]["atts"]
    assert d["band"]["atts"]["NAME"].startswith("This is a netCDF dimension but not a netCDF variable.")
    assert d["t"]["shape"] == "" and d["t"]["type"] == "f8"
    assert np.array_equal(_read(io_demo, f, "Rad", "short", tmp_path).reshape(ny, nx), c1)
    assert _read(io_demo, f, "band_id", "int", tmp_path)[0] == 13                  # byte in the file, converted on read
    assert _read(io_demo, f, "t", "double", tmp_path)[0] == 7.1e8
    assert abs(_read(io_demo, f, "planck_fk2", "float", tmp_path)[0] - 1392.74) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("extra,method", [([], "vof"), (["-sosm"], "sosm"), (["-pd", "-no_outrad", "-kiters", "3"], "vof_pd")])
def test_octane_command_line_end_to_end(io_demo, capi, tmp_path, extra, method):
    """octane -i1 a.nc -i2 b.nc -o dir/: files in, outfile.nc out == navcal + flow + pix2uv through the library."""
    nx, ny = 200, 144
    c1, c2 = _counts(nx, ny, 11)
    files = []
    for i, (c, t) in enumerate(((c1, 7.1e8), (c2, 7.1e8 + 300.0))):
        raw = tmp_path / f"rad{i}.bin"
        c.tofile(raw)
        f = tmp_path / f"g{i}.nc"
        subprocess.check_call([io_demo, "--make-goes", str(f), str(nx), str(ny), str(raw), repr(t), "13", "-0.031332", "0.081212"])
        files.append(f)
    outdir = str(tmp_path) + "/"
    r = subprocess.run([os.path.join(LIBD, "octane"), "-i1", str(files[0]), "-i2", str(files[1]), "-o", outdir] + extra,
                       capture_output=True, text=True)
    assert r.returncode == 0 and "outfile.nc written" in r.stdout, r.stdout + r.stderr
    out = tmp_path / "outfile.nc"
    d = _dump(io_demo, out)
    # the same steps through the library
    mx, mn = capi.bandminmax(13)
    kw = dict(xScale=5.6e-05, xOffset=-0.031332, yScale=-5.6e-05, yOffset=0.081212, radScale=0.04572892, radOffset=-1.6443,
              rpol=6356752.31414, req=6378137.0, H=np.float32(35786023.0) + np.float32(6378137.0), lam0=LAM0,
              fk1=10803.3, fk2=1392.74, bc1=0.07550, bc2=0.99975, kap1=0.0015839,
              maxin=mx, minin=mn, maxout=255.0, minout=0.0, cal=0, minx=0, maxx=nx, miny=0, maxy=ny)
    x = np.arange(nx, dtype=np.int16); y = np.arange(ny, dtype=np.int16)
    im1 = capi.navcal(c1, x, y, capi.NavcalParams(donav=1, **kw))[0]
    im2 = capi.navcal(c2, x, y, capi.NavcalParams(donav=0, **kw))[0]
    if method == "sosm":
        u, v = capi.sosm(im1, im2)
    else:
        u, v = capi.flow(im1, im2, capi.FlowParams(kiters=3) if method == "vof_pd" else capi.FlowParams())
    nav = capi.Nav(pph=float(np.float32(35786023.0)), req=float(np.float32(6378137.0)), rpol=float(np.float32(6356752.31414)),
                   lam0=float(LAM0), xScale=5.6e-05, xOffset=-0.031332, yScale=-5.6e-05,
                   yOffset=0.081212, g2xOffset=-0.031332, g2yOffset=0.081212, nx=nx, ny=ny)
    assert im1.min() > 0 and np.abs(u).mean() > 0.5                      # on the disc, and something moves
    want = capi.pix2uv(nav, 7.1e8, 7.1e8 + 300.0, u, v, pixuv=1 if method == "vof_pd" else 0)
    assert np.array_equal(_read(io_demo, out, "U", "short", tmp_path).reshape(ny, nx), want[0])
    assert np.array_equal(_read(io_demo, out, "V", "short", tmp_path).reshape(ny, nx), want[1])
    assert d["U"]["atts"]["units"] == ("x-pixels" if method == "vof_pd" else "meters per second")
    assert abs(float(d["U"]["atts"]["scale_factor"]) - 0.01) < 1e-9
    if method == "vof_pd":
        assert np.array_equal(_read(io_demo, out, "Upix", "float", tmp_path).reshape(ny, nx), u)
        assert "Rad" not in d and "U_raw" in d
    else:
        assert np.array_equal(_read(io_demo, out, "U_raw", "short", tmp_path).reshape(ny, nx), want[2])
        assert np.array_equal(_read(io_demo, out, "Rad", "short", tmp_path).reshape(ny, nx), c1)
        assert abs(float(d["Rad"]["atts"]["add_offset"]) + 1.6443) < 1e-6
    s = d["optical_flow_settings"]["atts"]
    assert abs(float(s["dt_seconds"]) - 300.0) < 1e-3
    assert _read(io_demo, out, "optical_flow_settings", "int", tmp_path)[0] == (4 if method == "sosm" else 1)
    if method == "sosm":
        assert s["Rad"] == "2" and s["SRad"] == "2"
    else:
        assert float(s["alpha"]) == 5.0 and s["K_Iterations"] == ("3" if method == "vof_pd" else "4")
    assert _read(io_demo, out, "t", "double", tmp_path)[0] == 7.1e8
    assert d["goes_imager_projection"]["atts"]["grid_mapping_name"] == "geostationary"


@pytest.mark.gpu
def test_octane_command_line_with_first_guess_file(io_demo, capi, tmp_path):
    """-firstguess fg.nc -lambdac 0.3: navigated winds UFG / VFG are turned into pixel displacements (oct_uv2pix), seed
    the coarsest level and weigh in through the hint term."""
    nx, ny = 200, 144
    c1, c2 = _counts(nx, ny, 21)
    files = []
    for i, (c, t) in enumerate(((c1, 7.1e8), (c2, 7.1e8 + 300.0))):
        raw = tmp_path / f"rad{i}.bin"
        c.tofile(raw)
        f = tmp_path / f"g{i}.nc"
        subprocess.check_call([io_demo, "--make-goes", str(f), str(nx), str(ny), str(raw), repr(t), "13", "-0.031332", "0.081212"])
        files.append(f)
    rng = np.random.RandomState(2)
    ufg = (12.0 + 2.0 * rng.randn(ny, nx)).astype(np.float32)            # m/s
    vfg = (-6.0 + 2.0 * rng.randn(ny, nx)).astype(np.float32)
    uvb = tmp_path / "uv.bin"
    np.concatenate([ufg.ravel(), vfg.ravel()]).tofile(uvb)
    fg = tmp_path / "fg.nc"
    subprocess.check_call([io_demo, "--make-fg", str(fg), str(nx), str(ny), str(uvb)])
    outdir = str(tmp_path) + "/"
    r = subprocess.run([os.path.join(LIBD, "octane"), "-i1", str(files[0]), "-i2", str(files[1]), "-o", outdir, "-firstguess", str(fg),
                        "-lambdac", "0.3", "-pd"], capture_output=True, text=True)
    assert r.returncode == 0 and "outfile.nc written" in r.stdout, r.stdout + r.stderr
    out = tmp_path / "outfile.nc"
    mx, mn = capi.bandminmax(13)
    lam0 = LAM0
    kw = dict(xScale=5.6e-05, xOffset=-0.031332, yScale=-5.6e-05, yOffset=0.081212, radScale=0.04572892, radOffset=-1.6443,
              rpol=6356752.31414, req=6378137.0, H=np.float32(35786023.0) + np.float32(6378137.0), lam0=lam0,
              fk1=10803.3, fk2=1392.74, bc1=0.07550, bc2=0.99975, kap1=0.0015839,
              maxin=mx, minin=mn, maxout=255.0, minout=0.0, cal=0, minx=0, maxx=nx, miny=0, maxy=ny)
    x = np.arange(nx, dtype=np.int16); y = np.arange(ny, dtype=np.int16)
    im1, lat, lon = capi.navcal(c1, x, y, capi.NavcalParams(donav=1, **kw))[:3]
    im2 = capi.navcal(c2, x, y, capi.NavcalParams(donav=0, **kw))[0]
    nav = capi.Nav(pph=float(np.float32(35786023.0)), req=float(np.float32(6378137.0)), rpol=float(np.float32(6356752.31414)),
                   lam0=float(lam0), xScale=5.6e-05, xOffset=-0.031332, yScale=-5.6e-05, yOffset=0.081212,
                   g2xOffset=-0.031332, g2yOffset=0.081212, nx=nx, ny=ny)
    u0, v0 = capi.uv2pix(nav, 7.1e8, 7.1e8 + 300.0, ufg, vfg, lat, lon, x, y)
    assert np.abs(u0).mean() > 0.5                                        # 12 m/s over 300 s on a 2 km grid
    u, v = capi.flow(im1, im2, capi.FlowParams(lambdac=0.3), u0, v0)
    gu = _read(io_demo, out, "Upix", "float", tmp_path).reshape(ny, nx)
    gv = _read(io_demo, out, "Vpix", "float", tmp_path).reshape(ny, nx)
    assert np.array_equal(gu, u), (float(gu.mean()), float(u.mean()), float(u0.mean()), float(np.abs(gu - u).max()), r.stdout)
    assert np.array_equal(gv, v)
    s = _dump(io_demo, out)["optical_flow_settings"]["atts"]
    assert s["dofirstguess"] == "1" and abs(float(s["lambdac"]) - 0.3) < 1e-12
