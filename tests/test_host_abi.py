"""The host-side drop-in: layout of Image/OFFlags/GOESNAVVar/GOESVar against the reference's headers, the
C++ shim with the reference's signatures, and the `octane` command-line contract (ref src/main.cc)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"
BUILD = os.path.join(ROOT, "tests", "cpp", "build")


def _compile(src, out, inc, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    from conftest import SAN_FLAGS
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", *SAN_FLAGS, "-I", inc, os.path.join(ROOT, "tests", "cpp", src), "-o", out, *extra])
    return out


@pytest.fixture(scope="module")
def host_demo(capi):
    capi.lib()
    from conftest import host_libdir, host_make_args
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "octane_amd", "csrc"), "-s", "-f", "Makefile.host", *host_make_args()])
    return _compile("host_demo.cpp", os.path.join(BUILD, "host_demo"), os.path.join(ROOT, "include"),
                    ["-L", host_libdir(), "-loctane_host", "-loctane_vof", "-Wl,-rpath," + host_libdir()])


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="reference headers not on this machine")
def test_type_layouts_equal_the_reference_headers():
    mine = subprocess.check_output([_compile("print_layout.cpp", os.path.join(BUILD, "layout_mine"), os.path.join(ROOT, "include"))])
    ref = subprocess.check_output([_compile("print_layout.cpp", os.path.join(BUILD, "layout_ref"), REF_INC)])
    assert mine == ref
    assert mine.count(b"\n") > 100


def _parse(host_demo, *args):
    out = subprocess.check_output([host_demo, "--parse-only", *args]).decode()
    return dict(line.split("=", 1) for line in out.strip().splitlines())


def test_command_line_defaults_and_readme_example(host_demo):
    # ref README.md:36: octane -i1 f1.nc -i2 f2.nc -alpha 5 -lambda 1
    d = _parse(host_demo, "-i1", "a.nc", "-i2", "b.nc", "-alpha", "5", "-lambda", "1")
    assert d["f1"] == "a.nc" and d["f2"] == "b.nc" and d["help"] == "0" and d["ftype"] == "GOES"
    assert float(d["alpha"]) == 5.0 and float(d["lambda"]) == 1.0 and float(d["lambdac"]) == 0.0
    assert (d["kiters"], d["liters"], d["cgiters"], d["dozim"], d["setdevice"]) == ("4", "3", "30", "1", "0")
    assert float(d["scaleF"]) == 0.5 and float(d["scsig"]) == 400.0 and d["oftype"] == "1" and d["outdir"] == "./"
    assert _parse(host_demo, "-i1", "a")["help"] == "1"          # fewer than 4 argv -> usage text (main.cc:112)


def test_command_line_quirks_are_kept(host_demo):
    d = _parse(host_demo, "-i1", "a", "-i2", "b", "-scsig", "3", "-set_device", "2", "-corn", "-cgiters", "99",
               "-kiters", "6", "-liters", "5", "-brox", "-pd", "-lambdac", "0.25", "-o", "/tmp/x/", "-normmin2", "1.5")
    assert float(d["scsig"]) == 9.0            # -scsig squares its argument (main.cc:229)
    assert d["setdevice"] == "1"               # -set_device is 1-based (main.cc:313)
    assert d["docorn"] == "0"                  # -corn sets docorn = 0 (main.cc:270-273)
    assert d["cgiters"] == "30"                # documented, never parsed (main.cc:144)
    assert (d["kiters"], d["liters"], d["dozim"], d["pixuv"], d["oftype"]) == ("6", "5", "0", "1", "3")
    assert float(d["lambdac"]) == 0.25 and d["outdir"] == "/tmp/x/"
    assert d["setNormMin2"] == "0" and float(d["NormMin2"]) == 1.5
    d = _parse(host_demo, "-i1", "a", "-i2", "b", "-Polar", "-i1cth", "c.nc", "-sosm")
    assert d["ftype"] == "POLAR" and d["dopolar"] == "1" and d["doCTH"] == "0" and d["oftype"] == "4"
    d = _parse(host_demo, "-i1", "a", "-i2", "b", "-Merc", "-ic21", "c", "-ic22", "d", "-no_outraw", "-firstguess", "fg.nc")
    assert d["ftype"] == "MERC" and d["doc2"] == "1" and d["fc21"] == "c" and d["outraw"] == "0" and d["dofirstguess"] == "1"


@pytest.mark.gpu
def test_cpp_host_path_equals_the_python_binding(host_demo, capi, tmp_path):
    """oct_optical_flow() through the C++ shim == capi.flow + capi.pix2uv through ctypes, bit for bit."""
    from octane_amd import synth
    nx, ny = 144, 100
    a, b = synth.lattice_scene(nx, ny, seed=31)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(a.tobytes()); f.write(b.tobytes())
    r = subprocess.run([host_demo, "--run", str(nx), str(ny), str(inp), str(outp), "-i1", "x", "-i2", "y", "-kiters", "3", "-alpha", "6"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dT=300" in r.stdout
    raw = open(outp, "rb").read()
    n = nx * ny
    u = np.frombuffer(raw, np.float32, n, 0).reshape(ny, nx)
    v = np.frombuffer(raw, np.float32, n, 4 * n).reshape(ny, nx)
    shorts = np.frombuffer(raw, np.int16, 4 * n, 8 * n).reshape(4, ny, nx)
    ue, ve = capi.flow(a, b, capi.FlowParams(kiters=3, alpha=6.0))
    assert np.array_equal(u, ue) and np.array_equal(v, ve)
    nav = capi.Nav(pph=35786023.0, req=6378137.0, rpol=6356752.31414, lam0=-1.308996939, xScale=5.6e-05, xOffset=-0.101332,
                   yScale=-5.6e-05, yOffset=0.128212, g2xOffset=-0.101332, g2yOffset=0.128212, nx=nx, ny=ny)
    # the shim navigates with the strict build + the two float multiply-adds of the base position fused (what nvcc's -fmad=true makes of
    # the reference's kernel, include/octane_vof.h OCTANE_NAV_FMAD_FLOAT) ...
    want = capi.pix2uv(nav, 1000.0, 1300.0, ue, ve, 0, capi.NAV_GEOS | capi.NAV_FMAD_FLOAT)
    for i in range(4):
        assert np.array_equal(shorts[i], want[i])
    # ... and OCTANE_PIX2UV_FMAD=0 in the environment makes it the unfused build (a reference compiled with -fmad=false)
    import os
    r = subprocess.run([host_demo, "--run", str(nx), str(ny), str(inp), str(outp), "-i1", "x", "-i2", "y", "-kiters", "3", "-alpha", "6"],
                       capture_output=True, text=True, env=dict(os.environ, OCTANE_PIX2UV_FMAD="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    shorts0 = np.frombuffer(open(outp, "rb").read(), np.int16, 4 * n, 8 * n).reshape(4, ny, nx)
    strict = capi.pix2uv(nav, 1000.0, 1300.0, ue, ve)
    for i in range(4):
        assert np.array_equal(shorts0[i], strict[i])


@pytest.mark.gpu
def test_cpp_host_path_with_row_bands(host_demo, capi, tmp_path):
    """OCTANE_VOF_BANDS=2 sends oct_variational_optical_flow() through the row-band solve (two virtual bands on
    device 0 here); the flow is the plain solve's up to reduction order, and the navigated shorts follow it."""
    import os
    from conftest import rel_l2
    from octane_amd import synth
    nx, ny = 320, 288
    a, b = synth.lattice_scene(nx, ny, seed=7)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(a.tobytes()); f.write(b.tobytes())
    env = dict(os.environ, OCTANE_VOF_BANDS="2", OCTANE_TUNE_MIN_BAND_PIXELS="1")
    r = subprocess.run([host_demo, "--run", str(nx), str(ny), str(inp), str(outp), "-i1", "x", "-i2", "y", "-kiters", "3"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    n = nx * ny
    u = np.frombuffer(raw, np.float32, n, 0).reshape(ny, nx)
    v = np.frombuffer(raw, np.float32, n, 4 * n).reshape(ny, nx)
    ue, ve = capi.flow(a, b, capi.FlowParams(kiters=3))
    assert rel_l2(u, v, ue, ve) < 2e-5


@pytest.mark.gpu
def test_cpp_host_path_dispatches_sosm(host_demo, capi, tmp_path):
    """-sosm through oct_optical_flow() (ref oct_optical_flow.cc:57-64) == capi.sosm, bit for bit, with -rad / -srad."""
    from octane_amd import synth
    nx, ny = 160, 112
    a, b = (x[0] for x in synth.lattice_scene(nx, ny, seed=13))
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(a.tobytes()); f.write(b.tobytes())
    r = subprocess.run([host_demo, "--run", str(nx), str(ny), str(inp), str(outp), "-i1", "x", "-i2", "y", "-sosm", "-rad", "3", "-srad", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    n = nx * ny
    u = np.frombuffer(raw, np.float32, n, 0).reshape(ny, nx)
    v = np.frombuffer(raw, np.float32, n, 4 * n).reshape(ny, nx)
    ue, ve = capi.sosm(a, b, 3, 1)
    assert np.array_equal(u, ue) and np.array_equal(v, ve)
