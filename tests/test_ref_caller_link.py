"""The literal drop-in proof from the reference's side (build container only; nothing of it travels to the GPU box): the
reference's own caller translation unit, src/oct_optical_flow.cc, is compiled WHERE IT LIES with the REFERENCE's headers and
linked against liboctane_host.so.  It declares the entry points itself (oct_optical_flow.cc:11-17) and calls them with
`Image` / `OFFlags` BY VALUE (:63,:67,:91), so a resolved link + a traced call proves the mangled names and the by-value
calling convention from the reference's side, not from this repository's headers.

The one include of that file the image lacks, <netcdf>, is not used by it; an EMPTY file of that name is put on the
include path in a temporary directory.  This is an ABI link check -- it pins nothing about the oracle or the solver's results."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
REF_TU = os.path.join(REF, "src", "oct_optical_flow.cc")

pytestmark = pytest.mark.skipif(not os.path.isfile(REF_TU), reason="the reference tree is not on this machine")


@pytest.fixture(scope="module")
def built(tmp_path_factory, capi):
    capi.lib()
    from conftest import host_libdir, host_make_args
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "octane_amd", "csrc"), "-s", "-f", "Makefile.host", *host_make_args()])
    d = tmp_path_factory.mktemp("ref_caller")          # outside the repository: objects of reference code never enter the tree
    open(d / "netcdf", "w").close()                     # the unused include
    obj = str(d / "ref_oct_optical_flow.o")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", os.path.join(REF, "include"), "-I", str(d), "-c", REF_TU, "-o", obj])
    exe = str(d / "ref_caller")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I", os.path.join(REF, "include"), os.path.join(ROOT, "tests", "cpp", "ref_caller_main.cpp"), obj,
                           "-o", exe, "-L", host_libdir(), "-loctane_host", "-L", os.path.join(ROOT, "octane_amd"), "-loctane_vof",
                           "-Wl,-rpath," + host_libdir(), "-Wl,-rpath," + os.path.join(ROOT, "octane_amd")])
    return obj, exe, os.path.join(host_libdir(), "liboctane_host.so")


def _syms(args):
    return {l.split()[-1] for l in subprocess.check_output(["nm", *args]).decode().splitlines() if l.strip()}


def test_every_entry_point_the_reference_caller_needs_is_exported(built):
    obj, _, lib = built
    want = {s for s in _syms(["-u", obj]) if re.match(r"_Z\d+oct_", s)}
    have = _syms(["-D", "--defined-only", lib])
    # the five the reference declares at src/oct_optical_flow.cc:11-17 -- all of them, mangled with Image / OFFlags / GOESVar& as the reference spells them
    demangled = subprocess.check_output(["c++filt", *sorted(want)]).decode().splitlines()
    assert len(want) == 5, demangled
    assert {d.split("(")[0] for d in demangled} == {"oct_patch_match_optical_flow", "oct_variational_optical_flow", "oct_pix2uv_cuda", "oct_uv2pix", "oct_srsal_cu"}
    assert want <= have, sorted(want - have)
    # and the symbol the object DEFINES is the one the library exports for hosts that link the library's own caller instead
    assert {s for s in _syms(["--defined-only", obj]) if s.startswith("_Z16oct_optical_flow")} <= have


def test_the_reference_caller_reaches_the_library_with_its_by_value_arguments(built):
    """Run it.  Without a GPU the library prints the reference's own message and exit(0)s (ref .cu:1255-1259) -- after the trace line that
    shows what arrived through oct_variational_optical_flow(Image, Image, float*, float*, float*, int, int, int, OFFlags)."""
    _, exe, _ = built
    r = subprocess.run([exe, "64", "48"], capture_output=True, text=True, env=dict(os.environ, OCTANE_HOST_TRACE="1"), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    tr = [l for l in r.stderr.splitlines() if l.startswith("TRACE oct_variational_optical_flow")]
    assert len(tr) == 1, r.stderr
    t = tr[0]
    assert "nx=64 ny=48 nc=1 " in t and "geo1i={64,48,1,11.5}" in t and "geo2i={64,48,1,22.25}" in t and "u0=0 v0=0" in t
    assert "alpha=5.5 lambda=1.25 lambdac=0.125 scaleF=0.5 scsig=400 kiters=3 liters=2 cgiters=7 dozim=1 setdevice=0 ftype=GOES" in t
    if "No gpus available for use, exiting" in r.stdout:
        return                                            # the build container: the call arrived, the reference's exit path was taken
    # a machine with a GPU and the reference tree: the whole call chain ran
    assert "TRACE oct_pix2uv_cuda t1=1000 t2=1300 nav={nx=64 ny=48" in r.stderr
    assert "rc=1 dT=300" in r.stdout
