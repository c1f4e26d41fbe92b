"""ctypes front-end to the CPU oracle (oracle/vof_oracle.c, oracle/pix2uv_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from octane_amd/.  "parity unpinned" for the
whole solver -- see the header of vof_oracle.c for what is and is not pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_S = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")


class Params(C.Structure):
    _fields_ = [("alpha", C.c_double), ("lambda_", C.c_double), ("lambdac", C.c_double),
                ("scaleF", C.c_double), ("kiters", C.c_int), ("liters", C.c_int),
                ("cgiters", C.c_int), ("dozim", C.c_int)]


class Nav(C.Structure):
    _fields_ = [("pph", C.c_double), ("req", C.c_double), ("rpol", C.c_double), ("lam0", C.c_double),
                ("xScale", C.c_float), ("xOffset", C.c_float), ("yScale", C.c_float), ("yOffset", C.c_float),
                ("g2xOffset", C.c_float), ("g2yOffset", C.c_float),
                ("lat1", C.c_float), ("lon1", C.c_float), ("lon0", C.c_float), ("R", C.c_float),
                ("minX", C.c_int), ("minY", C.c_int), ("nx", C.c_int), ("ny", C.c_int)]


class NavcalParams(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("xScale", "xOffset", "yScale", "yOffset", "radScale", "radOffset", "rpol", "req", "H",
                                         "lam0", "fk1", "fk2", "bc1", "bc2", "kap1", "maxin", "minin", "maxout", "minout")] + \
               [(k, C.c_int) for k in ("cal", "donav", "minx", "maxx", "miny", "maxy")]


_TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                        C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int)


class _Trace(C.Structure):
    _fields_ = [("cb", _TRACE_FN), ("user", C.c_void_p)]


def build(force: bool = False) -> None:
    """Compile the oracle (and, when /root/reference is present, oracle/_ref)."""
    # make decides what is out of date (a changed vof_oracle.c must not be shadowed by an older .so)
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + ["all"])
    ref = os.environ.get("OCT_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref, "src")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + ["ref", f"REF={ref}"])
        # the cross-check build of the reference's CUDA translation units (hipify-perl + hipcc, out of tree; tests/test_gpu_refhip.py
        # skips without it): best effort, it pins nothing and must not fail the build
        if subprocess.call(["make", "-C", _HERE, "-s", "refhip", f"REF={ref}"], stderr=subprocess.DEVNULL) != 0:
            print("oracle: the hipified cross-check library (make refhip) did not build; tests/test_gpu_refhip.py will skip", file=sys.stderr)


def ref_helpers_path() -> str:
    return os.path.join(_HERE, "_ref", "liboct_ref_helpers.so")


_libs: dict[str, C.CDLL] = {}


def lib(flavour: str = "strict") -> C.CDLL:
    """flavour: 'strict' (no FMA contraction), 'fma', or 'omp' (strict arithmetic, loops spread over the host
    cores with OpenMP; bit-identical to 'strict' under the grid dot schedule), or 'fma_omp' (the same for 'fma')."""
    if flavour in _libs:
        return _libs[flavour]
    build()
    name = {"strict": "liboct_oracle.so", "fma": "liboct_oracle_fma.so", "omp": "liboct_oracle_omp.so",
            "fma_omp": "liboct_oracle_fma_omp.so"}[flavour]
    if os.environ.get("OCT_SANITIZE") == "1":      # the ASan + UBSan builds (`make sanitize`; the process runs under LD_PRELOAD=libasan.so:libubsan.so)
        subprocess.check_call(["make", "-C", _HERE, "-s", "sanitize"])
        L = C.CDLL(os.path.join(_HERE, "_san", name))
    else:
        L = C.CDLL(os.path.join(_HERE, name))
    L.oct_oracle_num_threads.restype = C.c_int
    L.oct_oracle_vof.restype = C.c_int
    L.oct_oracle_vof.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, _F, _F, C.POINTER(Params), C.c_void_p]
    L.oct_oracle_level_dims.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.oct_oracle_level_factor.restype = C.c_float
    L.oct_oracle_level_factor.argtypes = [C.c_float, C.c_int, C.c_int]
    L.oct_oracle_blur_halfwidth.restype = C.c_int
    L.oct_oracle_blur_halfwidth.argtypes = [C.c_float]
    L.oct_oracle_gauss_taps.argtypes = [C.c_float, C.c_int, _F]
    L.oct_oracle_blur_rows.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int]
    L.oct_oracle_blur_cols.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int]
    L.oct_oracle_clamp_coord.restype = C.c_float
    L.oct_oracle_clamp_coord.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_int)]
    L.oct_oracle_bilinear.restype = C.c_float
    L.oct_oracle_bilinear.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.oct_oracle_bicubic.restype = C.c_float
    L.oct_oracle_bicubic.argtypes = [_F, C.c_float, C.c_float, C.c_int, C.c_int]
    L.oct_oracle_decimate.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_float]
    L.oct_oracle_gradient.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_int]
    L.oct_oracle_upsample_flow.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    L.oct_oracle_nnz_before.restype = C.c_long
    L.oct_oracle_nnz_before.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]
    L.oct_oracle_set_dot_schedule.argtypes = [C.c_int]
    L.oct_oracle_pix2uv.restype = C.c_int
    L.oct_oracle_pix2uv.argtypes = [C.POINTER(Nav), C.c_double, C.c_double, _F, _F, C.c_int, C.c_int,
                                    _S, _S, _S, _S, C.POINTER(C.c_float)]
    _libs[flavour] = L
    return L


@dataclass
class FlowParams:
    """The OFFlags fields the solver reads (ref include/offlags.h, .cu:1229-1241),
    with the CLI defaults of ref src/main.cc:78-96."""
    alpha: float = 5.0
    lambda_: float = 1.0
    lambdac: float = 0.0
    scaleF: float = 0.5
    kiters: int = 4
    liters: int = 3
    cgiters: int = 30
    dozim: int = 1

    def c(self) -> Params:
        return Params(self.alpha, self.lambda_, self.lambdac, self.scaleF,
                      self.kiters, self.liters, self.cgiters, self.dozim)


def num_threads(flavour: str = "strict") -> int:
    """Host threads the given flavour spreads its loops over."""
    return int(lib(flavour).oct_oracle_num_threads())


def host_cpu_share() -> int:
    """CPUs this process may actually use: the scheduler affinity, cut down to the cgroup CPU quota if one is set
    (a GPU box shows all 256 hardware threads but grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def set_threads(n: int, flavour: str = "omp") -> None:
    lib(flavour).oct_oracle_set_threads(int(n))


REF_GRID_THREADS = 20 * 16 * 128   # the reference's launch: 20 SMs (hard-coded, ref .cu:1422) x 16 blocks x 128 threads


def flow(img1: np.ndarray, img2: np.ndarray, prm: FlowParams | None = None,
         u0: np.ndarray | None = None, v0: np.ndarray | None = None,
         trace: dict | None = None, flavour: str = "strict", dot_threads: int = 0):
    """Run the oracle solver.  Images are [nc, ny, nx] or [ny, nx] float32
    (x fastest, channel-planar: Image.data[i + nx*j + nx*ny*c]).
    Returns (u, v, total_pcg_iterations).  If `trace` is a dict it is filled with
    {(tag, level, gnc, l): array[nplanes, ny, nx]}."""
    prm = prm or FlowParams()
    a = np.ascontiguousarray(img1, dtype=np.float32)
    b = np.ascontiguousarray(img2, dtype=np.float32)
    if a.ndim == 2:
        a = a[None]
        b = b[None]
    nc, ny, nx = a.shape
    u = np.zeros((ny, nx), np.float32) if u0 is None else np.array(u0, dtype=np.float32, order="C", copy=True)
    v = np.zeros((ny, nx), np.float32) if v0 is None else np.array(v0, dtype=np.float32, order="C", copy=True)
    L = lib(flavour)
    tr_ptr = None
    keep = None
    if trace is not None:
        def _cb(user, tag, k, gnc, l, data, tnx, tny, npl):
            arr = np.ctypeslib.as_array(data, shape=(npl, tny, tnx)).copy()
            trace[(tag.decode(), k, gnc, l)] = arr
        keep = _TRACE_FN(_cb)
        t = _Trace(keep, None)
        tr_ptr = C.cast(C.pointer(t), C.c_void_p)
    p = prm.c()
    # dot-product schedule: 0 = one thread (what the survey recorded), >0 = the CUDA launch
    # geometry (see dotf in vof_oracle.c); large frames need the latter to be meaningful.
    L.oct_oracle_set_dot_schedule(dot_threads)
    try:
        its = L.oct_oracle_vof(a, b, nx, ny, nc, u, v, C.byref(p), tr_ptr)
    finally:
        L.oct_oracle_set_dot_schedule(0)
    if its < 0:
        raise RuntimeError(f"oracle failed with code {its}")
    return u, v, its


def level_dims(nx: int, ny: int, factor: float):
    lx, ly = C.c_int(), C.c_int()
    lib().oct_oracle_level_dims(nx, ny, factor, C.byref(lx), C.byref(ly))
    return lx.value, ly.value


P2U_SITES = ("F1 base x (float)", "F2 base y (float)", "D1 rate_x*dt+xi", "D2 (..)*xScale+xOffset", "D3 rate_y*dt+yi", "D4 (..)*yScale+yOffset",
             "D5 sds", "D6 cos2y+k*sin2y", "D7 sin2x+cos2x*(..)", "D8 H*H-req*req", "D9 b*b-4ac", "D10 (H-sx)^2+sy^2", "H1 haversine a")
P2U_FLOAT_SITES = 0b11          # the two float multiply-adds of the base position: what decides the shorts (profiles/r5_pix2uv_sites.txt)
P2U_ALL_SITES = (1 << len(P2U_SITES)) - 1


def pix2uv(nav: Nav, t1: float, t2: float, u: np.ndarray, v: np.ndarray, pixuv: int = 0, mode: int = 0, flavour: str = "strict", sites: int = 0):
    """Returns (ur, vr, ur2, vr2, dT, sector_moved).  flavour "fma": the build with contracted multiply-adds (the compiler's choice of
    sites); `sites` (strict build): bit mask of the multiply-add sites evaluated as ONE fused operation, pix2uv_oracle.c."""
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    n = u.size
    ur, vr, ur2, vr2 = (np.zeros(n, np.int16) for _ in range(4))
    dT = C.c_float()
    L = lib(flavour)
    L.oct_oracle_pix2uv_fma_sites(C.c_uint(sites))
    try:
        moved = L.oct_oracle_pix2uv(C.byref(nav), t1, t2, u.ravel(), v.ravel(), pixuv, mode, ur, vr, ur2, vr2, C.byref(dT))
    finally:
        L.oct_oracle_pix2uv_fma_sites(C.c_uint(0))
    shp = u.shape
    return ur.reshape(shp), vr.reshape(shp), ur2.reshape(shp), vr2.reshape(shp), dT.value, moved


def navcal(data2: np.ndarray, x: np.ndarray, y: np.ndarray, prm: NavcalParams):
    """Returns (data3, lat, lon, data2s, xs, ys) for the window of prm."""
    d2 = np.ascontiguousarray(data2, np.int16)
    ny, nx = d2.shape
    ww, wh = prm.maxx - prm.minx, prm.maxy - prm.miny
    data3, lat, lon = (np.zeros((wh, ww), np.float32) for _ in range(3))
    d2s = np.zeros((wh, ww), np.int16)
    xs, ys = np.zeros(ww, np.int16), np.zeros(wh, np.int16)
    L = lib()
    L.oct_oracle_navcal.argtypes = [_S, _S, _S, C.c_int, C.c_int, C.POINTER(NavcalParams), _F, _F, _F, _S, _S, _S]
    L.oct_oracle_navcal.restype = None
    L.oct_oracle_navcal(d2.ravel(), np.ascontiguousarray(x, np.int16), np.ascontiguousarray(y, np.int16), nx, ny, C.byref(prm),
                        data3.ravel(), lat.ravel(), lon.ravel(), d2s.ravel(), xs, ys)
    return data3, lat, lon, d2s, xs, ys


class ProjNavcalParams(C.Structure):
    _fields_ = [("xScale", C.c_float), ("xOffset", C.c_float), ("yScale", C.c_float), ("yOffset", C.c_float),
                ("lon0", C.c_float), ("lat1", C.c_float), ("R", C.c_float), ("donav", C.c_int), ("mode", C.c_int),
                ("minx", C.c_int), ("maxx", C.c_int), ("miny", C.c_int), ("maxy", C.c_int)]


def proj_navcal(data2: np.ndarray, x: np.ndarray, y: np.ndarray, prm: ProjNavcalParams):
    """Polar (mode 1) / mercator (mode 2) navigation.  Returns (data3, lat, lon, data2s, xs, ys) for prm's window."""
    d2 = np.ascontiguousarray(data2, np.float32)
    ny, nx = d2.shape
    ww, wh = prm.maxx - prm.minx, prm.maxy - prm.miny
    data3, lat, lon = (np.zeros((wh, ww), np.float32) for _ in range(3))
    d2s = np.ones((wh, ww), np.int16)
    xs, ys = np.zeros(ww, np.int16), np.zeros(wh, np.int16)
    L = lib()
    L.oct_oracle_proj_navcal.argtypes = [_F, _S, _S, C.c_int, C.c_int, C.POINTER(ProjNavcalParams), _F, _F, _F, _S, _S, _S]
    L.oct_oracle_proj_navcal.restype = None
    L.oct_oracle_proj_navcal(d2.ravel(), np.ascontiguousarray(x, np.int16), np.ascontiguousarray(y, np.int16), nx, ny,
                             C.byref(prm), data3.ravel(), lat.ravel(), lon.ravel(), d2s.ravel(), xs, ys)
    return data3, lat, lon, d2s, xs, ys


def sosm(img1, img2, rad: int = 2, srad: int = 2, u0=None, v0=None, flavour: str = "strict"):
    """Patch-matching flow (restatement).  Returns (u, v)."""
    a = np.ascontiguousarray(img1, np.float32); b = np.ascontiguousarray(img2, np.float32)
    ny, nx = a.shape
    u = np.zeros((ny, nx), np.float32) if u0 is None else np.array(u0, np.float32, order="C", copy=True)
    v = np.zeros((ny, nx), np.float32) if v0 is None else np.array(v0, np.float32, order="C", copy=True)
    L = lib(flavour)
    L.oct_oracle_sosm.argtypes = [_F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int]
    L.oct_oracle_sosm.restype = None
    L.oct_oracle_sosm(a.ravel(), b.ravel(), u.ravel(), v.ravel(), nx, ny, rad, srad)
    return u, v


def ref_sosm(img1, img2, rad: int = 2, srad: int = 2, u0=None, v0=None):
    """The REFERENCE's own oct_patch_match_optical_flow, from oracle/_ref (only where /root/reference exists)."""
    a = np.ascontiguousarray(img1, np.float32); b = np.ascontiguousarray(img2, np.float32)
    ny, nx = a.shape
    u = np.zeros((ny, nx), np.float32) if u0 is None else np.array(u0, np.float32, order="C", copy=True)
    v = np.zeros((ny, nx), np.float32) if v0 is None else np.array(v0, np.float32, order="C", copy=True)
    R = C.CDLL(ref_helpers_path())
    R.oct_ref_patch_match.argtypes = [_F, _F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int]
    R.oct_ref_patch_match.restype = None
    R.oct_ref_patch_match(a.ravel(), b.ravel(), u.ravel(), v.ravel(), nx, ny, rad, srad)
    return u, v


def uv2pix(nav: Nav, t1: float, t2: float, u, v, lat, lon, gx, gy):
    uu = np.array(u, np.float32, order="C", copy=True); vv = np.array(v, np.float32, order="C", copy=True)
    L = lib()
    L.oct_oracle_uv2pix.argtypes = [C.POINTER(Nav), C.c_double, C.c_double, _F, _F, _F, _F, _S, _S]
    L.oct_oracle_uv2pix.restype = None
    L.oct_oracle_uv2pix(C.byref(nav), t1, t2, uu.ravel(), vv.ravel(), np.ascontiguousarray(lat, np.float32).ravel(),
                        np.ascontiguousarray(lon, np.float32).ravel(), np.ascontiguousarray(gx, np.int16), np.ascontiguousarray(gy, np.int16))
    return uu, vv


def srsal(u, v, cth):
    uu = np.ascontiguousarray(u, np.float32); vv = np.ascontiguousarray(v, np.float32); cc = np.ascontiguousarray(cth, np.float32)
    ny, nx = uu.shape
    uo, vo = np.zeros_like(uu), np.zeros_like(vv)
    L = lib()
    L.oct_oracle_srsal.argtypes = [_F, _F, _F, C.c_int, C.c_int, _F, _F]
    L.oct_oracle_srsal.restype = None
    L.oct_oracle_srsal(uu.ravel(), vv.ravel(), cc.ravel(), nx, ny, uo.ravel(), vo.ravel())
    return uo, vo


# ---- the CROSS-CHECK build of the reference's CUDA translation units (oracle/Makefile `refhip`, oracle/ref_hip_wrap.cc) ------------
# hipify-perl + hipcc on the reference's own kernel text, out of tree; a tool stand-in that pins nothing (the oracle stays "parity
# unpinned") but the only independent witness of the restatement.  Needs a GPU to run; None where the library was not built.
def refhip_path() -> str:
    return os.path.join(_HERE, "_ref", "liboct_ref_hip.so")


def _one_hip_runtime() -> None:
    """One HIP runtime per process (as octane_amd/capi.py does for the product library): the PyTorch-ROCm wheel bundles its own
    libamdhip64.so; a library that pulled in /opt/rocm's copy first would leave whichever runtime initialises second without a GPU."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def refhip_lib():
    p = refhip_path()
    if not os.path.exists(p) and os.path.isdir("/root/reference/src"):
        subprocess.call(["make", "-C", _HERE, "-s", "refhip"])
    if not os.path.exists(p):
        return None
    _one_hip_runtime()
    L = C.CDLL(p)
    L.oct_refhip_vof.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, _F, _F, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.oct_refhip_pix2uv.argtypes = [C.c_double] * 4 + [C.c_float] * 10 + [C.c_int] * 4 + [C.c_double, C.c_double, _F, _F, C.c_int, C.c_int,
                                                                                         _S, _S, _S, _S, C.POINTER(C.c_float), C.c_int]
    return L


def refhip_flow(img1, img2, prm: "FlowParams | None" = None, u0=None, v0=None, device: int = 0):
    """The reference's own kernel (hipified) on the GPU: returns (u, v).  Prints what the reference prints."""
    L = refhip_lib()
    prm = prm or FlowParams()
    a = np.ascontiguousarray(img1, dtype=np.float32)
    b = np.ascontiguousarray(img2, dtype=np.float32)
    if a.ndim == 2:
        a, b = a[None], b[None]
    nc, ny, nx = a.shape
    u = np.zeros((ny, nx), np.float32) if u0 is None else np.array(u0, dtype=np.float32, order="C", copy=True)
    v = np.zeros((ny, nx), np.float32) if v0 is None else np.array(v0, dtype=np.float32, order="C", copy=True)
    p = prm.c()
    L.oct_refhip_vof(a, b, nx, ny, nc, u, v, p.alpha, p.lambda_, p.lambdac, p.scaleF, 400.0, p.kiters, p.liters, p.cgiters, p.dozim, device)
    return u, v


def refhip_pix2uv(nav: Nav, t1: float, t2: float, u, v, pixuv: int = 0, mode: int = 0, device: int = 0):
    L = refhip_lib()
    u = np.array(u, np.float32, order="C", copy=True)
    v = np.array(v, np.float32, order="C", copy=True)
    n = u.size
    ur, vr, ur2, vr2 = (np.zeros(n, np.int16) for _ in range(4))
    dT = C.c_float()
    L.oct_refhip_pix2uv(nav.pph, nav.req, nav.rpol, nav.lam0, nav.xScale, nav.xOffset, nav.yScale, nav.yOffset, nav.g2xOffset, nav.g2yOffset,
                        nav.lat1, nav.lon1, nav.lon0, nav.R, nav.minX, nav.minY, nav.nx, nav.ny, t1, t2, u.ravel(), v.ravel(), pixuv, mode,
                        ur, vr, ur2, vr2, C.byref(dT), device)
    shp = u.shape
    return ur.reshape(shp), vr.reshape(shp), ur2.reshape(shp), vr2.reshape(shp), dT.value
