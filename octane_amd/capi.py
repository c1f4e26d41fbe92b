"""ctypes binding of the C-ABI in include/octane_vof.h (liboctane_vof.so).

This is the product path seen from Python: every call goes through the HIP library.  There is
no CPU fallback -- if the library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCTANE_LIB selects another build of the same library (tools/build_variants.sh + tools/time_variants.py: A/B timing of kernel variants)
LIB_PATH = os.environ.get("OCTANE_LIB") or os.path.join(_HERE, "liboctane_vof.so")

OK, E_INVALID, E_NODEVICE, E_HIP, E_TOOSMALL, E_NOMEM = 0, -1, -2, -3, -4, -5
MEM_HOST, MEM_DEVICE = 0, 1
NAV_GEOS, NAV_POLAR, NAV_MERC = 0, 1, 2
NAV_FMAD = 0x100      # or into mode: the navigation kernel built with fused multiply-adds (include/octane_vof.h)
NAV_FMAD_FLOAT = 0x200   # or into mode: the strict build with exactly the two float sites of the base position fused (the C++ shim's default)

# The DIAGNOSTIC library (make -C octane_amd/csrc DIAG=1: the same sources + -DOCTANE_DIAG=1).  It exports, on top of EXPORTS, what
# include/octane_vof_dev.h declares under OCTANE_DIAG: the tuning knob octane_vof_tune, probes, self-tests, the stamped kernel copies --
# and only it contains the two-pass form of the PCG iteration.  tools/ and the form-against-form tests bind it with diag(); the product
# path (flow(), bench.py, smoke()) never does.
DIAG_LIB_PATH = os.path.join(_HERE, "liboctane_vof_diag.so")
DIAG_EXPORTS = ("octane_vof_tune", "octane_vof_mid_geometry", "octane_vof_row_rotation", "octane_selftest_rcp", "octane_selftest_assembly_math", "octane_selftest_assembly_math_bits",
                "octane_vof_plan_probe", "octane_vof_plan_probe_stamps", "octane_vof_mid_stamps")

# every symbol include/octane_vof.h, include/octane_extras.h and the product section of include/octane_vof_dev.h declare: exactly what
# liboctane_vof.so exports (tests/test_capi_cpu.py compares the three lists: headers, this tuple, `nm -D`)
EXPORTS = (
    "octane_vof_default_params", "octane_vof_run", "octane_vof_solve", "octane_vof_release_cache", "octane_vof_plan_create", "octane_vof_plan_destroy",
    "octane_vof_plan_device_bytes", "octane_vof_plan_run", "octane_vof_plan_solve", "octane_vof_plan_wait", "octane_vof_plan_last_iterations",
    "octane_vof_plan_persist_state", "octane_vof_plan_set_lane_mode",
    "octane_vof_plan_placement_trials", "octane_vof_plan_set_trace", "octane_vof_plan_set_profiling", "octane_vof_plan_get_profile", "octane_vof_plan_get_launch_times",
    "octane_vof_batch_run",
    "octane_vof_tiled_create", "octane_vof_tiled_destroy", "octane_vof_tiled_load", "octane_vof_tiled_solve",
    "octane_vof_tiled_wait", "octane_vof_tiled_fetch", "octane_vof_tiled_run", "octane_vof_tiled_banded_levels",
    "octane_vof_tiled_band_rows", "octane_vof_tiled_last_iterations", "octane_vof_tiled_last_copies",
    "octane_vof_tiled_device_bytes", "octane_vof_band_partition",
    "octane_vof_mp_create", "octane_vof_mp_handles", "octane_vof_mp_connect", "octane_vof_mp_run", "octane_vof_mp_banded_levels",
    "octane_vof_mp_last_iterations", "octane_vof_mp_destroy",
    "octane_vof_tiled_transport_info", "octane_vof_mp_transport_info", "octane_vof_transport_name", "octane_vof_mp_set_exchange",
    "octane_vof_mp_selfcheck",
    "octane_pix2uv_run", "octane_navcal_run", "octane_bandminmax",
    "octane_proj_navcal_run", "octane_uv2pix_run", "octane_srsal_run", "octane_sosm_run",
    "octane_last_error", "octane_device_count",
)


def diag():
    """A second, independent binding of this module onto the DIAGNOSTIC library (its own module object, its own loaded library): what
    exposes Plan.tune / Plan.probe / the self-tests.  Raises ImportError when `make -C octane_amd/csrc DIAG=1` has not been run."""
    return variant(DIAG_LIB_PATH)


def dev():
    """The binding developer tools use: this module when its library already has the developer exports (OCTANE_LIB names a diagnostic or
    variant build), the diagnostic library's binding otherwise."""
    import sys
    return sys.modules[__name__] if hasattr(lib(), "octane_vof_tune") else diag()


_variants: dict = {}


def variant(lib_path: str):
    """This module bound to another build of the library (the diagnostic library, a kernel variant of tools/build_variants.sh)."""
    import importlib.util
    import sys
    lib_path = os.path.abspath(lib_path)
    if lib_path == os.path.abspath(LIB_PATH):
        return sys.modules[__name__]
    if lib_path in _variants:
        return _variants[lib_path]
    if not os.path.exists(lib_path):
        raise ImportError(f"{lib_path} is missing (the diagnostic library is built by `make -C octane_amd/csrc DIAG=1`)")
    name = f"{__name__}__{len(_variants)}_{os.path.basename(lib_path).replace('.', '_')}"
    spec = importlib.util.spec_from_file_location(name, os.path.abspath(__file__))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.LIB_PATH = lib_path
    _variants[lib_path] = mod
    return mod


class VofParams(C.Structure):
    _fields_ = [("alpha", C.c_double), ("lambda_", C.c_double), ("lambdac", C.c_double),
                ("scaleF", C.c_double), ("scsig", C.c_double), ("kiters", C.c_int), ("liters", C.c_int),
                ("cgiters", C.c_int), ("dozim", C.c_int), ("device", C.c_int)]


class VofProfile(C.Structure):
    _fields_ = [("pass_a_ms", C.c_double), ("pass_a_launches", C.c_longlong),
                ("pass_b_ms", C.c_double), ("pass_b_launches", C.c_longlong),
                ("assemble_ms", C.c_double), ("assemble_launches", C.c_longlong),
                ("update_ms", C.c_double), ("update_launches", C.c_longlong),
                ("setup_ms", C.c_double), ("total_ms", C.c_double), ("finest_pixels", C.c_longlong)]


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


CAL_RAW, CAL_TEMP, CAL_REF, CAL_BRIT = 0, 1, 2, 3
TRANSPORT_INPLACE, TRANSPORT_COPY, TRANSPORT_COLLECTIVE = 0, 1, 2
TRANSPORT_NAMES = ("inplace", "copy", "collective")


class TransportInfo(C.Structure):
    """octane_vof_transport_info: what the first-contact self-check of a row-band plan decided (include/octane_vof.h)."""
    _fields_ = [("transport", C.c_int), ("q_dma", C.c_int), ("selfcheck", C.c_int), ("candidates_tried", C.c_int), ("forced", C.c_int),
                ("peer_ok", C.c_int), ("ndevices", C.c_int), ("nbands", C.c_int), ("check_rel_l2", C.c_double * 4), ("exchange", C.c_char * 48)]

    def as_dict(self):
        return {"transport_used": TRANSPORT_NAMES[self.transport] if 0 <= self.transport < 3 else str(self.transport),
                "q_dma": bool(self.q_dma), "selfcheck": {0: "not run", 1: "first candidate passed", 2: "downgraded", -1: "failed"}.get(self.selfcheck, self.selfcheck),
                "candidates_tried": self.candidates_tried, "forced": bool(self.forced), "peer_ok": bool(self.peer_ok),
                "devices": self.ndevices, "bands": self.nbands, "check_rel_l2": [x for x in self.check_rel_l2 if x >= 0],
                "exchange": self.exchange.decode() or None}


class Xfer(C.Structure):
    _fields_ = [("peer", C.c_int), ("send", C.c_int), ("buf", C.c_void_p), ("bytes", C.c_size_t)]


XCHG_ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t)
XCHG_SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Xfer))
ALLGATHER_BYTES_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class Exchange(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", XCHG_ALL_GATHER_FN), ("sendrecv", XCHG_SENDRECV_FN), ("name", C.c_char * 48)]

TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                       C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int)

_lib = None


class ProjNavcalParams(C.Structure):
    _fields_ = [("xScale", C.c_float), ("xOffset", C.c_float), ("yScale", C.c_float), ("yOffset", C.c_float),
                ("lon0", C.c_float), ("lat1", C.c_float), ("R", C.c_float), ("donav", C.c_int), ("mode", C.c_int),
                ("minx", C.c_int), ("maxx", C.c_int), ("miny", C.c_int), ("maxy", C.c_int)]


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so
    (SONAME libamdhip64.so.7) but asks for it as "libamdhip64.so"; if this library pulled in
    /opt/rocm's copy first, a later `import torch` would map a second runtime and see no GPU.
    Loading torch's copy by path first makes both resolve to the same object, whatever the
    import order (torch tensors and streams are handed to this library in bench.py/tests)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):  # pragma: no cover
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


STREAM_OWN = (1 << 64) - 1     # OCTANE_STREAM_OWN, (void *)-1


class OctaneError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = ""
        try:
            msg = lib().octane_last_error().decode()
        except Exception:  # pragma: no cover
            pass
        super().__init__(f"{where} failed with code {code}: {msg}")


def lib() -> C.CDLL:
    """Load liboctane_vof.so; raises (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C octane_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.octane_vof_default_params.argtypes = [C.POINTER(VofParams)]
    L.octane_vof_default_params.restype = None
    L.octane_vof_run.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(VofParams)]
    L.octane_vof_solve.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(VofParams)]
    L.octane_vof_release_cache.restype = None
    L.octane_vof_plan_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(VofParams)]
    L.octane_vof_plan_destroy.argtypes = [vp]
    L.octane_vof_plan_device_bytes.argtypes = [vp]
    L.octane_vof_plan_device_bytes.restype = C.c_size_t
    L.octane_vof_plan_placement_trials.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    L.octane_vof_plan_placement_trials.restype = C.c_int
    L.octane_vof_plan_run.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
    L.octane_vof_plan_solve.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
    L.octane_vof_plan_wait.argtypes = [vp]
    L.octane_vof_plan_last_iterations.argtypes = [vp]
    L.octane_vof_plan_last_iterations.restype = C.c_longlong
    L.octane_vof_plan_set_trace.argtypes = [vp, TRACE_FN, vp]
    L.octane_vof_plan_set_profiling.argtypes = [vp, C.c_int]
    L.octane_vof_plan_get_profile.argtypes = [vp, C.POINTER(VofProfile)]
    L.octane_vof_plan_get_launch_times.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.octane_vof_plan_set_lane_mode.argtypes = [vp, C.c_int]
    if hasattr(L, "octane_vof_tune"):          # the diagnostic library (include/octane_vof_dev.h, OCTANE_DIAG section)
        L.octane_vof_tune.argtypes = [vp, C.c_char_p, C.c_int]
        L.octane_vof_plan_probe.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.octane_selftest_assembly_math.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_ulonglong)]
        L.octane_selftest_assembly_math_bits.argtypes = [C.c_int, C.c_double]
        L.octane_selftest_rcp.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
        L.octane_vof_mid_geometry.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.octane_vof_row_rotation.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.octane_vof_batch_run.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int,
                                       C.POINTER(vp), C.POINTER(vp), C.POINTER(VofParams), C.c_int,
                                       C.POINTER(C.c_int)]
    L.octane_vof_tiled_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(VofParams), C.c_int,
                                          C.POINTER(C.c_int), C.c_longlong]
    L.octane_vof_tiled_destroy.argtypes = [vp]
    L.octane_vof_tiled_load.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.octane_vof_tiled_solve.argtypes = [vp]
    L.octane_vof_tiled_wait.argtypes = [vp]
    L.octane_vof_tiled_fetch.argtypes = [vp, vp, vp, C.c_int]
    L.octane_vof_tiled_run.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.octane_vof_tiled_banded_levels.argtypes = [vp]
    L.octane_vof_tiled_band_rows.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.octane_vof_tiled_last_iterations.argtypes = [vp]
    L.octane_vof_tiled_last_iterations.restype = C.c_longlong
    L.octane_vof_tiled_last_copies.argtypes = [vp]
    L.octane_vof_tiled_last_copies.restype = C.c_longlong
    L.octane_vof_tiled_device_bytes.argtypes = [vp]
    L.octane_vof_tiled_device_bytes.restype = C.c_size_t
    L.octane_vof_band_partition.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.octane_vof_mp_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(VofParams), C.c_int, C.c_int, C.c_longlong, C.c_char_p]
    L.octane_vof_mp_handles.argtypes = [vp, vp]
    L.octane_vof_mp_connect.argtypes = [vp, vp]
    L.octane_vof_mp_run.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.octane_vof_mp_banded_levels.argtypes = [vp]
    L.octane_vof_mp_last_iterations.argtypes = [vp]
    L.octane_vof_mp_last_iterations.restype = C.c_longlong
    L.octane_vof_mp_destroy.argtypes = [vp]
    L.octane_vof_tiled_transport_info.argtypes = [vp, C.POINTER(TransportInfo)]
    L.octane_vof_mp_transport_info.argtypes = [vp, C.POINTER(TransportInfo)]
    L.octane_vof_transport_name.argtypes = [C.c_int]
    L.octane_vof_transport_name.restype = C.c_char_p
    L.octane_vof_mp_set_exchange.argtypes = [vp, C.POINTER(Exchange)]
    L.octane_vof_mp_selfcheck.argtypes = [vp, ALLGATHER_BYTES_FN, vp]
    L.octane_pix2uv_run.argtypes = [C.POINTER(Nav), C.c_double, C.c_double, vp, vp, C.c_int, C.c_int,
                                    vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]
    L.octane_navcal_run.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(NavcalParams), vp, vp, vp, vp, vp, vp, C.c_int]
    L.octane_proj_navcal_run.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(ProjNavcalParams), vp, vp, vp, vp, vp, vp, C.c_int]
    L.octane_sosm_run.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int]
    L.octane_bandminmax.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.octane_uv2pix_run.argtypes = [C.POINTER(Nav), C.c_double, C.c_double, vp, vp, vp, vp, vp, vp, C.c_int]
    L.octane_srsal_run.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.octane_last_error.restype = C.c_char_p
    L.octane_device_count.restype = C.c_int
    _lib = L
    return L


@dataclass
class FlowParams:
    """The OFFlags fields the solver reads, CLI defaults of ref src/main.cc:78-96."""
    alpha: float = 5.0
    lambda_: float = 1.0
    lambdac: float = 0.0
    scaleF: float = 0.5
    scsig: float = 400.0
    kiters: int = 4
    liters: int = 3
    cgiters: int = 30
    dozim: int = 1
    device: int = 0

    def c(self) -> VofParams:
        return VofParams(self.alpha, self.lambda_, self.lambdac, self.scaleF, self.scsig,
                         self.kiters, self.liters, self.cgiters, self.dozim, self.device)


def default_params() -> FlowParams:
    p = VofParams()
    lib().octane_vof_default_params(C.byref(p))
    return FlowParams(p.alpha, p.lambda_, p.lambdac, p.scaleF, p.scsig, p.kiters, p.liters, p.cgiters, p.dozim, p.device)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Plan:
    """Device state for one (nx, ny, nchan, params); reuse it across image pairs."""

    def __init__(self, nx: int, ny: int, nchan: int = 1, params: FlowParams | None = None):
        self.nx, self.ny, self.nchan = nx, ny, nchan
        self.params = params or FlowParams()
        self._h = C.c_void_p()
        self._keep = None
        p = self.params.c()
        rc = lib().octane_vof_plan_create(C.byref(self._h), nx, ny, nchan, C.byref(p))
        if rc != OK:
            raise OctaneError(rc, "octane_vof_plan_create")

    def close(self):
        if self._h:
            lib().octane_vof_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return lib().octane_vof_plan_device_bytes(self._h)

    def placement_trials(self):
        """ms per finest-level PCG iteration of every candidate arena timed at creation ([] when none were)."""
        buf = (C.c_double * 8)()
        n = lib().octane_vof_plan_placement_trials(self._h, buf, 8)
        return [buf[i] for i in range(min(n, 8))]

    def run_host(self, img1, img2, u0=None, v0=None):
        """Host numpy buffers in, (u, v) numpy out.  img: [nchan, ny, nx] or [ny, nx]."""
        a, b = _f32(img1), _f32(img2)
        assert a.size == self.nchan * self.ny * self.nx and b.size == a.size
        u = np.zeros((self.ny, self.nx), np.float32) if u0 is None else np.array(u0, np.float32, order="C", copy=True)
        v = np.zeros((self.ny, self.nx), np.float32) if v0 is None else np.array(v0, np.float32, order="C", copy=True)
        rc = lib().octane_vof_plan_run(self._h, _ptr(a), _ptr(b), _ptr(u), _ptr(v), MEM_HOST, None)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_plan_run")
        return u, v

    def run_device(self, img1_ptr: int, img2_ptr: int, u_ptr: int, v_ptr: int, stream: int = 0):
        """Device pointers (dense [nchan, ny, nx] / [ny, nx] float32); enqueues on `stream` (a hipStream_t, 0 = the null
        stream, STREAM_OWN = the plan's private stream: inputs must be complete, wait() tells when the outputs are)."""
        rc = lib().octane_vof_plan_run(self._h, C.c_void_p(img1_ptr), C.c_void_p(img2_ptr), C.c_void_p(u_ptr),
                                       C.c_void_p(v_ptr), MEM_DEVICE, C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_plan_run")

    def solve_device(self, img1_ptr: int, img2_ptr: int, u_out_ptr: int, v_out_ptr: int, u0_ptr: int = 0, v0_ptr: int = 0,
                     stream: int = 0):
        """Like run_device with the first guess (0 = none: zero flow) and the result in separate device buffers."""
        rc = lib().octane_vof_plan_solve(self._h, C.c_void_p(img1_ptr), C.c_void_p(img2_ptr),
                                         C.c_void_p(u0_ptr) if u0_ptr else None, C.c_void_p(v0_ptr) if v0_ptr else None,
                                         C.c_void_p(u_out_ptr), C.c_void_p(v_out_ptr), MEM_DEVICE,
                                         C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_plan_solve")

    def wait(self):
        rc = lib().octane_vof_plan_wait(self._h)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_plan_wait")

    def last_iterations(self) -> int:
        return int(lib().octane_vof_plan_last_iterations(self._h))

    def set_trace(self, store: dict | None):
        if store is None:
            lib().octane_vof_plan_set_trace(self._h, C.cast(None, TRACE_FN), None)
            self._keep = None
            return

        def _cb(user, tag, k, gnc, l, data, nx, ny, npl):
            store[(tag.decode(), k, gnc, l)] = np.ctypeslib.as_array(data, shape=(npl, ny, nx)).copy()
        self._keep = TRACE_FN(_cb)
        lib().octane_vof_plan_set_trace(self._h, self._keep, None)

    def set_profiling(self, on: bool):
        lib().octane_vof_plan_set_profiling(self._h, 1 if on else 0)

    def set_lane_mode(self, mode: int):
        """How many other plans share the device: 0 none, 2 one (the two lanes of a batch), 1 two or more (include/octane_vof.h)."""
        rc = lib().octane_vof_plan_set_lane_mode(self._h, mode)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_plan_set_lane_mode")

    def tune(self, key: str, value: int):
        """Developer knob: DIAGNOSTIC library only (capi.diag().Plan(...).tune(...)); the product library does not export it."""
        L = lib()
        if not hasattr(L, "octane_vof_tune"):
            raise AttributeError("octane_vof_tune is exported by the diagnostic library only: bind it with octane_amd.capi.diag()")
        rc = L.octane_vof_tune(self._h, key.encode(), value)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tune")

    def probe(self, level: int, iterations: int = 20):
        """(pass A ms, pass B ms) of one pyramid level timed in isolation (diagnostic library only; clobbers the planes)."""
        a, b = C.c_double(), C.c_double()
        if not hasattr(lib(), "octane_vof_plan_probe"):
            raise AttributeError("octane_vof_plan_probe is exported by the diagnostic library only: bind it with octane_amd.capi.diag()")
        rc = lib().octane_vof_plan_probe(self._h, level, iterations, C.byref(a), C.byref(b))
        if rc != OK:
            raise OctaneError(rc, "octane_vof_plan_probe")
        return a.value, b.value

    def launch_times(self):
        """ms of every finest-level PCG launch of the last profiled run, in launch order."""
        n = lib().octane_vof_plan_get_launch_times(self._h, None, 0)
        if n <= 0:
            return []
        buf = (C.c_float * n)()
        lib().octane_vof_plan_get_launch_times(self._h, buf, n)
        return list(buf)

    def profile(self) -> VofProfile:
        p = VofProfile()
        lib().octane_vof_plan_get_profile(self._h, C.byref(p))
        return p


def band_partition(rows: int, nbands: int):
    """Row-band edges [0, ..., rows] of a level with `rows` rows, or None when it stays replicated (no GPU needed)."""
    e = (C.c_int * (nbands + 1))()
    rc = lib().octane_vof_band_partition(rows, nbands, e)
    if rc < 0:
        raise OctaneError(rc, "octane_vof_band_partition")
    return list(e) if rc == 1 else None


def band_devices(nbands: int):
    """Device of every band for a row-band solve on THIS box: with two or more GPUs visible the bands go round-robin over
    real devices (min(nbands, count) of them, so that peer reads, cross-device events and peer copies over xGMI are what
    runs); on a one-GPU box every band is a virtual band on device 0.  `OCTANE_TEST_DEVICES` (e.g. "0,0,1,1") overrides."""
    env = os.environ.get("OCTANE_TEST_DEVICES")
    if env:
        ids = [int(t) for t in env.split(",") if t.strip() != ""]
        return [ids[b % len(ids)] for b in range(nbands)]
    n = int(lib().octane_device_count())
    if n < 2:
        return [0] * nbands
    use = min(n, nbands)
    return [b % use for b in range(nbands)]


class TiledPlan:
    """One frame solved by `nbands` row bands, band b on devices[b] (ids may repeat: virtual bands on one GPU).
    Levels below `min_band_pixels` are solved redundantly by every band (include/octane_vof.h)."""

    def __init__(self, nx: int, ny: int, nchan: int = 1, params: FlowParams | None = None, nbands: int = 2,
                 devices=None, min_band_pixels: int = 0):
        self.nx, self.ny, self.nchan, self.nbands = nx, ny, nchan, nbands
        self.params = params or FlowParams()
        self._h = C.c_void_p()
        p = self.params.c()
        dv = None
        if devices is not None:
            assert len(devices) == nbands
            dv = (C.c_int * nbands)(*devices)
        rc = lib().octane_vof_tiled_create(C.byref(self._h), nx, ny, nchan, C.byref(p), nbands, dv, min_band_pixels)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tiled_create")

    def close(self):
        if self._h:
            lib().octane_vof_tiled_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def banded_levels(self) -> int:
        return int(lib().octane_vof_tiled_banded_levels(self._h))

    @property
    def device_bytes(self) -> int:
        return lib().octane_vof_tiled_device_bytes(self._h)

    def band_rows(self, level: int, band: int):
        """(banded?, y0, y1) of `band` at pyramid level `level` (0 = coarsest)."""
        y0, y1 = C.c_int(), C.c_int()
        rc = lib().octane_vof_tiled_band_rows(self._h, level, band, C.byref(y0), C.byref(y1))
        if rc < 0:
            raise OctaneError(rc, "octane_vof_tiled_band_rows")
        return bool(rc), y0.value, y1.value

    def run_host(self, img1, img2, u0=None, v0=None):
        a, b = _f32(img1), _f32(img2)
        assert a.size == self.nchan * self.ny * self.nx and b.size == a.size
        u = np.zeros((self.ny, self.nx), np.float32) if u0 is None else np.array(u0, np.float32, order="C", copy=True)
        v = np.zeros((self.ny, self.nx), np.float32) if v0 is None else np.array(v0, np.float32, order="C", copy=True)
        rc = lib().octane_vof_tiled_run(self._h, _ptr(a), _ptr(b), _ptr(u), _ptr(v), MEM_HOST)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tiled_run")
        return u, v

    def load_device(self, img1_ptr: int, img2_ptr: int, u_ptr: int, v_ptr: int):
        rc = lib().octane_vof_tiled_load(self._h, C.c_void_p(img1_ptr), C.c_void_p(img2_ptr), C.c_void_p(u_ptr),
                                         C.c_void_p(v_ptr), MEM_DEVICE)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tiled_load")

    def solve(self):
        rc = lib().octane_vof_tiled_solve(self._h)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tiled_solve")

    def wait(self):
        rc = lib().octane_vof_tiled_wait(self._h)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tiled_wait")

    def fetch_device(self, u_ptr: int, v_ptr: int):
        rc = lib().octane_vof_tiled_fetch(self._h, C.c_void_p(u_ptr), C.c_void_p(v_ptr), MEM_DEVICE)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tiled_fetch")

    def last_iterations(self) -> int:
        return int(lib().octane_vof_tiled_last_iterations(self._h))

    def last_copies(self) -> int:
        return int(lib().octane_vof_tiled_last_copies(self._h))

    def transport_info(self) -> dict:
        """Which transport the bands use and what the first-contact self-check found (octane_vof_transport_info)."""
        ti = TransportInfo()
        rc = lib().octane_vof_tiled_transport_info(self._h, C.byref(ti))
        if rc != OK:
            raise OctaneError(rc, "octane_vof_tiled_transport_info")
        return ti.as_dict()


MP_HANDLE_BYTES = 128


class MpPlan:
    """One band of a row-band solve with one band per process (octane_vof_mp_*).  Every method is collective over the
    `world` ranks.  `all_gather` is a callable taking this rank's bytes and returning the list of every rank's bytes in
    rank order (e.g. a wrapper of torch.distributed.all_gather_object).  `exchange` (optional; octane_amd.exchange.TorchExchange)
    is the host program's collective library for the collective transport: with it the bands still solve the frame where HIP IPC
    is not available, and the first-contact self-check (run here unless selfcheck=False) may fall back to it."""

    def __init__(self, nx: int, ny: int, nchan: int, params: FlowParams, rank: int, world: int, shm_name: str, all_gather,
                 min_band_pixels: int = 0, exchange=None, selfcheck: bool = True):
        self.nx, self.ny, self.nchan, self.rank, self.world = nx, ny, nchan, rank, world
        self._h = C.c_void_p()
        self._exchange = exchange
        p = params.c()
        import time as _time
        _t0 = _time.perf_counter()

        def _note(what):     # OCTANE_MP_TRACE=1: where a creation that takes long spends its time (every rank speaks)
            if os.environ.get("OCTANE_MP_TRACE") == "1":
                import sys as _sys
                print(f"octane MpPlan rank {rank} [{_time.perf_counter() - _t0:7.2f} s] {what}", file=_sys.stderr, flush=True)
        rc = lib().octane_vof_mp_create(C.byref(self._h), nx, ny, nchan, C.byref(p), rank, world, min_band_pixels, shm_name.encode())
        if rc != OK:
            raise OctaneError(rc, "octane_vof_mp_create")
        _note("band plan created")
        if exchange is not None:
            rc = lib().octane_vof_mp_set_exchange(self._h, C.byref(exchange.c_struct()))
            if rc != OK:
                raise OctaneError(rc, "octane_vof_mp_set_exchange")
        buf = C.create_string_buffer(MP_HANDLE_BYTES)
        rc = lib().octane_vof_mp_handles(self._h, buf)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_mp_handles")
        blobs = all_gather(buf.raw)
        assert len(blobs) == world and all(len(b) == MP_HANDLE_BYTES for b in blobs)
        _note("handles gathered")
        rc = lib().octane_vof_mp_connect(self._h, C.create_string_buffer(b"".join(blobs), MP_HANDLE_BYTES * world))
        if rc != OK:
            raise OctaneError(rc, "octane_vof_mp_connect")
        _note("connected (IPC mappings opened or declined)")

        def _ag(user, mine, allp, nbytes):          # the library's byte all-gather, carried by the host program's
            try:
                got = all_gather(C.string_at(mine, nbytes))
                C.memmove(allp, b"".join(got), nbytes * world)
                return 0
            except Exception as e:  # pragma: no cover
                print(f"octane MpPlan: all-gather callback failed: {e!r}", flush=True)
                return 1
        self._ag = ALLGATHER_BYTES_FN(_ag)
        if selfcheck:
            rc = lib().octane_vof_mp_selfcheck(self._h, self._ag, None)
            if rc != OK:
                raise OctaneError(rc, "octane_vof_mp_selfcheck")
            _note("first-contact self-check done")

    def transport_info(self) -> dict:
        ti = TransportInfo()
        rc = lib().octane_vof_mp_transport_info(self._h, C.byref(ti))
        if rc != OK:
            raise OctaneError(rc, "octane_vof_mp_transport_info")
        return ti.as_dict()

    def close(self):
        if self._h:
            lib().octane_vof_mp_destroy(self._h)
            self._h = C.c_void_p()

    @property
    def banded_levels(self) -> int:
        return int(lib().octane_vof_mp_banded_levels(self._h))

    def last_iterations(self) -> int:
        return int(lib().octane_vof_mp_last_iterations(self._h))

    def run_host(self, img1, img2, u0=None, v0=None):
        """Whole pair on every rank (numpy); returns (u, v) on rank 0, (None, None) elsewhere."""
        a, b = _f32(img1), _f32(img2)
        u = np.zeros((self.ny, self.nx), np.float32)
        v = np.zeros((self.ny, self.nx), np.float32)
        g0 = _f32(u0) if u0 is not None else None
        g1 = _f32(v0) if v0 is not None else None
        rc = lib().octane_vof_mp_run(self._h, _ptr(a), _ptr(b), _ptr(g0) if g0 is not None else None,
                                     _ptr(g1) if g1 is not None else None, _ptr(u), _ptr(v), MEM_HOST)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_mp_run")
        return (u, v) if self.rank == 0 else (None, None)

    def run_device(self, img1_ptr: int, img2_ptr: int, u_out_ptr: int, v_out_ptr: int, u0_ptr: int = 0, v0_ptr: int = 0):
        """Dense device buffers on this rank's device (complete when the call is made); blocking."""
        rc = lib().octane_vof_mp_run(self._h, C.c_void_p(img1_ptr), C.c_void_p(img2_ptr), C.c_void_p(u0_ptr) if u0_ptr else None,
                                     C.c_void_p(v0_ptr) if v0_ptr else None, C.c_void_p(u_out_ptr), C.c_void_p(v_out_ptr), MEM_DEVICE)
        if rc != OK:
            raise OctaneError(rc, "octane_vof_mp_run")


def flow(img1, img2, params: FlowParams | None = None, u0=None, v0=None):
    """One-shot solve through octane_vof_run (allocate, upload, solve, download, free)."""
    a, b = _f32(img1), _f32(img2)
    if a.ndim == 2:
        a, b = a[None], b[None]
    nc, ny, nx = a.shape
    u = np.zeros((ny, nx), np.float32) if u0 is None else np.array(u0, np.float32, order="C", copy=True)
    v = np.zeros((ny, nx), np.float32) if v0 is None else np.array(v0, np.float32, order="C", copy=True)
    p = (params or FlowParams()).c()
    rc = lib().octane_vof_run(_ptr(a), _ptr(b), nx, ny, nc, _ptr(u), _ptr(v), C.byref(p))
    if rc != OK:
        raise OctaneError(rc, "octane_vof_run")
    return u, v


def flow_inplace(img1, img2, u, v, params: FlowParams | None = None) -> None:
    """octane_vof_run on the caller's own buffers, exactly as the C++ shim calls it: u / v (C-contiguous float32 [ny, nx])
    hold the first guess on entry and the flow on return; nothing is copied or allocated on the host side."""
    a, b = img1, img2
    for x in (a, b, u, v):
        if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags["C_CONTIGUOUS"]):
            raise ValueError("flow_inplace needs C-contiguous float32 arrays")
    if a.ndim == 2:
        a, b = a[None], b[None]
    nc, ny, nx = a.shape
    p = (params or FlowParams()).c()
    rc = lib().octane_vof_run(_ptr(a), _ptr(b), nx, ny, nc, _ptr(u), _ptr(v), C.byref(p))
    if rc != OK:
        raise OctaneError(rc, "octane_vof_run")


def flow_into(img1, img2, u, v, params: FlowParams | None = None) -> None:
    """One-shot solve with the zero first guess NOT uploaded (octane_vof_solve with u0 = v0 = NULL): u / v are written only."""
    a, b = _f32(img1), _f32(img2)
    if a.ndim == 2:
        a, b = a[None], b[None]
    nc, ny, nx = a.shape
    p = (params or FlowParams()).c()
    rc = lib().octane_vof_solve(_ptr(a), _ptr(b), nx, ny, nc, None, None, _ptr(u), _ptr(v), C.byref(p))
    if rc != OK:
        raise OctaneError(rc, "octane_vof_solve")


def release_cache() -> None:
    """Free the plan octane_vof_run keeps between calls."""
    lib().octane_vof_release_cache()


def batch_flow(pairs, params: FlowParams | None = None, devices=None):
    """pairs: list of (img1, img2) host arrays of one shape.  Returns list of (u, v)."""
    if not pairs:
        return []
    a0 = _f32(pairs[0][0])
    if a0.ndim == 2:
        a0 = a0[None]
    nc, ny, nx = a0.shape
    n = len(pairs)
    ims1 = [_f32(p[0]) for p in pairs]
    ims2 = [_f32(p[1]) for p in pairs]
    us = [np.zeros((ny, nx), np.float32) for _ in range(n)]
    vs = [np.zeros((ny, nx), np.float32) for _ in range(n)]
    arr = C.c_void_p * n
    devices = list(devices) if devices is not None else list(range(max(1, lib().octane_device_count())))
    dv = (C.c_int * len(devices))(*devices)
    p = (params or FlowParams()).c()
    rc = lib().octane_vof_batch_run(n, arr(*[_ptr(x) for x in ims1]), arr(*[_ptr(x) for x in ims2]), nx, ny, nc,
                                    arr(*[_ptr(x) for x in us]), arr(*[_ptr(x) for x in vs]), C.byref(p),
                                    len(devices), dv)
    if rc != OK:
        raise OctaneError(rc, "octane_vof_batch_run")
    return list(zip(us, vs))


def pix2uv(nav: Nav, t1: float, t2: float, u, v, pixuv: int = 0, mode: int = NAV_GEOS, device: int = 0):
    """Returns (ur, vr, ur2, vr2, dT, sector_moved) -- shorts in cm/s (x100)."""
    uu, vv = _f32(u), _f32(v)
    n = uu.size
    ur, vr, ur2, vr2 = (np.zeros(n, np.int16) for _ in range(4))
    dT, moved = C.c_float(), C.c_int()
    rc = lib().octane_pix2uv_run(C.byref(nav), t1, t2, _ptr(uu), _ptr(vv), pixuv, mode, _ptr(ur), _ptr(vr),
                                 _ptr(ur2), _ptr(vr2), C.byref(dT), C.byref(moved), device)
    if rc != OK:
        raise OctaneError(rc, "octane_pix2uv_run")
    shp = uu.shape
    return ur.reshape(shp), vr.reshape(shp), ur2.reshape(shp), vr2.reshape(shp), dT.value, moved.value


def bandminmax(band: int):
    """(max, min) radiance of an ABI band, ref src/oct_normalize_geo.cc:9-88."""
    mx, mn = C.c_float(), C.c_float()
    rc = lib().octane_bandminmax(band, C.byref(mx), C.byref(mn))
    if rc != OK:
        raise OctaneError(rc, "octane_bandminmax")
    return mx.value, mn.value


def navcal(data2, x, y, prm: NavcalParams, device: int = 0):
    """Raw counts [ny, nx] int16 + scaled grid coordinates -> (data3, lat, lon, data2s, xs, ys) of prm's window."""
    d2 = np.ascontiguousarray(data2, np.int16)
    ny, nx = d2.shape
    xx, yy = np.ascontiguousarray(x, np.int16), np.ascontiguousarray(y, np.int16)
    ww, wh = prm.maxx - prm.minx, prm.maxy - prm.miny
    data3, lat, lon = (np.zeros((max(wh, 0), max(ww, 0)), np.float32) for _ in range(3))
    d2s = np.zeros((max(wh, 0), max(ww, 0)), np.int16)
    xs, ys = np.zeros(max(ww, 0), np.int16), np.zeros(max(wh, 0), np.int16)
    rc = lib().octane_navcal_run(_ptr(d2), _ptr(xx), _ptr(yy), nx, ny, C.byref(prm), _ptr(data3), _ptr(lat), _ptr(lon),
                                 _ptr(d2s), _ptr(xs), _ptr(ys), device)
    if rc != OK:
        raise OctaneError(rc, "octane_navcal_run")
    return data3, lat, lon, d2s, xs, ys


def sosm(img1, img2, rad: int = 2, srad: int = 2, u0=None, v0=None, device: int = 0):
    """Patch-matching flow (-sosm) of one-channel images [ny, nx]; returns (u, v)."""
    a, b = _f32(img1), _f32(img2)
    ny, nx = a.shape
    u = np.zeros((ny, nx), np.float32) if u0 is None else np.array(u0, np.float32, order="C", copy=True)
    v = np.zeros((ny, nx), np.float32) if v0 is None else np.array(v0, np.float32, order="C", copy=True)
    rc = lib().octane_sosm_run(_ptr(a), _ptr(b), nx, ny, _ptr(u), _ptr(v), rad, srad, device)
    if rc != OK:
        raise OctaneError(rc, "octane_sosm_run")
    return u, v


def proj_navcal(data2, x, y, prm: ProjNavcalParams, device: int = 0):
    """Polar / mercator navigation of a re-mapped float image [ny, nx] -> (data3, lat, lon, data2s, xs, ys) of prm's window."""
    d2 = np.ascontiguousarray(data2, np.float32)
    ny, nx = d2.shape
    xx, yy = np.ascontiguousarray(x, np.int16), np.ascontiguousarray(y, np.int16)
    ww, wh = prm.maxx - prm.minx, prm.maxy - prm.miny
    data3, lat, lon = (np.zeros((max(wh, 0), max(ww, 0)), np.float32) for _ in range(3))
    d2s = np.ones((max(wh, 0), max(ww, 0)), np.int16)
    xs, ys = np.zeros(max(ww, 0), np.int16), np.zeros(max(wh, 0), np.int16)
    rc = lib().octane_proj_navcal_run(_ptr(d2), _ptr(xx), _ptr(yy), nx, ny, C.byref(prm), _ptr(data3), _ptr(lat), _ptr(lon),
                                      _ptr(d2s), _ptr(xs), _ptr(ys), device)
    if rc != OK:
        raise OctaneError(rc, "octane_proj_navcal_run")
    return data3, lat, lon, d2s, xs, ys


def uv2pix(nav: Nav, t1: float, t2: float, u, v, lat, lon, gx, gy, device: int = 0):
    """First-guess winds (m/s) -> pixel displacements; returns new (u, v)."""
    uu = np.array(u, np.float32, order="C", copy=True); vv = np.array(v, np.float32, order="C", copy=True)
    la, lo = _f32(lat), _f32(lon)
    xx, yy = np.ascontiguousarray(gx, np.int16), np.ascontiguousarray(gy, np.int16)
    rc = lib().octane_uv2pix_run(C.byref(nav), t1, t2, _ptr(uu), _ptr(vv), _ptr(la), _ptr(lo), _ptr(xx), _ptr(yy), device)
    if rc != OK:
        raise OctaneError(rc, "octane_uv2pix_run")
    return uu, vv


def srsal(u, v, cth, device: int = 0):
    """37x37 bilateral smoothing of the flow guided by cth; returns new (u, v)."""
    uu = np.array(u, np.float32, order="C", copy=True); vv = np.array(v, np.float32, order="C", copy=True)
    cc = _f32(cth)
    ny, nx = uu.shape
    rc = lib().octane_srsal_run(_ptr(uu), _ptr(vv), _ptr(cc), nx, ny, device)
    if rc != OK:
        raise OctaneError(rc, "octane_srsal_run")
    return uu, vv
