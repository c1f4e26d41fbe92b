import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def pytest_collection_modifyitems(config, items):
    """The full-size oracle cases (tests/test_gpu_fullsize.py) run LAST: their CPU oracle legs are computed by a worker thread from
    collection time on, under the GPU work of every other module (see that file's CASES)."""
    last = [it for it in items if it.nodeid.split("::")[0].endswith("test_gpu_fullsize.py")]
    if last:
        items[:] = [it for it in items if it not in last] + last


def pytest_collection_finish(session):
    if session.config.option.collectonly or os.environ.get("OCT_NO_ORACLE_PREFETCH") == "1":
        return
    names = [getattr(getattr(it, "function", None), "_oracle_case", None) for it in session.items]
    names = [n for n in names if n]
    if not names:
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return                               # the tests will skip / fail by themselves; nothing to compute ahead
        from oracle import oct_oracle
        oct_oracle.build()
        mod = next(it.module for it in session.items if getattr(getattr(it, "function", None), "_oracle_case", None))
        mod.PREFETCH.start(oct_oracle, names)
    except Exception as e:                       # noqa: BLE001 -- the cases are then computed in place
        print(f"conftest: oracle prefetch not started ({e!r})", file=sys.stderr)


def rel_l2(u, v, uo, vo):
    """Relative L2 distance of the flow field (u,v) from (uo,vo): the parity metric of
    BASELINE.json's north_star (bar: 1e-4)."""
    num = ((np.asarray(u, np.float64) - uo) ** 2 + (np.asarray(v, np.float64) - vo) ** 2).sum()
    den = (np.asarray(uo, np.float64) ** 2 + np.asarray(vo, np.float64) ** 2).sum()
    return float(np.sqrt(num / den)) if den > 0 else float(np.sqrt(num))


SANITIZE = os.environ.get("OCT_SANITIZE") == "1"      # `make sanitize`: host libraries and oracle built with ASan + UBSan (the process is preloaded)
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"] if SANITIZE else []


def host_libdir():
    """Where liboctane_host.so / liboctane_io.so are: octane_amd/, or octane_amd/_san/ under `make sanitize`."""
    return os.path.join(ROOT, "octane_amd", "_san") if SANITIZE else os.path.join(ROOT, "octane_amd")


def host_make_args():
    return ["SAN=1"] if SANITIZE else []


@pytest.fixture(scope="session")
def oracle():
    from oracle import oct_oracle
    oct_oracle.build()
    return oct_oracle


@pytest.fixture(scope="session")
def golden_ref():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_helpers.npz"))


@pytest.fixture(scope="session")
def golden_flow():
    return np.load(os.path.join(ROOT, "tests", "golden", "oracle_flow.npz"))


@pytest.fixture(scope="session")
def capi():
    """The product binding.  Builds the HIP library if it is not there yet (hipcc cross-compiles
    without a GPU)."""
    from octane_amd import capi as _capi
    if not os.path.exists(_capi.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "octane_amd", "csrc"), "-s"])
    return _capi


@pytest.fixture(scope="session")
def capi_diag(capi):
    """The same binding on the DIAGNOSTIC library (liboctane_vof_diag.so: the product's sources + -DOCTANE_DIAG=1): the only build that
    exports octane_vof_tune, the probes and the self-tests and contains the two-pass form of the PCG iteration.  Tests that compare
    kernel FORMS with each other run on it; every parity test against the oracle runs on the product library (`capi`)."""
    try:
        return capi.diag()
    except ImportError as e:
        pytest.skip(str(e))
