"""pytest configuration: `-m "not gpu"` runs on the CPU-only build box (oracle, host logic, ABI, gloo);
`-m gpu` runs on an MI355X and goes through the C ABI of libgprc_native.so."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOL = 1e-10  # BASELINE.json north_star: normwise max|d| / max|ref| on mean and variance, fp64


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def _ensure_built():
    import gprc_amd  # noqa: F401  (alias loader)
    from gprc_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    from oracle import oracle
    oracle.build()


_ensure_built()


def gpu_available():
    from gprc_amd import _native
    try:
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if gpu_available():
        return
    skip = pytest.mark.skip(reason="no MI355X visible (the native path has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    def __init__(self):
        z = np.load(os.path.join(ROOT, "tests", "golden", "gprc_golden.npz"), allow_pickle=False)
        self.z = z
        self.cases = json.loads(bytes(z["manifest"]).decode())

    def get(self, case, key):
        return self.z[f"{case['name']}/{key}"]

    def of_type(self, t):
        return [c for c in self.cases if c["type"] == t]


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    return oracle


def nerr(got, ref):
    """Normwise error max|got - ref| / max|ref| (the tolerance definition of DESIGN.md)."""
    got, ref = np.asarray(got, dtype=float), np.asarray(ref, dtype=float)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all()
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300))


KERNEL_ARG_ORDER = {"constant": ("c",), "linear": ("sigma",), "polynomial": ("sigma", "p"), "sqrexp": ("l",),
                    "gammaexp": ("l", "gamma"), "rationalquadratic": ("l", "alpha")}


def oracle_params(kind, params):
    out = []
    for a in KERNEL_ARG_ORDER[kind]:
        out.extend(np.atleast_1d(np.asarray(params[a], dtype=float)).tolist())
    return out
