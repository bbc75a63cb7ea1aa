import glob
import os
import sys

import numpy as np
import pytest

# the library reads its test / A/B switches (MDT_CONFIG, MDT_WAVE, MDT_RANK_VARIANT, ...) only behind this gate, sampled when it
# is first used: a stray variable in a user's shell never changes which path serves a request
os.environ.setdefault("MDT_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(estimators=False, diag=False):
    """fixture names; the est_<name> files (ess(kind = estimator), mcse(std)) and the diag_<name> files (bfmi,
    gelmandiag) share the input of <name>"""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    if diag:
        return [n for n in names if n.startswith("diag_")]
    return [n for n in names if not n.startswith("diag_") and n.startswith("est_") == estimators]


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def ar1(phi, D, C, P, rng, sigma=None):
    """AR(1) chains with stationary N(0, 1) marginals (test/helpers.jl:4-20 recipe)."""
    sigma = np.sqrt(1 - phi ** 2) if sigma is None else sigma
    e = rng.standard_normal((D, C, P)) * sigma
    x = np.empty_like(e)
    x[0] = rng.standard_normal((C, P)) * (sigma / np.sqrt(1 - phi ** 2))
    for t in range(1, D):
        x[t] = phi * x[t - 1] + e[t]
    return x


def assert_close(a, b, rtol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), (what, a, b)
    m = ~np.isnan(a)
    inf = m & np.isinf(b)
    assert np.array_equal(a[inf], b[inf]), what
    m &= ~np.isinf(b)
    if m.any():
        err = np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300))
        assert err <= rtol, (what, err)


@pytest.fixture(scope="session")
def rng():
    return np.random.default_rng(1)
