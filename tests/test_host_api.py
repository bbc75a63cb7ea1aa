"""CPU: host-side logic of the API mirror that runs before anything touches the GPU --
argument validation with the reference's error behaviour, superchain bookkeeping, sharding."""
import numpy as np
import pytest

import __graft_entry__ as g


@pytest.fixture(scope="module")
def pkg():
    return g.load_package()


def test_unknown_kind_raises_argument_error(pkg):
    # test/ess_rhat.jl:94-97
    x = np.zeros((5, 3, 5))
    for f in (pkg.ess, pkg.ess_rhat, pkg.rhat):
        with pytest.raises(pkg.ArgumentError):
            f(x, kind="foo")
    with pytest.raises(pkg.ArgumentError):
        pkg.ess(x, kind=lambda v: v.mean())  # mymean: unsupported estimator
    with pytest.raises(pkg.ArgumentError):
        pkg.ess(x, kind="rank")  # `ess` has no :rank kind (src/ess_rhat.jl:279-286)


def test_autocov_method_types(pkg):
    assert issubclass(pkg.AutocovMethod, pkg.AbstractAutocovMethod)
    assert issubclass(pkg.FFTAutocovMethod, pkg.AbstractAutocovMethod)
    assert issubclass(pkg.BDAAutocovMethod, pkg.AbstractAutocovMethod)

    class Mine(pkg.AbstractAutocovMethod):
        pass
    with pytest.raises(pkg.ArgumentError):
        pkg.ess(np.zeros((100, 2, 1)), autocov_method=Mine())


def test_validate_superchain_ids(pkg):
    # src/rhat_nested.jl:68-81, test/rhat_nested.jl:39-60
    from mcmcdiagnostictools_jl_amd.api import _validate_superchain_ids as v
    inds = v([2, 1, 1, 2, 3, 3], 6)
    assert inds.shape == (2, 3) and inds.flags.f_contiguous
    assert inds.T.tolist() == [[1, 2], [0, 3], [4, 5]]  # sorted unique ids, ascending chain indices
    assert v(["b", "a", "b", "a"], 4).T.tolist() == [[1, 3], [0, 2]]
    with pytest.raises(pkg.DimensionMismatch):
        v([1, 2, 3], 4)
    with pytest.raises(pkg.ArgumentError):
        v([1, 1, 1, 1], 4)  # < 2 superchains
    with pytest.raises(pkg.ArgumentError):
        v([1, 1, 1, 2], 4)  # unequal sizes
    with pytest.raises(pkg.ArgumentError):
        pkg.rhat_nested(np.zeros(10), [1] * 10)  # needs >= 2 dims
    with pytest.raises(pkg.ArgumentError):
        pkg.rhat_nested(np.zeros((10, 4, 2)), [1, 1, 2, 2], kind="foo")


def test_keyword_types(pkg):
    with pytest.raises(TypeError):
        pkg.ess(np.zeros((100, 2, 1)), maxlag=2.5)
    with pytest.raises(TypeError):
        pkg.rhat(np.zeros((100, 2, 1)), split_chains=2.0)


def test_prepare_layout_and_shapes(pkg):
    from mcmcdiagnostictools_jl_amd.api import _prepare
    import torch
    x = np.arange(2 * 3 * 4 * 5, dtype=np.float64).reshape(2, 3, 4, 5)
    pr = _prepare(x)
    assert (pr.D, pr.C, pr.P) == (2, 3, 20) and pr.param_shape == (4, 5) and not pr.scalar
    # memory is (params, chains, draws): draws fastest, param (i, j) -> flat i*5 + j
    assert pr.buf.shape == (20, 3, 2)
    assert np.array_equal(pr.buf[7], x[:, :, 1, 2].T)
    assert _prepare(np.zeros(7)).scalar and _prepare(np.zeros((7, 2))).scalar
    assert _prepare(np.zeros(7, dtype=np.int64)).buf.dtype == np.float64  # Int -> Float64
    assert _prepare(np.zeros((7, 2, 1), dtype=np.float32)).f32
    t = torch.zeros(5, 4, 100).permute(2, 1, 0)  # already draws-fastest: no copy
    pr = _prepare(t)
    assert pr.buf.data_ptr() == t.data_ptr()
    xm = np.ma.MaskedArray(np.zeros((10, 2, 3)), mask=False)
    xm[0, 0, 1] = np.ma.masked
    assert _prepare(xm).missing.tolist() == [False, True, False]


def test_shard_bounds(pkg):
    for P, W in ((10, 2), (10, 3), (7, 8), (1_000_000, 8)):
        b = [pkg.shard_bounds(P, r, W) for r in range(W)]
        assert b[0][0] == 0 and b[-1][1] == P
        assert all(b[i][1] == b[i + 1][0] for i in range(W - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_library_beta_inverse_against_scipy(pkg):
    """csrc/mdt_beta.hpp (one source for host and device): I_x(a, b) and its inverse -- the quadrature points of
    src/mcse.jl:105-106 and the F quantile of src/gelmandiag.jl:50 -- against scipy.special, through the host-only
    diagnostics entry point (no GPU is touched)."""
    import ctypes
    from scipy.special import betainc, betaincinv
    lib = pkg._lib.load()
    out = ctypes.c_double()
    cases = [(S * p + 1, S * (1 - p) + 1) for S in (5.3, 40.0, 333.0, 2500.0, 19000.0) for p in (0.5, 0.1, 0.9, 0.01, 0.25)]
    cases += [(d1 / 2, d2 / 2) for d1 in (1, 3, 7, 15, 63) for d2 in (2.5, 30.0, 400.0, 5000.0, 40000.0, 1e6)]
    worst_inv = worst_fwd = 0.0
    for a, b in cases:
        for q in (0.15865525393145705, 0.8413447460685429, 0.975, 0.9, 0.5, 0.025):
            assert lib.mdt_debug_beta(a, b, q, 1, ctypes.byref(out)) == 0
            xs = betaincinv(a, b, q)
            worst_inv = max(worst_inv, abs(out.value / xs - 1))
            assert lib.mdt_debug_beta(a, b, xs, 0, ctypes.byref(out)) == 0
            worst_fwd = max(worst_fwd, abs(out.value / betainc(a, b, xs) - 1))
    assert worst_inv < 2e-10 and worst_fwd < 2e-10, (worst_inv, worst_fwd)
    assert lib.mdt_debug_beta(-1.0, 2.0, 0.5, 1, ctypes.byref(out)) == 1  # MDT_EARG


def test_describe_finds_what_prepare_finds_without_copying(pkg):
    """reserve(like = array) (round 6): shape, eltype, location and context key of the request the array would make -- the rule of
    _prepare / _ctx_key -- without the dense copy _prepare takes of a host array"""
    api = pkg.api
    for x in (np.zeros((100, 4, 7), dtype=np.float32), np.zeros((50, 3)), np.zeros(40), np.zeros((20, 2, 3, 5)), np.zeros((30, 2, 4), dtype=np.int64)):
        d, pr = api._describe(x), api._prepare(x)
        assert (d.D, d.C, d.P, d.f32, d.dev) == (pr.D, pr.C, pr.P, pr.f32, pr.dev), x.shape
        assert api._ctx_key(d) == api._ctx_key(pr) == (0, 0)   # a host request runs on the context of (device 0, stream 0)
        a = api._args(d, 0, 0, 0, 2, 250, False, 0.1)
        assert (a.draws, a.chains, a.params, a.location, a.chain_stride, a.param_stride) == (pr.D, pr.C, pr.P, pkg._lib.MDT_HOST, 0, 0)
    with pytest.raises(pkg.ArgumentError):
        api._describe(np.float64(1.0))
