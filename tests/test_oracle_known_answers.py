"""CPU: pin the oracle on every known-answer / property test the reference's own suite holds
for the hot path.  Each test cites the reference test it restates."""
import numpy as np
import pytest

from conftest import ar1, assert_close
from oracle import c_oracle as co

KINDS = ["rank", "bulk", "tail", "basic"]
METHODS = ["direct", "fft", "bda"]


def test_niter_le_4_gives_nan_ess_but_valid_rhat(rng):
    # test/ess_rhat.jl:66-98
    x = rng.random((4, 3, 5))
    x2 = rng.random((5, 3, 5))
    x4 = rng.random((1, 3, 5))
    for kind in KINDS:
        e, r = co.ess_rhat(x, kind=kind, split_chains=1)
        assert np.all(np.isnan(e))
        _, r_only = co.ess_rhat(x, kind=kind, split_chains=1, want_ess=False)
        assert np.array_equal(r, r_only)
        e, _ = co.ess_rhat(x4, kind=kind, split_chains=2)
        assert np.all(np.isnan(e))
        co.ess_rhat(x2, kind=kind, split_chains=1)  # niter = 5 is allowed
        e, r = co.ess_rhat(x2, kind=kind, split_chains=2)
        assert np.all(np.isnan(e))
        assert np.array_equal(r, co.ess_rhat(x2, kind=kind, split_chains=2, want_ess=False)[1])


def test_maxlag_domain(rng):
    # test/ess_rhat.jl:90-92
    x3 = rng.random((100, 3, 5))
    for kind in KINDS:
        co.ess_rhat(x3, kind=kind, maxlag=1)
        with pytest.raises(co.OracleError) as ei:
            co.ess_rhat(x3, kind=kind, maxlag=0)
        assert ei.value.code == 2


def test_relative(rng):
    # test/ess_rhat.jl:100-113
    for kind in KINDS:
        for niter in (50, 100):
            for nchains in (2, 4):
                x = rng.random((niter, nchains, 2))
                S, R = co.ess_rhat(x, kind=kind)
                S2, R2 = co.ess_rhat(x, kind=kind, relative=True)
                assert_close(S2, S / (niter * nchains), 1e-14)
                assert np.array_equal(R, R2)


def test_batch_consistency_bitwise(rng):
    # test/ess_rhat.jl:167-204: slices give bit-identical results; ess/rhat/ess_rhat agree
    x = rng.standard_normal((300, 4, 6))
    for kind in KINDS:
        for s in (1, 2):
            for m, ml in (("direct", 100), ("bda", 10)):
                S, R = co.ess_rhat(x, kind=kind, split_chains=s, autocov=m, maxlag=ml)
                _, R1 = co.ess_rhat(x, kind=kind, split_chains=s, want_ess=False)
                assert np.array_equal(R, R1)
                for j in range(x.shape[2]):
                    Sj, Rj = co.ess_rhat(x[:, :, j], kind=kind, split_chains=s, autocov=m, maxlag=ml)
                    assert Sj[0] == S[j] and Rj[0] == R[j]


@pytest.mark.parametrize("scale", [1, 100])
@pytest.mark.parametrize("nchains,split", [(1, 1), (1, 2), (10, 2)])
def test_iid(scale, nchains, split):
    # test/ess_rhat.jl:210-240 (D reduced to 2000 to keep the CPU suite short)
    rng = np.random.default_rng(7)
    x = scale * rng.standard_normal((2000, nchains, 12))
    ntotal = x.shape[0] * x.shape[1]
    es, rs = co.ess_rhat(x, split_chains=split)
    ef, rf = co.ess_rhat(x, split_chains=split, autocov="fft")
    eb, rb = co.ess_rhat(x, split_chains=split, autocov="bda")
    assert_close(ef, es, 1.5e-8)  # ≈ with rtol sqrt(eps)
    assert np.array_equal(rs, rf) and np.array_equal(rs, rb)
    assert np.all(np.abs(es / ntotal - 1) < 0.15)
    assert np.all(np.abs(rs - 1) < 0.1)


def test_identical_samples_nan():
    # test/ess_rhat.jl:242-257
    x = np.ones((1000, 10, 4))
    for m in METHODS:
        e, r = co.ess_rhat(x, autocov=m)
        assert np.all(np.isnan(e)) and np.all(np.isnan(r))


def test_autocov_matches_statsbase_definition(rng):
    # test/ess_rhat.jl:14-26, 259-266: mean over chains of the demeaned biased autocovariance
    x = rng.standard_normal((1000, 10, 1))
    y = co.copyto_split(x[:, :, 0], 2)
    yc = y - y.mean(axis=0)
    n = y.shape[0]
    expect = np.array([np.mean([(yc[: n - k, j] * yc[k:, j]).sum() / n for j in range(y.shape[1])]) for k in range(20)])
    for m in ("direct", "fft"):
        assert_close(co.mean_autocov(x[:, :, 0], 2, m, nlags=20), expect, 1e-9)


def test_unmixed_epochs(rng):
    # test/ess_rhat.jl:268-276
    x = rng.standard_normal((1000, 4, 10)) + np.repeat([0.0, 10.0], 500)[:, None, None]
    e1, r1 = co.ess_rhat(x, kind="basic", split_chains=1)
    e2, r2 = co.ess_rhat(x, kind="basic", split_chains=2)
    assert np.all(np.abs(r1 - 1) < 0.1)
    assert np.all(e2 < e1) and np.all(r2 > 2)


@pytest.mark.parametrize("ndraws", [10, 100])
@pytest.mark.parametrize("phi", [-0.3, -0.9])
def test_antithetic_threshold_exact(ndraws, phi):
    # test/ess_rhat.jl:314-327: max ESS == ntotal*log10(ntotal) EXACTLY, min ESS > 0
    rng = np.random.default_rng(3)
    x = ar1(phi, ndraws, 4, 1000, rng)
    S, _ = co.ess_rhat(x, kind="basic")
    ntotal = ndraws * 4
    import math
    assert S.max() == ntotal * math.log10(ntotal)  # same libm log10 on both sides, as Julia uses its own on both
    assert S.min() > 0


def test_bulk_is_basic_of_rank_normalized_and_monotone_invariant(rng):
    # test/ess_rhat.jl:329-335
    from scipy.stats import cauchy, norm
    x = rng.standard_normal((1000, 4, 10))
    z = np.stack([co.rank_normalize(x[:, :, p]) for p in range(10)], axis=2)
    assert np.array_equal(co.ess_rhat(x, kind="bulk")[0], co.ess_rhat(z, kind="basic")[0])
    xc = cauchy.ppf(norm.cdf(x))
    assert np.array_equal(co.ess_rhat(x, kind="bulk")[0], co.ess_rhat(xc, kind="bulk")[0])


def test_tail_detects_mismatched_scales():
    # test/ess_rhat.jl:337-364
    rng = np.random.default_rng(11)
    phi = 0.1
    sig = np.sqrt(1 - phi ** 2) * np.array([0.1, 1, 10, 100])
    x = 10 + np.concatenate([ar1(phi, 1000, 1, 100, rng, sigma=s) for s in sig], axis=1)
    cut = 100 * 4
    S, R = co.ess_rhat(x, kind="basic")
    assert np.all(S >= cut) and np.all(R <= 1.01)
    S, R = co.ess_rhat(x, kind="bulk")
    assert np.all(S >= cut) and np.all(R <= 1.01)
    S, R = co.ess_rhat(x, kind="tail")
    assert np.all(S < cut) and np.all(R > 1.01)


def test_integer_input_tail_finite(rng):
    # test/ess_rhat.jl:367-375
    x = rng.integers(1, 10001, size=(1000, 4, 20))
    S, _ = co.ess_rhat(x, kind="tail")
    assert S.dtype == np.float64 and np.all(np.isfinite(S))


def test_copyto_split_rows():
    # test/utils.jl:26-56: 51 rows split 2 keeps rows 1:25, 27:51; split 3 on 11 rows drops 4 and 8
    x = np.arange(1, 52, dtype=np.float64)[:, None]
    out = co.copyto_split(x, 2)
    assert np.array_equal(out[:, 0], np.arange(1, 26)) and np.array_equal(out[:, 1], np.arange(27, 52))
    x = np.arange(1, 12, dtype=np.float64)[:, None]
    out = co.copyto_split(x, 3)  # niter 3, rem 2: pieces 1:3, 5:7, 9:11
    assert np.array_equal(out.T.ravel(), [1, 2, 3, 5, 6, 7, 9, 10, 11])


def test_rank_normalize_moments(rng):
    # test/utils.jl:98-107
    for shape in ((1000,), (1000, 4)):
        z = co.rank_normalize(rng.standard_normal(shape) * 7 + 3)
        assert abs(z.mean()) < 1e-13
        assert abs(z.std(ddof=1) - 1) < 1e-2


def test_fold_around_median(rng):
    # test/utils.jl:109-123
    x = rng.standard_normal((101, 4))
    assert_close(co.fold_around_median(x), np.abs(x - np.median(x)), 1e-15)


def test_rhat_nested_properties(rng):
    # test/rhat_nested.jl:113-155
    D, C, P = 200, 8, 5
    x = rng.standard_normal((D, C, P))
    inds = np.arange(C, dtype=np.int32).reshape(2, 4, order="F")
    for kind in KINDS:
        r = co.rhat_nested(x, inds, kind=kind)
        perm = rng.permutation(C)
        inv = np.empty(C, dtype=np.int64)
        inv[perm] = np.arange(C)
        r2 = co.rhat_nested(x[:, perm, :], inv[inds].astype(np.int32), kind=kind)  # permutation invariance (==)
        assert_close(r2, r, 1e-13)
    rb = co.rhat_nested(x, inds, kind="bulk")
    rt = co.rhat_nested(x, inds, kind="tail")
    assert np.array_equal(co.rhat_nested(x, inds, kind="rank"), np.maximum(rb, rt))
    # one chain per superchain, no split: Rn ≈ sqrt(Rs^2 + 1/ndraws)  (:132-146)
    inds1 = np.arange(C, dtype=np.int32).reshape(1, C)
    rn = co.rhat_nested(x, inds1, kind="basic", split_chains=1)
    _, rs = co.ess_rhat(x, kind="basic", split_chains=1)
    assert_close(rn, np.sqrt(rs ** 2 + 1 / D), 1e-12)


def test_mcse_mean_properties(rng):
    # test/mcse.jl:86-101 constant -> NaN; definition std/sqrt(ess) (src/mcse.jl:45-51)
    assert np.all(np.isnan(co.mcse_mean(np.ones((100, 4, 3)))))
    x = rng.standard_normal((400, 4, 3))
    S, _ = co.ess_rhat(x, kind="basic")
    expect = np.array([x[:, :, p].std(ddof=1) for p in range(3)]) / np.sqrt(S)
    assert_close(co.mcse_mean(x), expect, 1e-13)
