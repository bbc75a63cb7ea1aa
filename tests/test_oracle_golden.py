"""CPU: the C oracle and the independent numpy/scipy oracle against the committed golden vectors."""
import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden
from oracle import c_oracle as co
from oracle import numpy_oracle as no


def _keys(g):
    for k in g:
        if k.startswith("ess|"):
            _, kind, method, s, ml = k.split("|")
            yield k, "rhat|" + k[4:], kind, method, int(s), int(ml)


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith(("f32", "nested"))])
def test_c_oracle_matches_golden(name):
    g = load_golden(name)
    for ke, kr, kind, method, s, ml in _keys(g):
        e, r = co.ess_rhat(g["x"], kind=kind, autocov=method, split_chains=s, maxlag=ml)
        assert_close(e, g[ke], 1e-12, f"{name} {ke}")
        assert_close(r, g[kr], 1e-12, f"{name} {kr}")


@pytest.mark.parametrize("name", ["cfg1_randn_1000x4x10", "ar1_500x4x10", "ties_200x4x6", "odd_1001x3x3",
                                  "maxlag_edges_60x4x3", "short_9x3x2", "short_10x3x2"])
def test_numpy_oracle_matches_golden(name):
    g = load_golden(name)
    for ke, kr, kind, method, s, ml in _keys(g):
        if name.startswith("cfg1") and (method == "bda" or s == 1):
            continue  # keep the CPU suite short; bda/split-1 are covered on the other fixtures
        e, r = no.ess_rhat(g["x"], kind=kind, method=method, s=s, maxlag=ml)
        assert_close(e, g[ke], 1e-10, f"{name} {ke}")
        assert_close(r, g[kr], 1e-10, f"{name} {kr}")


def test_stage_outputs_cfg1():
    g = load_golden("cfg1_randn_1000x4x10")
    xp = g["x"][:, :, 0]
    r = co.tiedrank(xp)
    assert np.array_equal(r, g["stage_rank"])  # ranks bit-exact
    assert np.array_equal(np.sort(r), np.arange(1, xp.size + 1))  # continuous data: a permutation
    assert_close(co.rank_normalize(xp), g["stage_z"], 1e-14)
    assert_close(no.rank_normalize(xp), g["stage_z"], 1e-10)
    assert_close(co.fold_around_median(xp), g["stage_fold"], 0)
    assert_close([co.quantile(xp, 0.05), co.quantile(xp, 0.95)], g["stage_q05"], 0)
    for m in ("direct", "fft", "bda"):
        assert_close(co.mean_autocov(xp, 2, m, nlags=251), g["stage_acov_" + m], 1e-12)
    # FFT and direct autocovariances agree (test/ess_rhat.jl:259-266)
    assert_close(g["stage_acov_fft"], g["stage_acov_direct"], 1e-9)
    eb, et, rb, rt = co.bulk_tail(g["x"], autocov="fft")
    assert_close(np.stack([eb, et, rb, rt]), g["bulk_tail_fft"], 1e-12)


def test_ties_ranks_are_half_integers():
    g = load_golden("ties_200x4x6")
    r = co.tiedrank(g["x"][:, :, 0])
    assert np.array_equal(r, g["stage_rank"])
    assert np.all(r * 2 == np.round(r * 2))
    assert r.sum() == r.size * (r.size + 1) / 2


def test_special_values_ranks():
    g = load_golden("special_50x2x3")
    r0 = co.tiedrank(g["x"][:, :, 0])
    assert np.array_equal(r0, g["stage_rank0"])
    x0 = g["x"][:, :, 0].ravel(order="F")
    iz = np.where(x0 == 0)[0]
    assert len(iz) == 2 and r0[iz[0]] == r0[iz[1]]  # -0.0 ties 0.0 (StatsBase tiedrank uses ==)
    r1 = co.tiedrank(g["x"][:, :, 1])
    x1 = g["x"][:, :, 1].ravel(order="F")
    assert r1[np.argmax(x1)] == x1.size and r1[np.argmin(x1)] == 1


def test_nested_and_mcse_golden():
    g = load_golden("nested_200x16x4")
    for kind in ("basic", "bulk", "tail", "rank"):
        for s in (1, 2):
            assert_close(co.rhat_nested(g["x"], g["inds"], kind=kind, split_chains=s), g[f"nested|{kind}|{s}"], 1e-12)
    assert_close(co.mcse_mean(g["x"]), g["mcse_mean"], 1e-12)
    assert_close(no.mcse_mean(g["x"]), g["mcse_mean"], 1e-10)


def test_f32_golden():
    g = load_golden("f32_300x4x6")
    for kind in ("basic", "bulk", "tail", "rank"):
        for m in ("direct", "fft"):
            e, r = co.ess_rhat(g["x"], kind=kind, autocov=m)
            assert e.dtype == np.float32
            assert_close(e, g[f"ess|{kind}|{m}|2|250"], 1e-6)
            assert_close(r, g[f"rhat|{kind}|{m}|2|250"], 1e-6)
            # the Float32-faithful variant stays within Float32 noise of the f64-internal one
            assert_close(g[f"rhatF|{kind}|{m}|2|250"], r, 1e-3)


@pytest.mark.parametrize("name", golden_names(estimators=True))
def test_estimator_golden(name):
    """ess(x; kind = median | std | mad | quantile(p)) and mcse(x; kind = std) (src/ess_rhat.jl:626-659,
    src/mcse.jl:52-65): the C oracle reproduces the committed vectors (minted where C and numpy agreed)"""
    from oracle import c_oracle as co
    g = load_golden(name)
    x = load_golden(name[4:])["x"]
    n = 0
    for k, want in g.items():
        if k.startswith("est|"):
            _, est, p, m, s = k.split("|")
            got = co.ess_estimator(x, est, prob=float(p), autocov=m, split_chains=int(s))
            assert_close(got, want, 1e-12, f"{name} {k}")
            n += 1
        elif k.startswith("err|"):
            _, _, est, p, m, s = k.split("|")
            with pytest.raises(co.OracleError):
                co.ess_estimator(x, est, prob=float(p), autocov=m, split_chains=int(s))
    if "mcse_std" in g:
        assert_close(co.mcse_std(x), g["mcse_std"], 1e-12, name + " mcse_std")
    for k, want in g.items():
        if k.startswith("mcseq|"):
            _, est, p = k.split("|")
            assert_close(co.mcse_quantile(x, est, prob=float(p)), want, 1e-12, f"{name} {k}")
    assert n > 0


@pytest.mark.parametrize("name", golden_names(diag=True))
def test_diag_golden(name):
    """bfmi (src/bfmi.jl:38-45) and gelmandiag (src/gelmandiag.jl:1-60): the C oracle reproduces the committed
    vectors; diag_bfmi_reference holds the reference's own known answers (test/bfmi.jl:2-41)."""
    from oracle import c_oracle as co
    g = load_golden(name)
    if name == "diag_bfmi_reference":
        assert abs(co.bfmi(g["energy_a"]) - float(g["expect_a"])) < 1e-12
        assert abs(co.bfmi(g["energy_b"]) - float(g["expect_b"])) < 1e-9
        assert abs(co.bfmi(g["energy_b"].astype(np.int64)) - float(g["expect_b"])) < 1e-9
        em = np.repeat(g["energy_b"][:, None], 4, axis=1)  # repeat(energy, 1, 4), test/bfmi.jl:37-40
        assert_close(co.bfmi(em), np.full(4, float(g["expect_b"])), 1e-8, name)
        return
    x = load_golden(name[5:])["x"]
    for k, want in g.items():
        if k.startswith("gelman|"):
            _, alpha, which = k.split("|")
            psrf, ci = co.gelmandiag(x, alpha=float(alpha))
            assert_close(psrf if which == "psrf" else ci, want, 1e-12, f"{name} {k}")
    assert_close(co.bfmi(x.reshape(x.shape[0], -1, order="F")), g["bfmi"], 1e-12, name + " bfmi")


def test_gelmandiag_oracle_errors():
    """fewer than two chains: ErrorException (src/gelmandiag.jl:3, test/gelmandiag.jl:29)"""
    from oracle import c_oracle as co
    with pytest.raises(co.OracleError):
        co.gelmandiag(np.random.default_rng(0).standard_normal((100, 1, 3)))
