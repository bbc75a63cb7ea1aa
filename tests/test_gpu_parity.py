"""GPU: the HIP path, called through the C-ABI (libmdt_hip.so via ctypes), against the committed
golden vectors and against the oracle on seeded inputs.  Bar: ranks bit-exact; Float64 results
within 1e-6 relative (north_star) -- we assert 1e-9; Float32 within 1e-6 of the f64-internal
comparator."""
import math

import numpy as np
import pytest

from conftest import ar1, assert_close, golden_names, load_golden

pytestmark = pytest.mark.gpu

RTOL64 = 1e-9
FFT_ON_DEVICE = True
KINDS = ["rank", "bulk", "tail", "basic"]


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="module")
def methods(pkg):
    return {"direct": pkg.AutocovMethod(), "fft": pkg.FFTAutocovMethod(), "bda": pkg.BDAAutocovMethod()}


def dev(x):
    """(D, C, P) numpy -> cuda tensor with draws-fastest memory (the zero-copy layout)"""
    import torch
    x = np.asarray(x)
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 1, 0)))).cuda()
    return t.permute(2, 1, 0)


def host(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)


def _keys(g):
    for k in g:
        if k.startswith("ess|"):
            _, kind, method, s, ml = k.split("|")
            yield k, "rhat|" + k[4:], kind, method, int(s), int(ml)


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith(("f32", "nested"))])
def test_golden_f64(pkg, methods, name):
    import warnings
    g = load_golden(name)
    xd = dev(g["x"])
    for ke, kr, kind, method, s, ml in _keys(g):
        if method == "fft" and not FFT_ON_DEVICE:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            S, R = pkg.ess_rhat(xd, kind=kind, autocov_method=methods[method], split_chains=s, maxlag=ml)
        assert_close(host(S), g[ke], RTOL64, f"{name} {ke}")
        assert_close(host(R), g[kr], RTOL64, f"{name} {kr}")


def test_stage_ranks_bit_exact(pkg):
    from oracle import c_oracle as co
    for name, key in (("cfg1_randn_1000x4x10", "stage_rank"), ("ties_200x4x6", "stage_rank"),
                      ("special_50x2x3", "stage_rank0"), ("special_50x2x3", "stage_rank1")):
        g = load_golden(name)
        p = 1 if key.endswith("1") else 0
        ranks, z = pkg.rank_normalize(dev(g["x"][:, :, p:p + 1]))
        r = host(ranks)[:, :, 0].ravel(order="F")
        assert np.array_equal(r, g[key]), name  # bit-exact rank indices
        zz = host(z)[:, :, 0]
        assert_close(zz, co.rank_normalize(g["x"][:, :, p]), 1e-12, name)
    g = load_golden("cfg1_randn_1000x4x10")
    assert_close(host(pkg.rank_normalize(dev(g["x"][:, :, :1]))[1])[:, :, 0], g["stage_z"], 1e-12)
    # folded ranks
    ranks, _ = pkg.rank_normalize(dev(g["x"]), fold=True)
    for p in range(g["x"].shape[2]):
        want = co.tiedrank(co.fold_around_median(g["x"][:, :, p]))
        assert np.array_equal(host(ranks)[:, :, p].ravel(order="F"), want)


def test_bulk_tail_fused_matches_separate_and_oracle(pkg):
    from oracle import c_oracle as co
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal((2000, 4, 6)), ar1(0.9, 2000, 4, 3, rng), ar1(-0.5, 2000, 4, 3, rng),
                        ar1(0.99, 2000, 4, 2, rng)], axis=2)
    xd = dev(x)
    eb, et, rb, rt = [host(o) for o in pkg.bulk_tail(xd)]
    web, wet, wrb, wrt = co.bulk_tail(x)
    for a, b, nm in ((eb, web, "ess_bulk"), (et, wet, "ess_tail"), (rb, wrb, "rhat_bulk"), (rt, wrt, "rhat_tail")):
        assert_close(a, b, RTOL64, nm)
    Sb, Rb = pkg.ess_rhat(xd, kind="bulk")
    St, Rt = pkg.ess_rhat(xd, kind="tail")
    Sr, Rr = pkg.ess_rhat(xd, kind="rank")
    assert np.array_equal(host(Sb), eb) and np.array_equal(host(Rb), rb)  # batch/fusion consistency, bitwise
    assert np.array_equal(host(St), et) and np.array_equal(host(Rt), rt)
    assert np.array_equal(host(Sr), eb) and np.array_equal(host(Rr), np.maximum(rb, rt))
    assert np.array_equal(host(pkg.ess(xd, kind="bulk")), eb)
    assert np.array_equal(host(pkg.rhat(xd, kind="tail")), rt)


def test_config2_shape_subset(pkg):
    """10_000 draws x 4 chains (BASELINE config 2 per-parameter shape), 24 params vs the oracle"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.standard_normal((10000, 4, 12))] + [ar1(phi, 10000, 4, 3, rng) for phi in (-0.3, 0.3, 0.7, 0.9)], axis=2)
    x[:, 1, 12:] += 0.1
    got = [host(o) for o in pkg.bulk_tail(dev(x))]
    want = co.bulk_tail(x)
    for a, b in zip(got, want):
        assert_close(a, b, RTOL64)
    # position independence: same parameter anywhere in the batch gives identical bits
    perm = rng.permutation(x.shape[2])
    got2 = [host(o) for o in pkg.bulk_tail(dev(x[:, :, perm]))]
    for a, b in zip(got, got2):
        assert np.array_equal(a[perm], b)


def test_nested_and_mcse(pkg):
    g = load_golden("nested_200x16x4")
    xd = dev(g["x"])
    ids = np.repeat(np.arange(1, 5), 4)
    for kind in KINDS:
        for s in (1, 2):
            r = pkg.rhat_nested(xd, ids, kind=kind, split_chains=s)
            assert_close(host(r), g[f"nested|{kind}|{s}"], RTOL64, f"nested {kind} {s}")
    assert_close(host(pkg.mcse(xd)), g["mcse_mean"], RTOL64)
    # label-type and permutation invariance (test/rhat_nested.jl:102-130)
    r1 = host(pkg.rhat_nested(xd, ids, kind="rank"))
    r2 = host(pkg.rhat_nested(xd, [chr(96 + i) for i in ids], kind="rank"))
    assert np.array_equal(r1, r2)


def test_f32(pkg, methods):
    g = load_golden("f32_300x4x6")
    xd = dev(g["x"])
    assert xd.dtype.__str__() == "torch.float32"
    for kind in KINDS:
        S, R = pkg.ess_rhat(xd, kind=kind)
        assert host(S).dtype == np.float32
        assert_close(host(S), g[f"ess|{kind}|direct|2|250"], 1e-6, kind)
        assert_close(host(R), g[f"rhat|{kind}|direct|2|250"], 1e-6, kind)


def test_known_answers_on_device(pkg):
    import warnings
    rng = np.random.default_rng(3)
    # antithetic: max ESS == ntotal*log10(ntotal) exactly (test/ess_rhat.jl:314-327)
    for ndraws in (10, 100):
        x = ar1(-0.9, ndraws, 4, 500, rng)
        S = host(pkg.ess(dev(x), kind=np.mean))
        ntotal = ndraws * 4
        assert S.max() == ntotal * math.log10(ntotal) and S.min() > 0
    # constant -> NaN (test/ess_rhat.jl:242-257)
    S, R = pkg.ess_rhat(dev(np.ones((1000, 10, 4))))
    assert np.all(np.isnan(host(S))) and np.all(np.isnan(host(R)))
    # niter <= 4 -> NaN ESS, R-hat == rhat(...) (test/ess_rhat.jl:66-98); maxlag=0 -> DomainError
    x = rng.random((5, 3, 5))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        S, R = pkg.ess_rhat(dev(x), kind="bulk", split_chains=2)
        assert any("ESS cannot be computed" in str(i.message) for i in w)
    assert np.all(np.isnan(host(S)))
    assert np.array_equal(host(R), host(pkg.rhat(dev(x), kind="bulk", split_chains=2)))
    with pytest.raises(pkg.DomainError):
        pkg.ess_rhat(dev(rng.random((100, 3, 5))), maxlag=0)
    pkg.ess_rhat(dev(rng.random((100, 3, 5))), maxlag=1)
    # relative (test/ess_rhat.jl:100-113)
    x = rng.random((100, 4, 2))
    S, R = pkg.ess_rhat(dev(x), kind="tail")
    S2, R2 = pkg.ess_rhat(dev(x), kind="tail", relative=True)
    assert_close(host(S2), host(S) / 400, 1e-14)
    assert np.array_equal(host(R), host(R2))
    # scalar outputs for 2-d input; host (numpy) input path gives the same bits as device input
    x = rng.standard_normal((500, 4, 3))
    S = pkg.ess(dev(x)[:, :, 0].contiguous(), kind="bulk")
    assert isinstance(S, float)
    Sh, Rh = pkg.ess_rhat(x, kind="rank")
    Sd, Rd = pkg.ess_rhat(dev(x), kind="rank")
    assert isinstance(Sh, np.ndarray) and np.array_equal(Sh, host(Sd)) and np.array_equal(Rh, host(Rd))
    assert S == Sh[0]


def test_nan_semantics(pkg):
    """NaN ranks last in index order (isless / stable sortperm); median -> NaN; quantile -> ArgumentError"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(8)
    x = rng.standard_normal((300, 4, 3))
    x[5, 1, 1] = np.nan
    x[200, 3, 1] = np.nan
    ranks, _ = pkg.rank_normalize(dev(x))
    for p in range(3):
        assert np.array_equal(host(ranks)[:, :, p].ravel(order="F"), co.tiedrank(x[:, :, p]))
    R = host(pkg.rhat(dev(x), kind="rank"))
    _, want = co.ess_rhat(x, kind="rank", want_ess=False)
    assert_close(R, want, RTOL64)
    with pytest.raises(pkg.ArgumentError):
        pkg.ess(dev(x), kind="tail")


def test_methods_agree_and_rhat_is_method_independent(pkg, methods):
    """test/ess_rhat.jl:210-240: default == AutocovMethod; FFT ≈ direct (rtol sqrt(eps)); R-hat
    identical (==) across all three methods and equal to rhat(x) for every kind"""
    rng = np.random.default_rng(21)
    for scale, nchains, split in ((1, 1, 1), (50, 10, 2), (100, 4, 2)):
        x = scale * rng.standard_normal((2000, nchains, 8))
        xd = dev(x)
        for kind in KINDS:
            S0, R0 = pkg.ess_rhat(xd, kind=kind, split_chains=split)
            Sd, Rd = pkg.ess_rhat(xd, kind=kind, split_chains=split, autocov_method=methods["direct"])
            Sf, Rf = pkg.ess_rhat(xd, kind=kind, split_chains=split, autocov_method=methods["fft"])
            Sb, Rb = pkg.ess_rhat(xd, kind=kind, split_chains=split, autocov_method=methods["bda"])
            assert np.array_equal(host(S0), host(Sd))
            assert_close(host(Sf), host(Sd), 1.5e-8, f"fft vs direct {kind}")
            Ronly = host(pkg.rhat(xd, kind=kind, split_chains=split))
            for R in (R0, Rd, Rf, Rb):
                assert np.array_equal(host(R), Ronly), kind


def test_long_chain_shape(pkg, methods):
    """BASELINE config 3 per-parameter shape (100_000 draws x 8 chains), 3 params: the multi-round
    ranking (n = 800_000 > LDS chunk) and the tiled series path against the oracle"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(33)
    x = np.concatenate([rng.standard_normal((100_000, 8, 2)), ar1(0.9, 100_000, 8, 1, rng)], axis=2)
    got = [host(o) for o in pkg.bulk_tail(dev(x))]
    want = co.bulk_tail(x)
    for a, b, nm in zip(got, want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
        assert_close(a, b, RTOL64, nm)
    # FFTAutocovMethod beyond the in-LDS transform (8192 padded points): the same linear autocorrelation from
    # lagged products; BASELINE config 3 forces this method
    for kind in ("bulk", "tail", "basic"):
        S, R = pkg.ess_rhat(dev(x), kind=kind, autocov_method=methods["fft"])
        We, Wr = co.ess_rhat(x, kind=kind, autocov="fft")
        assert_close(host(S), We, RTOL64, f"fft long {kind} ess")
        assert_close(host(R), Wr, RTOL64, f"fft long {kind} rhat")
    xc = x.copy()
    xc[:50_000, 3, 0] = 0.25  # a constant split chain: 0/0 = NaN in the FFT method only (src/ess_rhat.jl:192)
    S = pkg.ess(dev(xc), kind="basic", autocov_method=methods["fft"])
    We, _ = co.ess_rhat(xc, kind="basic", autocov="fft")
    assert np.isnan(host(S)[0]) and np.isnan(We[0])
    assert_close(host(S), We, RTOL64, "fft long const chain")


def test_long_slices_with_point_masses_and_nans(pkg):
    """n = 80_000 per parameter (the two-level rank phase): a constant column and a 60 % point mass force
    its fallback, a few heavy ties and NaNs exercise the reserved bins / NaN tail; ranks bit-exact"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(77)
    D, C = 20_000, 4
    x = rng.standard_normal((D, C, 5))
    x[:, :, 0] = 1.5
    m = rng.random((D, C)) < 0.6
    x[:, :, 1][m] = 0.0
    x[:, :, 2] = rng.integers(0, 50, size=(D, C))
    x[::1000, 1, 3] = np.nan
    x[:, :, 4] = np.round(x[:, :, 4], 2)
    xd = dev(x)
    r, z = pkg.rank_normalize(xd)
    for p in range(5):
        assert np.array_equal(host(r)[:, :, p], co.tiedrank(x[:, :, p].ravel(order="F")).reshape(D, C, order="F"), equal_nan=True), p
    xq = x[:, :, [0, 1, 2, 4]]  # (the reference throws on NaN in the tail quantile, and so do we)
    got = [host(o) for o in pkg.bulk_tail(dev(xq))]
    want = co.bulk_tail(xq)
    for a, b, nm in zip(got, want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
        assert_close(a, b, RTOL64, nm)
    with pytest.raises(pkg.ArgumentError):
        pkg.bulk_tail(xd)


def test_f32_other_methods_and_sizes(pkg, methods):
    from oracle import c_oracle as co
    rng = np.random.default_rng(44)
    x = ar1(0.5, 2500, 16, 5, rng).astype(np.float32)  # n = 40_000 float32
    xd = dev(x)
    for kind in ("bulk", "tail"):
        for m in ("direct", "fft", "bda"):
            S, R = pkg.ess_rhat(xd, kind=kind, autocov_method=methods[m])
            We, Wr = co.ess_rhat(x, kind=kind, autocov=m)
            assert_close(host(S), We, 1e-6, f"{kind} {m}")
            assert_close(host(R), Wr, 1e-6, f"{kind} {m}")
    ids = np.repeat(np.arange(4), 4)
    inds = np.arange(16, dtype=np.int32).reshape(4, 4, order="F")
    assert_close(host(pkg.rhat_nested(xd, ids)), co.rhat_nested(x, inds), 1e-6)
    assert_close(host(pkg.mcse(xd)), co.mcse_mean(x), 1e-6)


def test_config5_shape_f32_nested_and_mcse(pkg):
    """BASELINE config 5 per-parameter shape: 10_000 draws x 16 chains (4 superchains of 4), Float32:
    n = 160_000 exceeds the 16-bit slot range and the pooled indicator stream, so the wide paths run"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(55)
    base = rng.standard_normal((1, 4, 3)) * 0.2
    x = (ar1(0.6, 10_000, 16, 3, rng) + np.repeat(base, 4, axis=1)).astype(np.float32)
    xd = dev(x)
    ids = np.repeat(np.arange(1, 5), 4)
    inds = np.arange(16, dtype=np.int32).reshape(4, 4, order="F")
    assert_close(host(pkg.rhat_nested(xd, ids, kind="rank")), co.rhat_nested(x, inds, kind="rank"), 1e-6)
    assert_close(host(pkg.mcse(xd)), co.mcse_mean(x), 1e-6)
    S, R = pkg.ess_rhat(xd, kind="tail")
    We, Wr = co.ess_rhat(x, kind="tail")
    assert_close(host(S), We, 1e-6)
    assert_close(host(R), Wr, 1e-6)


def test_both_rank_variants_agree_bitwise(pkg, monkeypatch):
    """the two rank-phase formulations (rounds re-stream the slice / partitioned rounds) give identical bits"""
    g = load_golden("cfg1_randn_1000x4x10")
    rng = np.random.default_rng(66)
    x = np.concatenate([g["x"], rng.integers(1, 11, size=(1000, 4, 3)).astype(np.float64)], axis=2)
    xd = dev(x)
    outs = {}
    # (the one-chunk path of round 4 serves this shape under variant 2 and sums its lag products in another order: it has
    # its own test, test_one_chunk_path_matches_the_general_path; here the three general formulations are compared)
    monkeypatch.setenv("MDT_ONECHUNK", "0")
    monkeypatch.setenv("MDT_WAVE", "0")                # (round 5: the wave-per-parameter kernel takes this shape under variant 2 and has its own test)
    for v in ("0", "1", "2"):
        monkeypatch.setenv("MDT_RANK_VARIANT", v)
        outs[v] = [host(o) for o in pkg.bulk_tail(xd)] + [host(pkg.rank_normalize(xd, fold=True)[0])]
    for v in ("1", "2"):
        for k, (a, b) in enumerate(zip(outs["0"], outs[v])):
            if k == 3 and v == "2":
                # R-hat of the folded scores: the two-level variant sums the split-chain moments inside its
                # ranking rounds (shift 0) instead of sweeping stored scores (shift = first draw): same
                # fixed-point unit, not the same bits
                assert_close(a, b, 1e-12, "rhat_tail")
            else:
                assert np.array_equal(a, b, equal_nan=True), (v, k)


@pytest.mark.parametrize("wave_forced", [False, True], ids=["rule", "wave"])
@pytest.mark.parametrize("name", golden_names(estimators=True))
def test_estimator_kinds(pkg, methods, name, monkeypatch, wave_forced):
    """ess(x; kind = estimator) for the estimators with an expectand proxy and mcse(kind = std) against the
    golden vectors; wave_forced: the same through the wave-per-parameter kernel wherever it can run
    (round 6: ess(kind = median | quantile(p)) on short slices -- the order-statistics-only ranking and the indicator phase of
    ess(kind = tail); everything else takes the general kernels as always)"""
    import warnings
    if wave_forced:
        monkeypatch.setenv("MDT_WAVE", "2")
    g = load_golden(name)
    x = load_golden(name[4:])["x"]
    xd = dev(x)
    tol = RTOL64 if x.dtype == np.float64 else 1e-6
    kinds = {"median": np.median, "std": np.std, "mad": "mad"}
    for k, want in g.items():
        if not (k.startswith("est|") or k.startswith("err|")):
            continue
        parts = k.split("|")
        est, p, m, s = parts[-4:]
        kind = pkg.Quantile(float(p)) if est == "quantile" else kinds[est]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if k.startswith("err|"):
                with pytest.raises(pkg.ArgumentError):
                    pkg.ess(xd, kind=kind, autocov_method=methods[m], split_chains=int(s))
            else:
                got = pkg.ess(xd, kind=kind, autocov_method=methods[m], split_chains=int(s))
                assert_close(host(got), want, tol, f"{name} {k}")
    if "mcse_std" in g:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert_close(host(pkg.mcse(xd, kind=np.std)), g["mcse_std"], tol, name + " mcse_std")
            for k, want in g.items():
                if k.startswith("mcseq|"):  # mcse(kind = median | quantile(p)), src/mcse.jl:66-118
                    _, est, p = k.split("|")
                    kind = np.median if est == "median" else pkg.Quantile(float(p))
                    assert_close(host(pkg.mcse(xd, kind=kind)), want, tol, f"{name} {k}")


def test_estimators_from_host_memory(pkg):
    """numpy input (MDT_HOST staging) through the estimator entry points, incl. the two-pass mdt_mcse_quantile"""
    import warnings
    g = load_golden("est_ar1_500x4x10")
    x = load_golden("ar1_500x4x10")["x"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert_close(pkg.ess(x, kind=np.median), g["est|median|0.5|direct|2"], RTOL64, "median host")
        assert_close(pkg.ess(x, kind=pkg.Quantile(0.9)), g["est|quantile|0.9|direct|2"], RTOL64, "q host")
        assert_close(pkg.mcse(x, kind=np.std), g["mcse_std"], RTOL64, "mcse std host")
        assert_close(pkg.mcse(x, kind=np.median), g["mcseq|median|0.5"], RTOL64, "mcse median host")
        assert_close(pkg.mcse(x, kind=pkg.Quantile(0.25)), g["mcseq|quantile|0.25"], RTOL64, "mcse q host")


def test_host_buffers_in_blocks(pkg, monkeypatch):
    """a host buffer of some size goes to the device in blocks of the params axis, copy of block i + 1 under the launch of
    block i (run_host_blocks); forced here on small arrays: every entry point returns the bits of the device-resident
    request, the status flags of any block are reported, uneven blocks and Float32 included"""
    import warnings
    rng = np.random.default_rng(77)
    D, C, P = 300, 4, 37
    x = ar1(0.5, D, C, P, rng) + 0.1 * rng.standard_normal((1, C, P))
    ids = np.array([1, 1, 2, 2])
    monkeypatch.setenv("MDT_HOST_PIPE_MIN", "1")
    monkeypatch.setenv("MDT_HOST_PIPE_BLOCKS", "3")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for xx in (x, x.astype(np.float32)):
            xd = dev(xx)
            for a, b in zip(pkg.bulk_tail(xx), pkg.bulk_tail(xd)):
                assert np.array_equal(host(a), host(b))
            for kind in KINDS:
                (e1, r1), (e2, r2) = pkg.ess_rhat(xx, kind=kind), pkg.ess_rhat(xd, kind=kind)
                assert np.array_equal(host(e1), host(e2)) and np.array_equal(host(r1), host(r2)), kind
            assert np.array_equal(host(pkg.rhat_nested(xx, ids)), host(pkg.rhat_nested(xd, ids)))
            assert np.array_equal(host(pkg.mcse(xx)), host(pkg.mcse(xd)))
            m1, n1 = pkg.mcse_rhat_nested(xx, ids)
            m2, n2 = pkg.mcse_rhat_nested(xd, ids)
            assert np.array_equal(host(m1), host(m2)) and np.array_equal(host(n1), host(n2))
            for kind in (np.median, np.std, "mad", pkg.Quantile(0.3)):
                assert np.array_equal(host(pkg.ess(xx, kind=kind)), host(pkg.ess(xd, kind=kind))), kind
            for kind in (np.std, np.median):
                assert np.array_equal(host(pkg.mcse(xx, kind=kind)), host(pkg.mcse(xd, kind=kind))), kind
        # a NaN in the LAST block: Statistics.quantile's ArgumentError is still reported (the flags accumulate over the blocks)
        xn = x.copy()
        xn[5, 1, P - 2] = np.nan
        with pytest.raises(ValueError):
            pkg.ess(xn, kind="tail")
        with pytest.raises(ValueError):
            pkg.ess(dev(xn), kind="tail")
        assert np.array_equal(host(pkg.rhat(xn)), host(pkg.rhat(dev(xn))))   # (a NaN is ranked last: R-hat stays finite)
        assert np.array_equal(host(pkg.ess(x, kind="tail")), host(pkg.ess(dev(x), kind="tail")))  # and nothing sticks to the context
        # the children of a multi-GPU context stage their own blocks the same way
        for a1, b1 in zip(pkg.bulk_tail(x, devices=[0]), pkg.bulk_tail(dev(x))):
            assert np.array_equal(host(a1), host(b1))
        # fewer parameters than two per block: the plain path
        assert np.array_equal(host(pkg.rhat(x[:, :, :4])), host(pkg.rhat(dev(x[:, :, :4]))))


def test_mcmcchains_storage_order(pkg, monkeypatch):
    """(iterations, parameters, chains) input -- MCMCChains storage -- read IN PLACE through chain_stride /
    param_stride (no permuted copy), and through the device-side permutation helper: bit-equal to the dense call"""
    import torch
    rng = np.random.default_rng(9)
    for dt in (np.float64, np.float32):
        for shape in ((300, 4, 7), (5000, 4, 9)):                 # rank variants 0 (small slices) and 2 (two-level)
            x = (ar1(0.6, *shape, rng) + 0.1 * rng.standard_normal(shape)).astype(dt)   # (draws, chains, params)
            # Julia's column-major (iterations, parameters, chains) = C-order (chains, parameters, iterations)
            store = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 2, 0)))).cuda()
            view = store.permute(2, 0, 1)                         # logical (draws, chains, params); strides (1, D*P, D)
            assert view.shape == shape and view.stride() == (1, shape[0] * shape[2], shape[0])
            assert np.array_equal(view.cpu().numpy(), x)
            before = torch.cuda.memory_allocated()
            a = pkg.ess_rhat(view)
            bt = pkg.bulk_tail(view)
            nested = pkg.rhat_nested(view, [1, 1, 2, 2])
            m = pkg.mcse(view)
            b = pkg.ess_rhat(dev(x))
            assert np.array_equal(host(a.ess), host(b.ess)) and np.array_equal(host(a.rhat), host(b.rhat))
            for u, w in zip(bt, pkg.bulk_tail(dev(x))):
                assert np.array_equal(host(u), host(w), equal_nan=True)
            assert np.array_equal(host(nested), host(pkg.rhat_nested(dev(x), [1, 1, 2, 2])))
            assert np.array_equal(host(m), host(pkg.mcse(dev(x))))
            for kind in ("basic", "tail", np.median):
                assert np.array_equal(host(pkg.ess(view, kind=kind)), host(pkg.ess(dev(x), kind=kind)))
            if shape[0] * shape[1] <= 8192:
                # the same through the wave-per-parameter kernel (forced: seven parameters do not fill it), every request kind it serves
                monkeypatch.setenv("MDT_WAVE", "2")
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    for kind in ("basic", "bulk", "tail", np.median, "mad", np.std, pkg.Quantile(0.3)):
                        assert np.array_equal(host(pkg.ess(view, kind=kind)), host(pkg.ess(dev(x), kind=kind))), kind
                    for kind in (np.mean, np.std, np.median, pkg.Quantile(0.7)):
                        assert np.array_equal(host(pkg.mcse(view, kind=kind)), host(pkg.mcse(dev(x), kind=kind))), kind
                    for u, w in zip(pkg.bulk_tail(view), pkg.bulk_tail(dev(x))):
                        assert np.array_equal(host(u), host(w), equal_nan=True)
                monkeypatch.delenv("MDT_WAVE")
            # nothing of the size of x was allocated on the way (outputs only)
            assert torch.cuda.memory_allocated() - before < x.nbytes // 4
            # the out-of-place helper still works
            t = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (0, 2, 1)))).cuda()  # (draws, params, chains) C-order input
            v = pkg.from_mcmcchains_layout(t)
            assert v.shape == shape and np.array_equal(v.cpu().numpy(), x) and v.permute(2, 1, 0).is_contiguous()
            c = pkg.ess_rhat(v)
            assert np.array_equal(host(c.ess), host(b.ess)) and np.array_equal(host(c.rhat), host(b.rhat))


def test_strided_views_of_dense_only_entry_points(pkg):
    """gelmandiag / bfmi read x as dense (draws, chains, params): a permuted or sliced view is copied dense by the host
    mirror (never handed over with strides), the C-ABI itself refuses strides there, and a strided view together with
    `devices=` takes the dense path too.  Round-2 advisor findings."""
    import ctypes
    import torch
    rng = np.random.default_rng(19)
    shape = (400, 4, 6)
    x = (ar1(0.5, *shape, rng) + 0.1 * rng.standard_normal(shape))
    store = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 2, 0)))).cuda()   # MCMCChains storage
    view = store.permute(2, 0, 1)                                                       # strides (1, D*P, D)
    assert view.stride() == (1, shape[0] * shape[2], shape[0])
    g_view, g_dense = pkg.gelmandiag(view), pkg.gelmandiag(dev(x))
    assert np.array_equal(host(g_view.psrf), host(g_dense.psrf)) and np.array_equal(host(g_view.psrfci), host(g_dense.psrfci))
    from oracle import c_oracle
    want = c_oracle.gelmandiag(x)
    assert_close(host(g_view.psrf), want[0], 1e-9)
    e = torch.from_numpy(rng.standard_normal((500, 8))).cuda()                          # (draws, chains), C-order: draws strided
    sl = e[:, ::2]
    assert np.array_equal(host(pkg.bfmi(sl)), host(pkg.bfmi(sl.contiguous())))
    assert_close(host(pkg.bfmi(sl)), c_oracle.bfmi(sl.cpu().numpy()), 1e-9)
    # devices= with a strided view: dense copy, same bits as the single context
    r1 = pkg.rhat(view, devices=[0])
    assert np.array_equal(host(r1), host(pkg.rhat(dev(x))))
    # the C-ABI refuses strides where the kernels would ignore them
    lib = pkg._lib.load()
    xd = dev(x).permute(2, 1, 0).contiguous()
    out = torch.empty(shape[2], dtype=torch.float64, device="cuda")
    out2 = torch.empty_like(out)
    a = pkg._lib.MdtArgs(shape[0], shape[1], shape[2], pkg._lib.MDT_F64, pkg._lib.MDT_DEVICE, 0, 0, 0, 1, 0, 1, 250, 0.1, None,
                         shape[0] * shape[2], shape[0])
    with pkg._lib.using(0, 0) as h:
        assert lib.mdt_gelmandiag(h, ctypes.byref(a), xd.data_ptr(), 0.05, out.data_ptr(), out2.data_ptr()) == pkg._lib.MDT_EARG
        assert lib.mdt_bfmi(h, ctypes.byref(a), xd.data_ptr(), out.data_ptr()) == pkg._lib.MDT_EARG
        assert lib.mdt_layout_from_dpc(h, ctypes.byref(a), xd.data_ptr(), out.data_ptr()) == pkg._lib.MDT_EARG


def test_mcse_quantile_strided_host_request(pkg):
    """mdt_mcse_quantile with location = MDT_HOST and strides stages the whole strided extent (it staged the dense
    size once: an out-of-bounds device read).  Through the raw C-ABI, against the dense host call."""
    import ctypes
    rng = np.random.default_rng(23)
    D, C, P = 600, 3, 5
    x = ar1(0.4, D, C, P, rng)
    store = np.ascontiguousarray(np.transpose(x, (1, 2, 0)))       # (chains, params, draws): cs = D*P, ps = D
    dense = np.ascontiguousarray(np.transpose(x, (2, 1, 0)))       # (params, chains, draws)
    lib = pkg._lib.load()
    o1, o2 = np.empty(P), np.empty(P)
    L = pkg._lib
    a_s = L.MdtArgs(D, C, P, L.MDT_F64, L.MDT_HOST, L.KIND["median"], 1, 0, 2, 0, 1, 250, 0.5, None, D * P, D)
    a_d = L.MdtArgs(D, C, P, L.MDT_F64, L.MDT_HOST, L.KIND["median"], 1, 0, 2, 0, 1, 250, 0.5, None, 0, 0)
    with L.using(0, 0) as h:
        assert lib.mdt_mcse_quantile(h, ctypes.byref(a_s), store.ctypes.data, o1.ctypes.data) == 0, lib.mdt_last_error(h)
        assert lib.mdt_mcse_quantile(h, ctypes.byref(a_d), dense.ctypes.data, o2.ctypes.data) == 0, lib.mdt_last_error(h)
    assert np.array_equal(o1, o2) and np.all(np.isfinite(o1))


def test_context_registry_pins_and_never_drops_flags(pkg):
    """a context in use cannot be evicted, one with unread sticky flags is not evicted either, and a handle that was
    destroyed is refused instead of being given a fresh lock (round-2 advisor finding on _lib.ctx / lock)"""
    import torch
    L = pkg._lib
    x = dev(np.random.default_rng(5).standard_normal((200, 2, 3)))
    streams = [torch.cuda.Stream() for _ in range(L.MAX_CTX_PER_DEVICE + 2)]
    with L.using(0, streams[0].cuda_stream) as h0:
        for st in streams[1:]:
            with torch.cuda.stream(st):
                pkg.rhat(x)
        assert any(e.h.value == h0.value for e in L._ctxs.values())   # held: survived the evictions
    gone = None
    for k, e in list(L._ctxs.items()):
        if k[0] == 0 and k[1] != 0:
            gone = e.h
    for st in [torch.cuda.Stream() for _ in range(L.MAX_CTX_PER_DEVICE)]:
        with torch.cuda.stream(st):
            pkg.rhat(x)
    if gone is not None and not any(e.h.value == gone.value for e in L._ctxs.values()):
        with pytest.raises(L.MdtRuntimeError):
            L.lock(gone)
    torch.cuda.synchronize()


def test_tail_only_request_takes_the_merge_and_changes_no_bit(pkg, monkeypatch):
    """Round 6: a tail-only request that needs R-hat -- ess_rhat(kind = tail), rhat(kind = tail) -- ranks every round of x for the
    sorted copy (no score is looked up or written) and folds by the merge of fold_merge_phase, as the fused request does, instead of
    an order-statistics-only ranking followed by a second FULL ranking of |x - median| (src/utils.jl:148-158, src/ess_rhat.jl:301-311,
    335-420).  MDT_TAIL_SRT=0 is the former route: the same bits on continuous data, ties, an odd number of draws (rows dropped
    by the split), Float32, 16 chains, an infinity (which sends the fold down the general ranking by itself), in the
    workgroup-per-parameter geometries; and the results are those of the fused request and of the oracle."""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(61)
    monkeypatch.setenv("MDT_WAVE", "0")  # (the wave-per-parameter kernel has its own path for these shapes and its own tests)
    cases = []
    x = ar1(0.5, 4000, 4, 12, rng)
    x[:, :, 3] = np.round(x[:, :, 3], 1)           # heavy ties
    x[9, 2, 5] = np.inf                            # (a NaN under a quantile is an ArgumentError, as in the reference: not here)
    cases.append(("f64 4 chains", x))
    cases.append(("config-2 slice", ar1(0.3, 10000, 4, 3, rng)))
    cases.append(("odd draws", ar1(0.2, 3001, 4, 5, rng)))
    cases.append(("f32 16 chains", ar1(0.3, 1500, 16, 6, rng).astype(np.float32)))
    cases.append(("short", ar1(0.6, 600, 4, 9, rng)))
    for what, x in cases:
        xd = dev(x)
        outs = {}
        for sw in ("1", "0"):
            monkeypatch.setenv("MDT_TAIL_SRT", sw)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                e, r = pkg.ess_rhat(xd, kind="tail")
                outs[sw] = [host(e), host(r), host(pkg.rhat(xd, kind="tail")), host(pkg.ess(xd, kind="tail"))]
        monkeypatch.delenv("MDT_TAIL_SRT", raising=False)
        for a, b in zip(outs["1"], outs["0"]):
            assert np.array_equal(a, b, equal_nan=True), what
        assert np.array_equal(outs["1"][1], outs["1"][2], equal_nan=True), what + ": rhat(tail) == ess_rhat(tail).rhat"
        if x.shape[1] % 2 == 0 and x.shape[1] >= 4:  # rhat_nested(kind = tail) takes the same route (src/rhat_nested.jl:118-124)
            C = x.shape[1]
            ids = np.repeat(np.arange(C // 2), 2)
            nest = {}
            for sw in ("1", "0"):
                monkeypatch.setenv("MDT_TAIL_SRT", sw)
                nest[sw] = host(pkg.rhat_nested(xd, ids, kind="tail"))
            monkeypatch.delenv("MDT_TAIL_SRT", raising=False)
            assert np.array_equal(nest["1"], nest["0"], equal_nan=True), what + ": nested R-hat of the folded draws"
            inds = np.arange(C, dtype=np.int32).reshape(C // 2, 2).T.copy()
            assert_close(nest["1"], co.rhat_nested(np.asarray(x, dtype=np.float64), inds, kind="tail"), RTOL64 if x.dtype == np.float64 else 1e-6, what + ": nested tail against the oracle")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fused = [host(o) for o in pkg.bulk_tail(xd)]
            want = co.bulk_tail(np.asarray(x, dtype=np.float64)) if x.dtype == np.float64 else None
        assert np.array_equal(outs["1"][1], fused[3], equal_nan=True), what + ": R-hat tail of the fused request"
        tol = RTOL64 if x.dtype == np.float64 else 2e-5
        assert_close(outs["1"][0], fused[1], tol, what + ": ESS tail of the fused request")
        if want is not None:
            assert_close(outs["1"][0], want[1], RTOL64, what + ": ESS tail against the oracle")
            assert_close(outs["1"][1], want[3], RTOL64, what + ": R-hat tail against the oracle")


def test_tile_major_partition_and_ticket_order_change_no_bit(pkg, monkeypatch):
    """Round 3: the tile-major partition (one pass instead of histogram + partition, MDT_PA=0 switches it off) and the
    cost-ordered tickets of the persistent launch (MDT_NO_ORDER=1 switches them off) must not change a single bit of
    any output: same ranks, same fixed-point moments, every parameter computed independently of its ticket."""
    import torch
    rng = np.random.default_rng(31)
    cases = []
    x = ar1(0.5, 4000, 4, 24, rng)                 # n = 16 000: two-level ranking, 8 split chains of 2000
    x[:, :, 3] = np.round(x[:, :, 3], 1)           # heavy ties
    x[:, 1, 5] += 0.7
    cases.append(("f64 4 chains", dev(x), None))
    x16 = ar1(0.3, 1500, 16, 6, rng).astype(np.float32)   # 32 split chains, Float32 (config 5's shape in small)
    cases.append(("f32 16 chains", dev(x16), np.repeat(np.arange(1, 5), 4)))
    xo = ar1(0.2, 3001, 4, 5, rng)                 # odd draws: rows are dropped by the split -> the two-pass form serves it
    cases.append(("odd draws", dev(xo), None))
    big = torch.randn(2600, 2, 8000, dtype=torch.float64, device="cuda").permute(2, 1, 0)  # > 4 tickets per workgroup: ordered launches
    cases.append(("many params", big, None))

    def run_all(xd, ids):
        out = [host(o) for o in pkg.bulk_tail(xd)]
        out.append(host(pkg.rhat(xd)))
        out.append(host(pkg.ess(xd, kind="tail")))
        if ids is not None:
            out.append(host(pkg.rhat_nested(xd, ids, kind="rank")))
        return out

    for what, xd, ids in cases:
        monkeypatch.setenv("MDT_PA", "2")           # the tile-major partition wherever it can run (the default takes it for long chains only)
        monkeypatch.delenv("MDT_NO_ORDER", raising=False)
        a = run_all(xd, ids)
        a2 = run_all(xd, ids)                      # second round: the tickets follow the measured costs now
        monkeypatch.delenv("MDT_PA", raising=False)
        for u, v in zip(a, run_all(xd, ids)):
            assert np.array_equal(u, v, equal_nan=True), what + ": default choice"
        monkeypatch.setenv("MDT_PA", "0")
        b = run_all(xd, ids)
        monkeypatch.setenv("MDT_NO_ORDER", "1")
        c = run_all(xd, ids)
        for u, v, w, y in zip(a, a2, b, c):
            assert np.array_equal(u, v, equal_nan=True), what + ": ticket order"
            assert np.array_equal(u, w, equal_nan=True), what + ": tile-major partition"
            assert np.array_equal(u, y, equal_nan=True), what + ": no order"
    monkeypatch.delenv("MDT_PA", raising=False)
    monkeypatch.delenv("MDT_NO_ORDER", raising=False)


def test_fused_mcse_and_nested_rhat(pkg):
    """mdt_mcse_nested (BASELINE config 5 in one launch): bit-identical to mcse(mean) and rhat_nested called one after
    the other, within 1e-6 of the oracle (Float32: of its Float64-internal comparator), for every nested kind, on the
    config-5 per-parameter shape (10 000 draws x 16 chains, four superchains) and on small / odd / many-chain shapes"""
    from oracle import c_oracle
    rng = np.random.default_rng(41)
    ids = np.repeat(np.arange(1, 5), 4)
    inds = np.arange(16, dtype=np.int32).reshape(4, 4, order="F")
    shapes = [((10000, 16, 5), np.float32), ((10000, 16, 3), np.float64), ((301, 16, 4), np.float32), ((40, 16, 6), np.float64)]
    for shape, dt in shapes:
        x = (ar1(0.6, *shape, rng) + 0.2 * rng.standard_normal((1, 16, shape[2]))).astype(dt)
        xd = dev(x)
        for kind in KINDS:
            m, r = pkg.mcse_rhat_nested(xd, ids, kind=kind)
            m2, r2 = pkg.mcse(xd), pkg.rhat_nested(xd, ids, kind=kind)
            assert np.array_equal(host(m), host(m2), equal_nan=True), (shape, kind, "mcse bits")
            assert np.array_equal(host(r), host(r2), equal_nan=True), (shape, kind, "nested bits")
            assert_close(host(m), c_oracle.mcse_mean(x), 1e-6, f"{shape} mcse")
            assert_close(host(r), c_oracle.rhat_nested(x, inds, kind=kind), 1e-6, f"{shape} nested {kind}")
    # more split chains than the moment tables hold (one thread per split chain): found by the round-3 soak -- the nested
    # combination was skipped there in the fused mode
    xm = ar1(0.4, 63, 410, 2, rng)
    idm = np.repeat(np.arange(2), 205)
    indm = np.arange(410, dtype=np.int32).reshape(2, 205).T.copy()
    for kind in KINDS:
        m, r = pkg.mcse_rhat_nested(dev(xm), idm, kind=kind, split_chains=1)
        assert np.array_equal(host(r), host(pkg.rhat_nested(dev(xm), idm, kind=kind, split_chains=1))), kind
        assert np.array_equal(host(m), host(pkg.mcse(dev(xm), split_chains=1))), kind
        assert_close(host(r), c_oracle.rhat_nested(xm, indm, kind=kind, split_chains=1), 1e-8, f"many chains nested {kind}")
    # host memory, and the errors of the two functions it fuses
    x = ar1(0.3, 200, 16, 3, rng)
    m, r = pkg.mcse_rhat_nested(x, ids)
    assert_close(m, c_oracle.mcse_mean(x), RTOL64, "host mcse")
    assert_close(r, c_oracle.rhat_nested(x, inds, kind="rank"), RTOL64, "host nested")
    with pytest.raises(pkg.DimensionMismatch):
        pkg.mcse_rhat_nested(x, ids[:5])
    with pytest.raises(pkg.ArgumentError):
        pkg.mcse_rhat_nested(x, np.ones(16, dtype=int))
    with pytest.raises(pkg.DomainError):
        pkg.mcse_rhat_nested(x, ids, maxlag=0)
    m1, r1 = pkg.mcse_rhat_nested(dev(x), ids, devices=[0])          # the multi-GPU context takes it too
    assert np.array_equal(host(m1), m) and np.array_equal(host(r1), r)


def test_presharded_multi_gpu_context(pkg):
    """Round 3: a multi-GPU context takes its input PRE-SHARDED (mdt_args.x_shards: block i on GPU i, nothing of the size
    of x moves), strided layouts, and mcse(median | quantile); the blocks follow ONE partition rule (mdt_shard_bounds ==
    dist.shard_bounds).  The test box has one GPU: the path still runs -- worker thread, block, gather -- over it, and a
    single-GPU context is the bit-exact reference.  The uneven split itself is exercised through host memory, where the
    library cuts the blocks (every child is the same GPU there; device indices must differ in a real context)."""
    import torch
    rng = np.random.default_rng(47)
    x = ar1(0.5, 3000, 4, 7, rng)
    xd = dev(x)
    ids = [1, 1, 2, 2]
    ref = [host(o) for o in pkg.bulk_tail(xd)]
    got = [host(o) for o in pkg.bulk_tail([xd], devices=[0])]                      # pre-sharded, one block
    for a, b in zip(ref, got):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(host(pkg.rhat([xd])), host(pkg.rhat(xd)))                 # devices from the blocks
    assert np.array_equal(host(pkg.rhat_nested([xd], ids, devices=[0])), host(pkg.rhat_nested(xd, ids)))
    m, r = pkg.mcse_rhat_nested([xd], ids)
    m0, r0 = pkg.mcse_rhat_nested(xd, ids)
    assert np.array_equal(host(m), host(m0)) and np.array_equal(host(r), host(r0))
    # mcse(median | quantile) on a multi-GPU context (refused in round 2)
    for kind in (np.median, pkg.Quantile(0.9), np.std, np.mean):
        assert np.array_equal(host(pkg.mcse(xd, kind=kind, devices=[0])), host(pkg.mcse(xd, kind=kind)))
        assert np.array_equal(host(pkg.mcse([xd], kind=kind)), host(pkg.mcse(xd, kind=kind)))
    # strided (MCMCChains storage) straight through a multi-GPU context: no dense copy
    store = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 2, 0)))).cuda()
    view = store.permute(2, 0, 1)
    before = torch.cuda.memory_allocated()
    rv = pkg.rhat(view, devices=[0])
    assert torch.cuda.memory_allocated() - before < x.nbytes // 4
    assert np.array_equal(host(rv), host(pkg.rhat(xd)))
    # wrong block sizes / wrong device are refused
    with pytest.raises(pkg.DimensionMismatch):
        pkg.rhat([xd[:, :, :3], xd[:, :, 3:]], devices=[0, 0])                      # 7 parameters over 2 blocks are 4 + 3
    with pytest.raises(pkg.ArgumentError):
        pkg.rhat([xd], devices=[1])
    # host memory through the raw C-ABI on a context over [0]: x_shards in host memory
    import ctypes
    L = pkg._lib
    lib = L.load()
    dense = np.ascontiguousarray(np.transpose(x, (2, 1, 0)))
    out = [np.empty(7) for _ in range(4)]
    ptrs = (ctypes.c_void_p * 1)(dense.ctypes.data)
    a = L.MdtArgs(3000, 4, 7, L.MDT_F64, L.MDT_HOST, 0, 3, 0, 2, 0, 1, 250, 0.1, None, 0, 0)
    a.x_shards = ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_void_p))
    with L.using((0,), None) as h:
        assert lib.mdt_bulk_tail(h, ctypes.byref(a), None, *[o.ctypes.data for o in out]) == 0, lib.mdt_last_error(h)
    for a_, b_ in zip(ref, out):
        assert np.array_equal(a_, b_, equal_nan=True)


def test_fewer_draws_than_splits(pkg):
    """draws < split_chains: niter == 0, everything is NaN (mean of nothing, src/ess_rhat.jl:391-406) -- and the
    kernel's early exit from a parameter must not wedge the workgroup (it did once)"""
    import torch
    import warnings
    x = torch.randn(3, 2, 5, dtype=torch.float64).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = pkg.rhat(x, split_chains=4)
        e = pkg.ess_rhat(x, split_chains=4)
    assert np.isnan(host(r)).all() and np.isnan(host(e.ess)).all() and np.isnan(host(e.rhat)).all()


@pytest.mark.parametrize("cfg", ["", "tiny", "small", "big"])
def test_one_draw_per_split_chain(pkg, monkeypatch, cfg):
    """draws / split_chains == 1: the corrected variance of ONE value is 0/0 = NaN in the reference (Statistics.var,
    src/ess_rhat.jl:529-534, src/rhat_nested.jl:168-172), so every R-hat is NaN -- until round 6 the fold merge and the moments-only
    ranking returned variance 0 for such a chain ("all its scores are the same bits") and R-hat came out +Inf; found by the soak when
    tail-only requests began to take the fold merge (5 draws split in three, point masses).  Every kind, the fused request and nested
    R-hat, continuous data and ties, Float64 and Float32, every geometry, against the oracle."""
    import warnings
    from oracle import c_oracle as co
    if cfg:
        monkeypatch.setenv("MDT_CONFIG", cfg)
    rng = np.random.default_rng(539)
    for D, C, s in ((5, 5, 3), (1, 6, 1), (3, 4, 2)):
        for ties in (False, True):
            for dt in (np.float64, np.float32):
                x = rng.standard_normal((D, C, 4))
                if ties:
                    x[rng.random(x.shape) < 0.5] = 0.25
                x = x.astype(dt)
                xd = dev(x)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    for kind in ("rank", "bulk", "tail", "basic"):
                        got = host(pkg.rhat(xd, kind=kind, split_chains=s))
                        want = co.ess_rhat(np.asarray(x, dtype=np.float64), kind=kind, split_chains=s)[1]
                        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).all(), (D, C, s, ties, dt, kind, got, want)
                    fused = [host(o) for o in pkg.bulk_tail(xd, split_chains=s)]
                    assert all(np.isnan(o).all() for o in fused), (D, C, s, ties, dt, fused)
                    if C % 2 == 0 and C >= 4:
                        ids = np.repeat(np.arange(C // 2), 2)
                        inds = np.arange(C, dtype=np.int32).reshape(C // 2, 2).T.copy()
                        for kind in ("rank", "tail", "bulk"):
                            got = host(pkg.rhat_nested(xd, ids, kind=kind, split_chains=s))
                            want = co.rhat_nested(np.asarray(x, dtype=np.float64), inds, kind=kind, split_chains=s)
                            assert np.array_equal(np.isnan(got), np.isnan(want)), (D, C, s, ties, dt, "nested " + kind, got, want)


def test_fuzz_shapes_and_kinds(pkg, methods):
    """seeded random shapes / splits / data flavours through every kind, against the oracle"""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(20261003)
    flavours = ("normal", "ar", "ties", "skew", "offset")
    for it in range(36):
        D = int(rng.integers(6, 2500))
        C = int(rng.integers(1, 7))
        P = int(rng.integers(1, 5))
        s = int(rng.integers(1, 4))
        fl = flavours[it % len(flavours)]
        if fl == "normal":
            x = rng.standard_normal((D, C, P))
        elif fl == "ar":
            x = ar1(float(rng.uniform(-0.9, 0.95)), D, C, P, rng)
        elif fl == "ties":
            x = rng.integers(0, int(rng.integers(2, 40)), size=(D, C, P)).astype(np.float64)
        elif fl == "skew":
            x = np.exp(3.0 * rng.standard_normal((D, C, P)))
        else:
            x = rng.standard_normal((D, C, P)) + 3.0 * rng.standard_normal((1, C, 1)) + 1e6
        ml = int(rng.choice([1, 3, 50, 250]))
        m = ("direct", "fft", "bda")[it % 3]
        xd = dev(x)
        what = f"it={it} {fl} D={D} C={C} P={P} s={s} ml={ml} {m}"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for kind in KINDS:
                S, R = pkg.ess_rhat(xd, kind=kind, autocov_method=methods[m], split_chains=s, maxlag=ml)
                We, Wr = co.ess_rhat(x, kind=kind, autocov=m, split_chains=s, maxlag=ml)
                assert_close(host(S), We, 1e-8, what + f" ess {kind}")
                assert_close(host(R), Wr, 1e-8, what + f" rhat {kind}")
            for est, kind in (("median", np.median), ("mad", "mad"), ("std", np.std)):
                got = pkg.ess(xd, kind=kind, autocov_method=methods[m], split_chains=s, maxlag=ml)
                assert_close(host(got), co.ess_estimator(x, est, autocov=m, split_chains=s, maxlag=ml), 1e-8, what + " " + est)


def test_long_heavy_tailed_slices(pkg):
    """n = 240_000 per parameter, log-normal with sigma 4 (the sample quantiles crowd into a few grid cells: the
    class search takes its maximum number of steps) and a Cauchy column; both workgroup geometries rank it alike"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(424242)
    D, C = 40_000, 6
    x = np.stack([np.exp(4.0 * rng.standard_normal((D, C))), rng.standard_cauchy((D, C)), ar1(0.5, D, C, 1, rng)[:, :, 0]], axis=2)
    xd = dev(x)
    r, _ = pkg.rank_normalize(xd)
    for p in range(3):
        assert np.array_equal(host(r)[:, :, p], co.tiedrank(x[:, :, p].ravel(order="F")).reshape(D, C, order="F")), p
    got = [host(o) for o in pkg.bulk_tail(xd)]
    want = co.bulk_tail(x)
    for a, b, nm in zip(got, want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
        assert_close(a, b, RTOL64, nm)


def test_fft_in_lds_transform(pkg, methods, monkeypatch):
    """FFTAutocovMethod is served from lagged products by default; MDT_FFT_LDS=1 runs the in-LDS radix-2
    transform instead: both against the golden vectors"""
    import warnings
    monkeypatch.setenv("MDT_FFT_LDS", "1")
    for name in ("cfg1_randn_1000x4x10", "ar1_500x4x10", "constant_100x3x2", "odd_1001x3x3"):
        g = load_golden(name)
        xd = dev(g["x"])
        for ke, kr, kind, method, s, ml in _keys(g):
            if method != "fft":
                continue
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                S, R = pkg.ess_rhat(xd, kind=kind, autocov_method=methods["fft"], split_chains=s, maxlag=ml)
            assert_close(host(S), g[ke], RTOL64, f"{name} {ke} (in-LDS FFT)")
            assert_close(host(R), g[kr], RTOL64, f"{name} {kr} (in-LDS FFT)")


def test_fuzz_kwargs_f32_nested(pkg, methods):
    """seeded random keyword combinations (relative, tail_prob, maxlag), Float32 inputs and nested R-hat"""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(777)
    for it in range(24):
        D = int(rng.integers(12, 1500))
        K, S = int(rng.integers(1, 4)), int(rng.integers(2, 4))
        C = K * S
        P = int(rng.integers(1, 4))
        s = int(rng.integers(1, 3))
        f32 = it % 3 == 0
        x = ar1(float(rng.uniform(-0.5, 0.9)), D, C, P, rng) + 0.3 * rng.standard_normal((1, C, 1))
        if f32:
            x = x.astype(np.float32)
        tol = 1e-6 if f32 else 1e-8
        tp = float(rng.choice([0.02, 0.1, 0.3, 0.5]))
        rel = bool(it % 2)
        ml = int(rng.choice([2, 7, 100, 250]))
        m = ("direct", "fft", "bda")[it % 3]
        xd = dev(x)
        what = f"it={it} D={D} C={C} P={P} s={s} f32={f32} tp={tp} rel={rel} ml={ml} {m}"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = [host(o) for o in pkg.bulk_tail(xd, relative=rel, autocov_method=methods[m], split_chains=s, maxlag=ml, tail_prob=tp)]
            want = co.bulk_tail(x, autocov=m, split_chains=s, maxlag=ml, relative=rel, tail_prob=tp)
            for a, b, nm in zip(got, want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
                assert_close(a, b, tol, what + " " + nm)
            ids = np.repeat(np.arange(1, S + 1), K)
            inds = np.arange(C, dtype=np.int32).reshape(K, S, order="F")
            for kind in KINDS:
                assert_close(host(pkg.rhat_nested(xd, ids, kind=kind, split_chains=s)),
                             co.rhat_nested(x, inds, kind=kind, split_chains=s), tol, what + " nested " + kind)
            assert_close(host(pkg.mcse(xd, autocov_method=methods[m], split_chains=s, maxlag=ml)),
                         co.mcse_mean(x, autocov=m, split_chains=s, maxlag=ml), tol, what + " mcse")


@pytest.mark.parametrize("name", golden_names(diag=True))
def test_bfmi_gelmandiag_golden(pkg, name):
    """bfmi (src/bfmi.jl:38-45) and gelmandiag (src/gelmandiag.jl:1-80) through the C-ABI against the committed
    vectors, from device and from host memory; diag_bfmi_reference = the reference's known answers
    (test/bfmi.jl:2-41)."""
    g = load_golden(name)
    if name == "diag_bfmi_reference":
        assert abs(pkg.bfmi(g["energy_a"]) - 0.6) < 1e-12
        assert abs(pkg.bfmi(g["energy_b"]) - 0.2406937229) < 1e-9
        assert abs(pkg.bfmi([int(v) for v in g["energy_b"]]) - 0.2406937229) < 1e-9
        em = np.repeat(g["energy_b"][:, None], 4, axis=1)
        assert_close(pkg.bfmi(em), np.full(4, 0.2406937229), 1e-8, name)
        assert_close(pkg.bfmi(em.T, dims=2), pkg.bfmi(em), 1e-15, name + " dims=2")  # test/bfmi.jl:41
        return
    x = load_golden(name[5:])["x"]
    f32 = x.dtype == np.float32
    tol = 2e-6 if f32 else RTOL64
    for src in (dev(x), x):
        for k, want in g.items():
            if k.startswith("gelman|"):
                _, alpha, which = k.split("|")
                r = pkg.gelmandiag(src, alpha=float(alpha))
                got = host(getattr(r, which))
                assert got.dtype == np.float64 and got.shape == (x.shape[2],)
                assert_close(got, want, tol, f"{name} {k}")
        e = x.reshape(x.shape[0], -1, order="F")
        got = host(pkg.bfmi(dev(e[:, :, None])[:, :, 0] if src is not x else e))
        assert_close(got, g["bfmi"], 2e-5 if f32 else RTOL64, name + " bfmi")


def test_gelmandiag_bfmi_shapes_and_errors(pkg):
    """result types (test/gelmandiag.jl:8-16), the two-chain requirement (:29), big seeded input vs the oracle"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(77)
    x = rng.standard_normal((100, 2, 4))
    r = pkg.gelmandiag(x)
    assert r._fields == ("psrf", "psrfci") and r.psrf.shape == (4,) and r.psrf.dtype == np.float64
    with pytest.raises(pkg.ErrorException):
        pkg.gelmandiag(x[:, :1, :])
    x = ar1(0.6, 4000, 6, 300, rng) + rng.standard_normal((1, 6, 300)) * 0.05
    want = co.gelmandiag(x, alpha=0.05)
    got = pkg.gelmandiag(dev(x))
    assert_close(host(got.psrf), want[0], RTOL64, "psrf")
    assert_close(host(got.psrfci), want[1], RTOL64, "psrfci")
    e = rng.standard_normal((777, 5)).cumsum(axis=0)
    assert_close(host(pkg.bfmi(e)), co.bfmi(e), RTOL64, "bfmi")
    one = rng.standard_normal(1)
    assert math.isnan(pkg.bfmi(one))  # var of one draw is NaN
    const = np.ones((50, 3, 2))
    r = pkg.gelmandiag(const)  # W = B = 0: NaN estimates, no F quantile taken (src/gelmandiag.jl:49)
    assert np.isnan(r.psrf).all() and np.isnan(r.psrfci).all()


def test_mcse_quantile_nan_and_async_passes(pkg):
    """mcse(kind = quantile(p)) on data holding a NaN: Statistics.quantile throws (src/mcse.jl:84 -> ess_rhat.jl:655);
    the flag raised by the first device pass must survive the second one.  kind = median propagates NaN instead."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(5)
    x = rng.standard_normal((400, 3, 6))
    x[17, 1, 2] = np.nan
    with pytest.raises(pkg.ArgumentError):
        pkg.mcse(dev(x), kind=pkg.Quantile(0.3))
    with pytest.raises(pkg.ArgumentError):
        pkg.mcse(x, kind=pkg.Quantile(0.3))
    got = host(pkg.mcse(dev(x), kind=np.median))
    want = co.mcse_quantile(x, "median")
    assert np.isnan(got[2]) and np.isnan(want[2])
    assert_close(got, want, RTOL64, "mcse median with a NaN parameter")
    # many parameters, device and host memory, against the oracle
    x = ar1(0.5, 600, 4, 700, rng)
    for kind, est, p in ((np.median, "median", 0.5), (pkg.Quantile(0.9), "quantile", 0.9)):
        want = co.mcse_quantile(x, est, prob=p)
        assert_close(host(pkg.mcse(dev(x), kind=kind)), want, RTOL64, f"mcse {est} device")
        assert_close(host(pkg.mcse(x, kind=kind)), want, RTOL64, f"mcse {est} host")


def test_concurrent_streams_do_not_share_scratch(pkg):
    """requests issued on different HIP streams may overlap on the device: every (device, stream) pair owns its
    context (scratch, ticket, status word), so the results are those of the serial calls, bit for bit"""
    import torch
    rng = np.random.default_rng(11)
    xa = dev(ar1(0.7, 2000, 4, 1500, rng))
    xb = dev(rng.standard_normal((2000, 4, 1500)))
    ref_a = [host(t) for t in pkg.bulk_tail(xa)]
    ref_b = [host(t) for t in pkg.bulk_tail(xb)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        with torch.cuda.stream(s1):
            ra = pkg.bulk_tail(xa, sync=False)
        with torch.cuda.stream(s2):
            rb = pkg.bulk_tail(xb, sync=False)
        outs.append((ra, rb))
    torch.cuda.synchronize()
    for ra, rb in outs:
        for got, want in zip(ra, ref_a):
            assert np.array_equal(host(got), want, equal_nan=True)
        for got, want in zip(rb, ref_b):
            assert np.array_equal(host(got), want, equal_nan=True)
    assert len({k for k in pkg._lib._ctxs if k[0] == 0}) <= pkg._lib.MAX_CTX_PER_DEVICE


def _full_size_properties(pkg, D, C, P, n_oracle, method=None):
    """size-independent properties on a BASELINE-sized, device-generated array (bench.synth): position independence,
    exact invariance under an exact monotone map, rank == max(bulk, tail), and the oracle on a spread sample"""
    import torch
    from bench import synth
    from oracle import c_oracle as co
    kw = {} if method is None else {"autocov_method": method}
    okw = {} if method is None else {"autocov": "fft"}
    x = synth(D, C, P, torch.device("cuda", 0), 4321).permute(2, 1, 0)
    full = [host(o) for o in pkg.bulk_tail(x, **kw)]
    assert all(f.shape == (P,) for f in full)
    # (a) any sub-batch reproduces the bits of the full batch
    sel = torch.arange(3, P, max(1, P // 97), device=x.device)
    sub = [host(o) for o in pkg.bulk_tail(x[:, :, sel].permute(2, 1, 0).contiguous().permute(2, 1, 0), **kw)]
    for f, s in zip(full, sub):
        assert np.array_equal(f[sel.cpu().numpy()], s, equal_nan=True)
    # (b) x -> 2x is exact and strictly increasing: ranks, folds (|2x - 2 med| = 2 |x - med|) and indicators do not move
    x2 = x * 2.0
    for f, s in zip(full, pkg.bulk_tail(x2, **kw)):
        assert np.array_equal(f, host(s), equal_nan=True)
    del x2
    # (c) rhat(kind = rank) == max(rhat_bulk, rhat_tail), ess(kind = bulk) == the fused bulk ESS  (src/ess_rhat.jl:416-420)
    assert np.array_equal(host(pkg.rhat(x, kind="rank")), np.maximum(full[2], full[3]))
    assert np.array_equal(host(pkg.ess(x, kind="bulk", **kw)), full[0])
    # (d) the oracle on a sample spread over the array (IID and AR(1) parameters alternate in blocks of four)
    pick = np.unique(np.linspace(0, P - 1, n_oracle).astype(np.int64) // 8 * 8 + np.arange(n_oracle) % 8).clip(0, P - 1)
    xs = x[:, :, torch.as_tensor(pick, device=x.device)].cpu().numpy()
    for f, w in zip(full, co.bulk_tail(xs, **okw)):
        assert_close(f[pick], w, RTOL64, f"full size {D}x{C}x{P}")
    # (e) the IID half (global parameters p with p mod 8 < 4, bench.synth) behaves like IID draws (test/ess_rhat.jl:233-235
    # asks for 10 % on the mean)
    iid = (np.arange(P) % 8) < 4
    assert abs(np.mean(full[0][iid]) / (D * C) - 1) < 0.1 and abs(np.mean(full[2][iid]) - 1) < 0.01
    assert np.all(np.isfinite(full[0])) and np.all(full[2] > 0.99)


def test_full_size_config2_properties(pkg):
    """BASELINE config 2: 10 000 draws x 4 chains x 10 000 params Float64 (3.2 GB)"""
    _full_size_properties(pkg, 10000, 4, 10000, 48)


def test_full_size_config3_properties(pkg, methods):
    """BASELINE config 3: 100 000 draws x 8 chains x 1 000 params Float64 (6.4 GB), FFTAutocovMethod forced"""
    _full_size_properties(pkg, 100000, 8, 1000, 3, methods["fft"])


@pytest.mark.parametrize("cfg", ["tiny", "small", "big"])
def test_every_geometry_on_the_golden_vectors(pkg, methods, monkeypatch, cfg):
    """the three workgroup geometries (256 / 512 / 1024 threads; 40 / 80 / 160 KiB of LDS) are compiled from one
    source and chosen by slice size; forced here (MDT_CONFIG is read per request), each reproduces the golden
    vectors, and R-hat -- fixed-point moments -- is bit-identical across them"""
    import warnings
    monkeypatch.setenv("MDT_CONFIG", cfg)
    rhats = {}
    for name in ("cfg1_randn_1000x4x10", "ar1_500x4x10", "ties_200x4x6", "odd_1001x3x3", "special_50x2x3"):
        g = load_golden(name)
        xd = dev(g["x"])
        for ke, kr, kind, method, s, ml in _keys(g):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r = pkg.ess_rhat(xd, kind=kind, autocov_method=methods[method], split_chains=s, maxlag=ml)
            assert_close(host(r.ess), g[ke], RTOL64, f"{cfg} {name} {ke}")
            assert_close(host(r.rhat), g[kr], RTOL64, f"{cfg} {name} {kr}")
            rhats[(name, kr)] = host(r.rhat)
    monkeypatch.delenv("MDT_CONFIG")
    for (name, kr), want in rhats.items():
        _, kind, method, s, ml = kr.split("|")
        if kind in ("basic", "tail") or method != "direct" or int(ml) != 250:
            continue  # (raw-draw moments are plain double sums; one method and maxlag is enough)
        g = load_golden(name)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = host(pkg.rhat(dev(g["x"]), kind=kind, split_chains=int(s)))
        assert np.array_equal(got, want, equal_nan=True), (cfg, name, kr)


def _fold_cases(rng, D, C):
    """parameters that exercise every branch of fold_merge_phase: no ties, folds tied across the median (x distinct),
    folds rounded together, ties in x (general ranking), one -Inf / +Inf, AR(1)"""
    n = D * C
    cols = [rng.standard_normal(n), ar1(0.8, D, C, 1, rng)[:, :, 0].T.reshape(-1)]
    d = np.abs(rng.standard_normal(n // 2)) + 1e-3
    cols.append(rng.permutation(np.concatenate([d, -d] + ([np.zeros(1)] if n % 2 else []))))  # median exactly 0: every fold is a tied pair
    part = rng.standard_normal(n)
    k = n // 10
    part[:k] = 2.0 + np.abs(part[:k])
    part[k:2 * k] = -part[:k]
    part[2 * k:] = part[2 * k:] - np.median(part[2 * k:])                       # a tenth of the folds tied in pairs ...
    cols.append(rng.permutation(part))
    cols.append(rng.permutation(np.concatenate([1e-17 * np.arange(1, n // 2 + 1), 1.0 + np.abs(rng.standard_normal(n - n // 2))])))
    few = rng.uniform(0.25, 0.75, n)
    few[:30] = 1e-17 * np.arange(1, 31)                                         # ... and a few rounded together (the exact pass)
    cols.append(rng.permutation(few))
    cols.append(np.round(rng.standard_normal(n) * 4) / 4)                       # heavy ties in x itself: the general ranking
    cols.append(np.round(rng.standard_normal(n) * 1e5) / 1e5)                   # a few hundred tied pairs in x: sorted side by side, exact pass
    inf = rng.standard_normal(n)
    inf[17] = np.inf
    inf[n - 5] = -np.inf
    cols.append(inf)
    odd = rng.standard_normal(n)
    cols.append(odd)
    return np.stack([c.reshape(C, D).T for c in cols], axis=2)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fold_merge_matches_general_ranking(pkg, monkeypatch, dtype):
    """rhat(kind = tail) from the merge of the sorted slice (fold_merge_phase) == from the general ranking of
    |x - median|, bit for bit (exact fixed-point moments), and both match the oracle"""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(2024)
    for D, C in ((5000, 4), (4001, 5)):
        x = _fold_cases(rng, D, C).astype(dtype)
        xd = dev(x)
        outs = {}
        for sw in ("1", "0"):
            monkeypatch.setenv("MDT_FOLD_MERGE", sw)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                outs[sw] = [host(o) for o in pkg.bulk_tail(xd)] + [host(pkg.rhat(xd, kind="rank"))]
                ids = np.repeat(np.arange(1, 3), C // 2) if C % 2 == 0 else None
                if ids is not None:
                    outs[sw].append(host(pkg.rhat_nested(xd, ids, kind="rank")))
        monkeypatch.delenv("MDT_FOLD_MERGE")
        for a, b in zip(outs["1"], outs["0"]):
            assert np.array_equal(a, b, equal_nan=True)
        want = co.bulk_tail(x)
        for a, b, nm in zip(outs["1"], want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
            assert_close(a, b, RTOL64 if dtype == np.float64 else 1e-6, f"{nm} {D}x{C} {np.dtype(dtype).name}")


def test_rhat_many_chains(pkg):
    """more split chains than the LDS moment tables hold (chains*split_chains > 256): the reference's own suite runs
    rhat_nested on 10 draws x 2048 chains x 32 params in 8 and 16 superchains and asks 1 < R < 1.001
    (test/rhat_nested.jl:157-169); rhat of every kind takes the same route (ESS: test_ess_many_chains)"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(2048)
    x = rng.standard_normal((10, 2048, 32))
    xd = dev(x)
    for nsuper in (8, 16):
        ids = np.repeat(np.arange(1, nsuper + 1), 2048 // nsuper)
        inds = np.arange(2048, dtype=np.int32).reshape(2048 // nsuper, nsuper, order="F")
        for kind in KINDS:
            for s in (1, 2):
                got = host(pkg.rhat_nested(xd, ids, kind=kind, split_chains=s))
                assert_close(got, co.rhat_nested(x, inds, kind=kind, split_chains=s), RTOL64, f"nested {nsuper} {kind} {s}")
        R = host(pkg.rhat_nested(xd, ids))
        assert R.min() > 1 and R.max() < 1.001
    for kind in KINDS:
        for s in (1, 2, 3):
            got = host(pkg.rhat(xd, kind=kind, split_chains=s))
            assert_close(got, co.ess_rhat(x, kind=kind, split_chains=s)[1], RTOL64, f"rhat {kind} {s}")
    # longer chains, fewer of them, Float32, one superchain layout that is not contiguous
    y = (ar1(0.5, 300, 320, 5, rng) + 0.05 * rng.standard_normal((1, 320, 1))).astype(np.float32)
    ids = np.tile(np.arange(4), 80)
    inds = np.stack([np.nonzero(ids == k)[0] for k in range(4)], axis=1).astype(np.int32)
    for kind in KINDS:
        assert_close(host(pkg.rhat_nested(dev(y), ids, kind=kind)), co.rhat_nested(y, inds, kind=kind), 1e-6, f"nested f32 {kind}")
        assert_close(host(pkg.rhat(dev(y), kind=kind)), co.ess_rhat(y, kind=kind)[1], 1e-6, f"rhat f32 {kind}")


def test_ess_many_chains(pkg, methods):
    """ESS of every kind, the estimator kinds and mcse with more split chains than the LDS moment tables hold
    (chains*split_chains > 256; refused until round 2): one thread per split chain (many_phase) against the oracle;
    rhat(x) and ess_rhat(x)[1] stay the same bits (test/ess_rhat.jl:167-204)"""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(4096)
    cases = [(ar1(0.6, 40, 320, 3, rng) + 0.1 * rng.standard_normal((1, 320, 1)), RTOL64),           # m = 640, 20 draws per split chain
             (ar1(0.3, 300, 200, 2, rng).astype(np.float32), 1e-6),                                  # two-level ranking (n = 60000), Float32
             (np.round(ar1(0.5, 64, 150, 2, rng) * 4) / 4, RTOL64)]                                  # ties
    for x, tol in cases:
        xd = dev(x)
        for s in (1, 2, 3):
            if x.shape[1] * s <= 256:
                continue
            for kind in KINDS:
                for mname in ("direct", "fft", "bda"):
                    if mname != "direct" and (s != 2 or kind == "rank"):
                        continue
                    S, R = [host(o) for o in pkg.ess_rhat(xd, kind=kind, split_chains=s, autocov_method=methods[mname])]
                    wS, wR = co.ess_rhat(x, kind=kind, split_chains=s, autocov=mname)
                    assert_close(S, wS, tol, f"ess {x.shape} {kind} {s} {mname}")
                    assert_close(R, wR, tol, f"rhat {x.shape} {kind} {s} {mname}")
                    assert np.array_equal(R, host(pkg.rhat(xd, kind=kind, split_chains=s)), equal_nan=True)
                    if kind != "rank":
                        assert np.array_equal(S, host(pkg.ess(xd, kind=kind, split_chains=s, autocov_method=methods[mname])), equal_nan=True)
            got = [host(o) for o in pkg.bulk_tail(xd, split_chains=s)]
            for g_, w_, nm in zip(got, co.bulk_tail(x, split_chains=s), ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
                assert_close(g_, w_, tol, f"bulk_tail {x.shape} {s} {nm}")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for est, kind, prob in (("median", np.median, 0.5), ("std", np.std, 0.5), ("mad", "mad", 0.5), ("quantile", pkg.Quantile(0.2), 0.2)):
                assert_close(host(pkg.ess(xd, kind=kind)), co.ess_estimator(x, est, prob=prob), tol, f"ess {est} {x.shape}")
            assert_close(host(pkg.ess(xd, kind="bulk", relative=True, maxlag=7)), co.ess_rhat(x, kind="bulk", relative=True, maxlag=7)[0], tol, "relative, maxlag")
            assert_close(host(pkg.mcse(xd)), co.mcse_mean(x), tol, f"mcse mean {x.shape}")
            assert_close(host(pkg.mcse(xd, kind=np.std)), co.mcse_std(x), tol, f"mcse std {x.shape}")
            assert_close(host(pkg.mcse(xd, kind=np.median)), co.mcse_quantile(x, "median"), tol, f"mcse median {x.shape}")
            assert_close(host(pkg.mcse(xd, kind=pkg.Quantile(0.9))), co.mcse_quantile(x, "quantile", 0.9), tol, f"mcse q {x.shape}")
    # a constant split chain: NaN under FFTAutocovMethod only (src/ess_rhat.jl:192), and too few draws: NaN ESS, R-hat kept
    x = ar1(0.2, 24, 300, 2, rng)
    x[:12, 5, 0] = 1.5
    for mname in ("direct", "fft"):
        S = host(pkg.ess(dev(x), kind="basic", autocov_method=methods[mname]))
        assert_close(S, co.ess_rhat(x, kind="basic", autocov=mname)[0], RTOL64, "constant split chain " + mname)
    assert np.isnan(host(pkg.ess(dev(x), kind="basic", autocov_method=methods["fft"]))[0])
    x = rng.standard_normal((8, 2048, 4))
    S, R = [host(o) for o in pkg.ess_rhat(dev(x))]
    assert np.all(np.isnan(S))
    assert_close(R, co.ess_rhat(x)[1], RTOL64, "rhat with too few draws for ESS")
    # hardly any draws (down to one per split chain: the 2 m moments then need more scratch than the slice), NaN, Inf, constant
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for D in (1, 2, 3, 4, 5, 6, 11):
            for s in (1, 2, 3):
                x = rng.standard_normal((D, 300, 2))
                for kind in KINDS:
                    try:
                        want = co.ess_rhat(x, kind=kind, split_chains=s)
                    except co.OracleError:  # (e.g. more splits than draws)
                        with pytest.raises((pkg.ArgumentError, pkg.DomainError, pkg.DimensionMismatch)):
                            pkg.ess_rhat(dev(x), kind=kind, split_chains=s)
                        continue
                    got = [host(o) for o in pkg.ess_rhat(dev(x), kind=kind, split_chains=s)]
                    assert_close(got[0], want[0], RTOL64, f"ess D={D} s={s} {kind}")
                    assert_close(got[1], want[1], RTOL64, f"rhat D={D} s={s} {kind}")
        x = rng.standard_normal((30, 400, 3))
        x[3, 7, 0] = np.nan
        x[5, 9, 1] = np.inf
        x[:, :, 2] = 2.0
        for kind in KINDS:
            try:
                want = co.ess_rhat(x, kind=kind)
            except co.OracleError:  # Statistics.quantile throws on NaN (tail ESS)
                with pytest.raises(pkg.ArgumentError):
                    pkg.ess_rhat(dev(x), kind=kind)
                want = co.ess_rhat(x[:, :, 1:], kind=kind)
                got = [host(o) for o in pkg.ess_rhat(dev(x[:, :, 1:]), kind=kind)]
            else:
                got = [host(o) for o in pkg.ess_rhat(dev(x), kind=kind)]
            assert_close(got[0], want[0], RTOL64, f"ess nan/inf/const {kind}")
            assert_close(got[1], want[1], RTOL64, f"rhat nan/inf/const {kind}")


def test_rhat_nested_permutation_invariance_bitwise(pkg):
    """R == R_perm under a simultaneous permutation of chains and superchain ids, with == as the reference's suite
    has it (test/rhat_nested.jl:113-130): the sums over the chains of a superchain run in sorted order"""
    rng = np.random.default_rng(77)
    for nsuper in (8, 16):
        for per in (2, 4):
            for s in (1, 2):
                nch = nsuper * per
                x = rng.standard_normal((100, nch, 10))
                ids = np.repeat(np.arange(1, nsuper + 1), per)
                perm = rng.permutation(nch)
                for kind in KINDS:
                    R = host(pkg.rhat_nested(dev(x), ids, kind=kind, split_chains=s))
                    Rp = host(pkg.rhat_nested(dev(x[:, perm, :]), ids[perm], kind=kind, split_chains=s))
                    assert np.array_equal(R, Rp), (nsuper, per, s, kind)
    # many chains: the one-thread-per-chain route
    x = rng.standard_normal((10, 2048, 8))
    ids = np.repeat(np.arange(1, 9), 256)
    perm = rng.permutation(2048)
    for kind in KINDS:
        assert np.array_equal(host(pkg.rhat_nested(dev(x), ids, kind=kind)), host(pkg.rhat_nested(dev(x[:, perm, :]), ids[perm], kind=kind)))
    # label values and types do not matter (:102-111)
    x = rng.standard_normal((100, 4, 10))
    for kind in KINDS:
        R = host(pkg.rhat_nested(dev(x), [1, 1, 2, 2], kind=kind))
        assert np.array_equal(R, host(pkg.rhat_nested(dev(x), [42, 42, 99, 99], kind=kind)))
        assert np.array_equal(R, host(pkg.rhat_nested(dev(x), ["a", "a", "b", "b"], kind=kind)))


def test_missing_and_param_axes_on_device(pkg):
    """Union{Missing,T} input (param-wise missing, test/ess_rhat.jl:115-132) as numpy masked arrays, and parameter
    axes of any rank: slices of a 4-d array reproduce the bits of the whole call (:167-204)"""
    import warnings
    rng = np.random.default_rng(31)
    for kind in KINDS:
        x = np.ma.MaskedArray(rng.standard_normal((1000, 4, 3)), mask=False)
        x[0, 0, 0] = np.ma.masked
        S1 = pkg.ess(x, kind="bulk" if kind == "rank" else kind)
        R1 = pkg.rhat(x, kind=kind)
        S2, R2 = pkg.ess_rhat(x, kind=kind)
        for o in (S1, R1, S2, R2):
            assert np.ma.getmaskarray(o)[0] and not np.ma.getmaskarray(o)[1:].any()
            assert np.all(np.isfinite(np.asarray(o[1:], dtype=np.float64)))
    x = rng.standard_normal((1000, 4, 10, 3))
    import torch
    xt = torch.from_numpy(x).cuda()
    for kind in KINDS:
        for s in (1, 2):
            R1 = host(pkg.rhat(xt, kind=kind, split_chains=s))
            assert R1.shape == (10, 3)
            for i in range(3):
                assert np.array_equal(host(pkg.rhat(xt[:, :, :, i], kind=kind, split_chains=s)), R1[:, i])
                for j in (0, 4, 9):
                    assert pkg.rhat(xt[:, :, j, i], kind=kind, split_chains=s) == R1[j, i]
            for method in (pkg.AutocovMethod(), pkg.BDAAutocovMethod()):
                for ml in (100, 10):
                    ke = "bulk" if kind == "rank" else kind
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        S1 = host(pkg.ess(xt, kind=ke, split_chains=s, autocov_method=method, maxlag=ml))
                        S2, R2 = pkg.ess_rhat(xt, kind=kind, split_chains=s, autocov_method=method, maxlag=ml)
                    assert np.array_equal(S1, host(S2)) and np.array_equal(R1, host(R2))
                    for i in range(3):
                        r = pkg.ess_rhat(xt[:, :, :, i], kind=kind, split_chains=s, autocov_method=method, maxlag=ml)
                        assert np.array_equal(host(r.ess), S1[:, i]) and np.array_equal(host(r.rhat), R1[:, i])
                        r = pkg.ess_rhat(xt[:, :, 7, i], kind=kind, split_chains=s, autocov_method=method, maxlag=ml)
                        assert r.ess == S1[7, i] and r.rhat == R1[7, i]


def test_async_requests_keep_their_flags(pkg):
    """sync = False cannot raise: a NaN under a quantile (the reference throws ArgumentError, Statistics.quantile at
    src/ess_rhat.jl:655) stays as a sticky flag on the context until mdt_ctx_status reports it -- through any number
    of later launches -- and the affected output is NaN"""
    import torch
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2000, 4, 50))
    clean = dev(x)
    x[17, 1, 5] = np.nan
    bad = dev(x)
    assert pkg.status() == 0
    with pytest.raises(pkg.ArgumentError):
        pkg.bulk_tail(bad)                                  # synchronous: raised here, nothing left behind
    assert pkg.status() == 0
    out = pkg.bulk_tail(bad, sync=False)
    for _ in range(70):                                      # more launches than parameter-ring slots, none synchronising
        pkg.bulk_tail(clean, sync=False)
    torch.cuda.synchronize()
    assert np.isnan(host(out[1])[5]) and np.isfinite(host(out[1])[[4, 6]]).all()
    assert pkg.status(clear=False) == pkg.FLAG_NAN_QUANTILE
    assert pkg.status() == pkg.FLAG_NAN_QUANTILE             # ... and cleared by this call
    assert pkg.status() == 0


def test_out_buffers_are_validated(pkg):
    import torch
    x = dev(np.random.default_rng(3).standard_normal((500, 4, 6)))
    good = [torch.empty(6, dtype=torch.float64, device="cuda") for _ in range(4)]
    assert pkg.bulk_tail(x, out=good) is good
    ref = pkg.bulk_tail(x)
    for a, b in zip(good, ref):
        assert torch.equal(a, b)
    for bad in (good[:3], [torch.empty(6, dtype=torch.float32, device="cuda")] * 4, [torch.empty(5, dtype=torch.float64, device="cuda")] * 4,
                [torch.empty(6, dtype=torch.float64)] * 4, [torch.empty(12, dtype=torch.float64, device="cuda")[::2]] * 4,
                [np.empty(6)] * 4):
        with pytest.raises((pkg.ArgumentError, pkg.DimensionMismatch)):
            pkg.bulk_tail(x, out=bad)


def test_multi_gpu_context_and_current_device(pkg):
    """mdt_ctx_create_multi over every GPU of the box (one on the test box: the path still runs -- blocks, threads,
    gather): bit-equal to the single-context result, from device and from host memory; and no call moves the calling
    thread's current device"""
    import torch
    ndev = torch.cuda.device_count()
    devices = list(range(ndev))
    rng = np.random.default_rng(12)
    x = ar1(0.4, 1200, 4, 37, rng)
    xd = dev(x)
    cur = torch.cuda.current_device()
    ref = [host(o) for o in pkg.bulk_tail(xd)]
    for src in (xd, x):
        got = [host(o) for o in pkg.bulk_tail(src, devices=devices)]
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True)
    ids = [1, 1, 2, 2]
    assert np.array_equal(host(pkg.rhat_nested(xd, ids, devices=devices)), host(pkg.rhat_nested(xd, ids)))
    assert np.array_equal(host(pkg.mcse(xd, devices=devices)), host(pkg.mcse(xd)))
    assert np.array_equal(host(pkg.mcse(xd, kind=np.std, devices=devices)), host(pkg.mcse(xd, kind=np.std)))
    assert np.array_equal(host(pkg.ess(xd, kind=np.median, devices=devices)), host(pkg.ess(xd, kind=np.median)))
    assert np.array_equal(host(pkg.mcse(xd, kind=np.median, devices=devices)), host(pkg.mcse(xd, kind=np.median)))  # (refused until round 3)
    r = pkg.ess_rhat(x, devices=devices)
    r0 = pkg.ess_rhat(xd)
    assert np.array_equal(host(r.ess), host(r0.ess)) and np.array_equal(host(r.rhat), host(r0.rhat))
    assert pkg.status(devices=devices) == 0
    assert torch.cuda.current_device() == cur
    if ndev > 1:   # a tensor on the last GPU: the call must leave the current device alone
        with torch.cuda.device(0):
            xl = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 1, 0)))).to(f"cuda:{ndev - 1}").permute(2, 1, 0)
            got = [host(o) for o in pkg.bulk_tail(xl)]
            assert torch.cuda.current_device() == 0
            for a, b in zip(got, ref):
                assert np.array_equal(a, b, equal_nan=True)


def test_shard_size_config4_properties(pkg):
    """BASELINE config 4's per-GPU shard: 10 000 draws x 4 chains x 125 000 params Float64 (40 GB resident)"""
    _full_size_properties(pkg, 10000, 4, 125000, 24)


def test_shard_size_config5_f32_nested_and_mcse(pkg):
    """BASELINE config 5's per-GPU shard: 10 000 draws x 16 chains (4 superchains) x 12 500 params Float32 (8 GB):
    mcse(kind = mean) and rhat_nested(kind = rank) -- sub-batch bits, exact invariance under x -> 2x, the oracle on a sample"""
    import torch
    from oracle import c_oracle as co
    D, C, P = 10000, 16, 12500
    from bench import synth_cfg5
    x = synth_cfg5(D, C, P, torch.device("cuda", 0), 55)
    half = P // 2
    xv = x.permute(2, 1, 0)
    ids = np.repeat(np.arange(1, 5), 4)
    inds = np.arange(16, dtype=np.int32).reshape(4, 4, order="F")
    full = [host(pkg.mcse(xv)), host(pkg.rhat_nested(xv, ids, kind="rank"))]
    assert all(f.shape == (P,) and np.all(np.isfinite(f)) for f in full)
    sel = torch.arange(5, P, P // 61, device="cuda")
    sub_x = x[sel].contiguous().permute(2, 1, 0)
    sub = [host(pkg.mcse(sub_x)), host(pkg.rhat_nested(sub_x, ids, kind="rank"))]
    for f, s_ in zip(full, sub):
        assert np.array_equal(f[sel.cpu().numpy()], s_)
    x2 = xv * 2.0
    assert np.array_equal(host(pkg.rhat_nested(x2, ids, kind="rank")), full[1])      # ranks and folds do not move
    assert_close(host(pkg.mcse(x2)), 2.0 * full[0], 1e-6)
    del x2
    pick = np.linspace(0, P - 1, 6).astype(np.int64)
    xs = x[torch.as_tensor(pick, device="cuda")].permute(2, 1, 0).cpu().numpy()
    assert_close(full[0][pick], co.mcse_mean(xs), 1e-6, "cfg5 mcse")
    assert_close(full[1][pick], co.rhat_nested(xs, inds, kind="rank"), 1e-6, "cfg5 nested")
    assert np.all(full[1] >= 1.0) and np.all(full[1] < 1.01)  # the shared initial values are forgotten after 10 000 draws


def test_long_chains_windowed_layout(pkg, methods, monkeypatch):
    """chains that do not fit the LDS: the ranking keeps its scores in partition order with a run table and halo strips,
    series_win_phase gathers time windows from them (no time-order scatter).  Against the oracle, and against the
    time-order route (MDT_WINOUT=0): the moments differ in their fixed-point shift only, so 1e-12; chain starts that
    do not fall on partition-tile boundaries, three splits, all three autocovariance methods, maxlag up to the cap,
    Float32, both geometries, R-hat only"""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(4242)
    cases = [(30011, 3, 2, np.float64, 250), (20000, 2, 3, np.float64, 250), (70001, 2, 2, np.float64, 1000), (26000, 5, 1, np.float32, 100)]
    for D, C, s, dt, ml in cases:
        x = np.concatenate([rng.standard_normal((D, C, 2)), ar1(0.9, D, C, 2, rng), ar1(-0.4, D, C, 1, rng)], axis=2).astype(dt)
        x[:, 0, 3] += 0.3
        xd = dev(x)
        tol = RTOL64 if dt == np.float64 else 1e-6
        for mname in ("direct", "fft", "bda"):
            got = {}
            for sw in ("1", "0"):
                monkeypatch.setenv("MDT_WINOUT", sw)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    got[sw] = [host(o) for o in pkg.bulk_tail(xd, autocov_method=methods[mname], split_chains=s, maxlag=ml)]
                    got[sw].append(host(pkg.rhat(xd, kind="rank", split_chains=s)))
                    got[sw].append(host(pkg.ess(xd, kind="bulk", autocov_method=methods[mname], split_chains=s, maxlag=ml)))
            monkeypatch.delenv("MDT_WINOUT")
            for a, b in zip(got["1"], got["0"]):
                assert_close(a, b, 1e-12, f"windowed vs time order {D}x{C} s={s} {mname}")
            assert np.array_equal(got["1"][4], np.maximum(got["1"][2], got["1"][3]))     # rhat(rank) == max(bulk, tail), same bits
            assert np.array_equal(got["1"][5], got["1"][0])                               # ess(bulk) == fused bulk ESS
            want = co.bulk_tail(x, autocov=mname, split_chains=s, maxlag=ml)
            for a, b, nm in zip(got["1"], want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
                assert_close(a, b, tol, f"{nm} {D}x{C} s={s} {mname} {np.dtype(dt).name}")


def test_tail_ess_beyond_131072_draws(pkg, monkeypatch):
    """the indicator phase holds one SPLIT chain's bits at a time (round 2: draws / split_chains <= 262 144 instead of
    draws <= 131 072); both routes to the bits (from the windows of the z-series, from x itself) against the oracle; longer
    split chains take the indicator as a proxy series (no limit)"""
    from oracle import c_oracle as co
    rng = np.random.default_rng(99)
    x = np.concatenate([rng.standard_normal((200_000, 2, 1)), ar1(0.8, 200_000, 2, 1, rng)], axis=2)
    xd = dev(x)
    want = co.bulk_tail(x)
    for sw in (None, "1"):
        if sw:
            monkeypatch.setenv("MDT_NO_IBITS", sw)
        got = [host(o) for o in pkg.bulk_tail(xd)]
        for a, b, nm in zip(got, want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
            assert_close(a, b, RTOL64, f"{nm} 200000 draws, bits from {'x' if sw else 'the z-series'}")
        assert_close(host(pkg.ess(xd, kind=np.median)), co.ess_estimator(x, "median"), RTOL64, "ess median 200000 draws")
    monkeypatch.delenv("MDT_NO_IBITS")
    assert np.isfinite(host(pkg.ess(dev(rng.standard_normal((300_000, 1, 1))), kind="tail"))).all()   # niter = 150 000
    # beyond the bit vectors (draws / split_chains > 262 144): the indicators as proxy series through the tiled series phase
    y = np.concatenate([rng.standard_normal((300_000, 2, 1)), ar1(0.7, 300_000, 2, 1, rng)], axis=2)
    yd = dev(y)
    for m in ("direct", "bda"):
        want = co.bulk_tail(y, split_chains=1, autocov=m)
        got = [host(o) for o in pkg.bulk_tail(yd, split_chains=1, autocov_method=getattr(pkg, {"direct": "AutocovMethod", "bda": "BDAAutocovMethod"}[m])())]
        for a, b, nm in zip(got, want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
            assert_close(a, b, RTOL64, f"{nm} 300000 draws unsplit {m}")
    for est, kind in (("median", np.median), ("mad", "mad"), ("quantile", pkg.Quantile(0.3))):
        assert_close(host(pkg.ess(yd, kind=kind, split_chains=1)), co.ess_estimator(y, est, prob=0.3, split_chains=1), RTOL64, f"ess {est} 300000 draws unsplit")


def test_one_chunk_path_matches_the_general_path(pkg, methods, monkeypatch):
    """Round 4: slices of at most 4000 values (1000 draws x 4 chains, what a sampler emits) are ranked and carried through
    the bulk series inside ONE LDS chunk (csrc/mdt_short.inc): x read once, no histogram / partition passes, every split
    chain resident at once, the lag products of all chains in one sweep over the pooled series.  MDT_ONECHUNK=0 sends the
    same request down the general path: R-hat must be the same BITS (same fixed-point moments), the tied ranks the same
    (ESS of ties and point masses), ESS within 1e-9 (other summation order of the lag products); a NaN in a slice falls
    back by itself; the oracle agrees with both."""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(404)
    x = ar1(0.6, 1000, 4, 14, rng)
    x[:, :, 1] = rng.standard_normal((1000, 4))            # IID
    x[:, :, 2] = np.round(x[:, :, 2], 1)                   # heavy ties
    x[:, 2, 3] = 0.25                                      # a constant chain: W > 0, exact zeros in its deviations
    x[:, :, 4] = 1.5                                       # a constant parameter: NaN, NaN
    x[5, 1, 5] = np.inf; x[6, 1, 5] = -np.inf              # infinities rank first / last
    x[7, 0, 6] = 0.0; x[8, 0, 6] = -0.0                    # signed zeros tie
    x[:, :, 7] = np.exp(3.0 * x[:, :, 7])                  # heavy right tail: crowded bins
    x[rng.random((1000, 4)) < 0.4, 8] = 0.25               # a point mass holding 40 % of the slice
    x[:, :, 9] += 3.0 * rng.standard_normal((1, 4))        # chains that do not agree: every lag sweep
    x[:, :, 10] = ar1(0.97, 1000, 4, 1, rng)[:, :, 0]      # slowly mixing: several sweeps
    x[:, :, 11] = -x[:, :, 11]
    x[1::2, :, 12] = -x[0:-1:2, :, 12]                     # antithetic: ESS at the cap
    shapes = [("1000x4", x), ("996x4", x[:996]), ("500x2", x[:500, :2]), ("f32", x.astype(np.float32)), ("640x6", np.concatenate([x[:640], x[:640, :2]], axis=1))]

    def run_all(xd, s):
        out = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out += [("fused%d" % i, host(o)) for i, o in enumerate(pkg.bulk_tail(xd, split_chains=s))]
            for kind in ("bulk", "tail", "rank"):
                r = pkg.ess_rhat(xd, kind=kind, split_chains=s, autocov_method=methods["bda"] if kind == "bulk" else methods["direct"], maxlag=250 if kind != "rank" else 40)
                out += [(kind + " ess", host(r.ess)), (kind + " rhat", host(r.rhat))]
            out.append(("rhat only", host(pkg.rhat(xd, split_chains=s))))
            out.append(("fft ess", host(pkg.ess(xd, kind="bulk", split_chains=s, autocov_method=methods["fft"]))))
        return out

    monkeypatch.setenv("MDT_WAVE", "0")                # (round 5: by default the wave-per-parameter kernel takes these shapes)
    for what, xs in shapes:
        for s in (2, 1):
            if xs.shape[0] % s or xs.shape[1] * s > 8:
                continue
            xd = dev(xs)
            monkeypatch.setenv("MDT_ONECHUNK", "2")    # wherever it can run (the default leaves slices of <= 2560 values to the 256-thread geometry)
            a = run_all(xd, s)
            monkeypatch.setenv("MDT_ONECHUNK", "0")
            b = run_all(xd, s)
            monkeypatch.delenv("MDT_ONECHUNK")
            for (ka, va), (kb, vb) in zip(a, b):
                if "rhat" in ka or ka in ("fused2", "fused3"):
                    assert np.array_equal(va, vb, equal_nan=True), (what, s, ka)
                else:
                    assert_close(va, vb, 1e-9 if xs.dtype == np.float64 else 2e-6, f"{what} s={s} {ka}")
            if xs.dtype == np.float64:
                want = co.bulk_tail(xs, split_chains=s)
                for i in range(4):
                    assert_close(a[i][1], want[i], RTOL64, f"{what} s={s} oracle {i}")
    # a NaN anywhere in a slice: the general path serves that parameter (and the others stay on the one-chunk path)
    xn = x[:, :, :4].copy()
    xn[17, 2, 1] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkeypatch.setenv("MDT_ONECHUNK", "2")
        a = [host(pkg.rhat(dev(xn), kind=k)) for k in ("bulk", "tail", "rank")]
        monkeypatch.setenv("MDT_ONECHUNK", "0")
        b = [host(pkg.rhat(dev(xn), kind=k)) for k in ("bulk", "tail", "rank")]
        monkeypatch.delenv("MDT_ONECHUNK")
    for u, v in zip(a, b):
        assert np.array_equal(u, v, equal_nan=True)
    assert pkg.status() in (0, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["auto", "big"])
def test_constant_chains_in_moments_only_rounds(pkg, monkeypatch, cfg):
    """rhat(kind = rank / bulk / tail) and rhat_nested from the moments-only rounds (no scores written): a split chain whose
    scores are all the same bits has variance EXACTLY 0 -- the rounds take the or / and of the scores only where a tie group
    of at least niter members exists (round 4).  Constant chains, constant parameters, a constant chain whose value also
    occurs in the other chains, and a chain that is constant but for one draw, in both geometries; against the oracle and
    bit-equal to the scores-writing path (ess_rhat)."""
    import warnings
    from oracle import c_oracle as co
    if cfg != "auto":
        monkeypatch.setenv("MDT_CONFIG", cfg)
    rng = np.random.default_rng(404)
    D, C = 7000, 4
    x = rng.standard_normal((D, C, 7))
    x[:, 1, 0] = 0.25                                   # one constant chain (both its halves)
    x[: D // 2, 2, 1] = -0.5                            # one constant split chain
    x[:, :, 2] = 3.0                                    # constant parameter: NaN
    x[:, 0, 3] = x[5, 1, 3]                             # a constant chain whose value also sits in another chain
    x[:, 3, 4] = 1.0
    x[17, 3, 4] = 1.0000001                             # constant but for one draw: not constant
    x[:, :, 5] = np.round(x[:, :, 5], 1)                # heavy ties, no constant chain
    xd = dev(x)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kind in ("rank", "bulk", "tail"):
            got = host(pkg.rhat(xd, kind=kind))
            want = co.ess_rhat(x, kind=kind)[1]
            assert_close(got, want, RTOL64, f"rhat {kind} {cfg}")
            both = host(pkg.ess_rhat(xd, kind=kind)[1])
            assert np.array_equal(got, both, equal_nan=True), kind  # (the scores-writing path: same moments, same bits)
        ids = np.repeat(np.arange(1, 3), C // 2)
        inds = np.arange(C).reshape(2, C // 2).T        # column k: the chains of superchain k
        got = host(pkg.rhat_nested(xd, ids, kind="rank"))
        assert_close(got, co.rhat_nested(x, inds, kind="rank"), RTOL64, f"nested {cfg}")


def _wave_case_array(rng):
    """the data flavours of the one-chunk test, one per parameter (1000 draws x 4 chains)"""
    x = ar1(0.6, 1000, 4, 16, rng)
    x[:, :, 1] = rng.standard_normal((1000, 4))            # IID
    x[:, :, 2] = np.round(x[:, :, 2], 1)                   # heavy ties: point masses larger than a round -> the general kernel
    x[:, 2, 3] = 0.25                                      # a constant chain: W > 0, exact zeros in its deviations
    x[:, :, 4] = 1.5                                       # a constant parameter: NaN, NaN
    x[5, 1, 5] = np.inf; x[6, 1, 5] = -np.inf              # infinities -> the general kernel
    x[7, 0, 6] = 0.0; x[8, 0, 6] = -0.0                    # signed zeros tie
    x[:, :, 7] = np.exp(3.0 * x[:, :, 7])                  # heavy right tail: crowded bins
    x[rng.random((1000, 4)) < 0.4, 8] = 0.25               # a point mass holding 40 % of the slice
    x[:, :, 9] += 3.0 * rng.standard_normal((1, 4))        # chains that do not agree: every lag sweep
    x[:, :, 10] = ar1(0.97, 1000, 4, 1, rng)[:, :, 0]      # slowly mixing: several sweeps
    x[:, :, 11] = -x[:, :, 11]
    x[1::2, :, 12] = -x[0:-1:2, :, 12]                     # antithetic: ESS at the cap
    x[:, :, 13] = np.round(x[:, :, 13], 3)                 # a few ties: half-integer ranks, equals among the folds
    x[::2, :, 14] = -x[1::2, :, 14]                        # symmetric pairs: every fold has an equal
    x[:, :, 15] = np.floor(4.0 * x[:, :, 15])              # a handful of distinct values: every class a point mass -> the general kernel
    return x


def test_wave_path_matches_the_general_path(pkg, methods, monkeypatch):
    """Round 5: slices of 256 ... 8192 values run ONE PARAMETER PER WAVE (csrc/mdt_wave.hip): no workgroup barrier, the slice
    class-sorted through a per-wave scratch, ranked round by round in ~11 KB of LDS, fold merge, z-series staged per split
    chain, lag products on the matrix pipe, popcount indicators.  MDT_WAVE=0 sends the same request down the
    workgroup-per-parameter kernels: R-hat must be the same BITS (same fixed-point moments, same expressions), ESS within
    1e-9 (other summation order of the lag products); parameters the path cannot serve (NaN, infinities, point masses, many
    tied folds) come back through the general kernel inside the same request; the oracle agrees with both."""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(505)
    x = _wave_case_array(rng)
    shapes = [("1000x4", x), ("997x4", x[:997]), ("500x4", x[:500]), ("250x4", x[:250]), ("64x4", x[:64]), ("f32", x.astype(np.float32)),
              ("640x6", np.concatenate([x[:640], x[:640, :2]], axis=1)), ("250x8", np.concatenate([x[:250], x[250:500]], axis=1)),
              ("1000x1", x[:, :1]), ("1280x3", np.concatenate([x, x[:280]], axis=0)[:, :3]),
              ("1000x8", np.concatenate([x, x[::-1]], axis=1)), ("1280x6", np.concatenate([np.concatenate([x, x[:280]], axis=0), np.concatenate([x[::-1], x[:280]], axis=0)[:, :2]], axis=1))]

    def run_all(xd, s):
        out = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out += [("fused%d" % i, host(o)) for i, o in enumerate(pkg.bulk_tail(xd, split_chains=s))]
            for kind in ("bulk", "tail", "rank"):
                r = pkg.ess_rhat(xd, kind=kind, split_chains=s, autocov_method=methods["direct"], maxlag=250 if kind != "rank" else 40)
                out += [(kind + " ess", host(r.ess)), (kind + " rhat", host(r.rhat))]
            out.append(("rhat only", host(pkg.rhat(xd, split_chains=s))))
            out.append(("rhat tail", host(pkg.rhat(xd, kind="tail", split_chains=s))))
            out.append(("ess tail rel", host(pkg.ess(xd, kind="tail", split_chains=s, relative=True))))
            out.append(("fft ess", host(pkg.ess(xd, kind="bulk", split_chains=s, autocov_method=methods["fft"]))))
            out.append(("fft ess tail", host(pkg.ess(xd, kind="tail", split_chains=s, autocov_method=methods["fft"], maxlag=7))))
        return out

    for what, xs in shapes:
        for s in (2, 1, 3):
            D, C = xs.shape[:2]
            if C * s > 16 or D // s > 1024 or D // s < 8 or D * C < 256 or D * C > 8192:
                continue
            xd = dev(xs)
            monkeypatch.setenv("MDT_WAVE", "2")
            a = run_all(xd, s)
            monkeypatch.setenv("MDT_WAVE", "0")
            b = run_all(xd, s)
            monkeypatch.delenv("MDT_WAVE")
            for (ka, va), (kb, vb) in zip(a, b):
                if "rhat" in ka or ka in ("fused2", "fused3"):
                    assert np.array_equal(va, vb, equal_nan=True), (what, s, ka, va, vb)
                else:
                    assert_close(va, vb, 1e-9 if xs.dtype == np.float64 else 2e-6, f"{what} s={s} {ka}")
            if xs.dtype == np.float64:
                want = co.bulk_tail(xs, split_chains=s)
                for i in range(4):
                    assert_close(a[i][1], want[i], RTOL64, f"{what} s={s} oracle {i}")
    # a NaN anywhere in a slice: the general kernel serves that parameter inside the same request (a NaN ranks last, isless;
    # the tail ESS of such a slice is an ArgumentError in the reference, Statistics.quantile)
    xn = x[:, :, :4].copy()
    xn[17, 2, 1] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkeypatch.setenv("MDT_WAVE", "2")
        a = [host(pkg.rhat(dev(xn), kind=k)) for k in ("bulk", "tail", "rank")] + [host(pkg.ess(dev(xn), kind="bulk"))]
        monkeypatch.setenv("MDT_WAVE", "0")
        b = [host(pkg.rhat(dev(xn), kind=k)) for k in ("bulk", "tail", "rank")] + [host(pkg.ess(dev(xn), kind="bulk"))]
        monkeypatch.delenv("MDT_WAVE")
    for u, v in zip(a[:3], b[:3]):
        assert np.array_equal(u, v, equal_nan=True)
    assert_close(a[3], b[3], 1e-9, "ess bulk beside a NaN slice")
    assert pkg.status() in (0, 2)


def test_wave_path_tail_ess_alone_ranks_only_the_rounds_of_its_quantiles(pkg, methods, monkeypatch):
    """Round 6: ess(kind = tail) alone (src/ess_rhat.jl:301-311) needs nothing of the ranking but two pairs of order statistics: the
    wave-per-parameter kernel then stores and ranks only the rounds that hold them and takes the indicators from x.  MDT_WAVE_OSONLY=0
    ranks everything and takes the indicators from the ranks: the same bits (the indicator sums are exact integers), on continuous
    data, heavy ties, Float32, every split; the general kernels (MDT_WAVE=0) and the oracle agree.  ess(kind = median | quantile(p)) take the
    same route with their one pair, ess(kind = mad) takes it twice: the second time the passes read |x - median(x)| where they read x."""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(707)
    x = _wave_case_array(rng)
    shapes = [("1000x4", x), ("997x4", x[:997]), ("250x4", x[:250]), ("f32", x.astype(np.float32)), ("1000x8", np.concatenate([x, x[::-1]], axis=1)),
              ("ties", np.round(x, 1))]
    for what, xs in shapes:
        for s in (2, 1, 3):
            D, C = xs.shape[:2]
            if C * s > 16 or D // s > 1024 or D // s < 8 or D * C < 256 or D * C > 8192:
                continue
            xd = dev(xs)
            outs = {}
            for tag, env in (("only", {"MDT_WAVE": "2"}), ("full", {"MDT_WAVE": "2", "MDT_WAVE_OSONLY": "0"}), ("wg", {"MDT_WAVE": "0"})):
                for k in ("MDT_WAVE", "MDT_WAVE_OSONLY"):
                    monkeypatch.delenv(k, raising=False)
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    outs[tag] = [host(pkg.ess(xd, kind="tail", split_chains=s)), host(pkg.ess(xd, kind="tail", split_chains=s, tail_prob=0.3, maxlag=20, relative=True)),
                                 host(pkg.ess(xd, kind="tail", split_chains=s, autocov_method=methods["fft"])),
                                 host(pkg.ess(xd, kind=np.median, split_chains=s)), host(pkg.ess(xd, kind=pkg.Quantile(0.37), split_chains=s, maxlag=60)),
                                 host(pkg.ess(xd, kind="mad", split_chains=s))]
            for k in ("MDT_WAVE", "MDT_WAVE_OSONLY"):
                monkeypatch.delenv(k, raising=False)
            for a, b, c in zip(outs["only"], outs["full"], outs["wg"]):
                assert np.array_equal(a, b, equal_nan=True), (what, s)
                assert_close(a, c, 1e-9 if xs.dtype == np.float64 else 2e-6, f"{what} s={s} against the general kernels")
            if xs.dtype == np.float64:
                e, _ = co.ess_rhat(xs, kind="tail", split_chains=s)
                assert_close(outs["only"][0], e, RTOL64, f"{what} s={s} oracle")
                assert_close(outs["only"][3], co.ess_estimator(xs, "median", split_chains=s), RTOL64, f"{what} s={s} oracle median")
                assert_close(outs["only"][5], co.ess_estimator(xs, "mad", split_chains=s), RTOL64, f"{what} s={s} oracle mad")


def test_wave_path_raw_draws_match_the_general_path(pkg, methods, monkeypatch):
    """Round 6: raw draws -- ess / rhat / ess_rhat(kind = basic) and mcse(kind = mean) (src/ess_rhat.jl:438-603, src/mcse.jl:45-51) -- on
    short slices through the series phase of the wave-per-parameter kernel (no ranking).  Against the workgroup-per-parameter
    kernels (MDT_WAVE=0): R-hat the same BITS (the moments are plain double sums taken in the same order: at most 1024 draws per
    split chain, one per virtual thread), ESS and mcse within 1e-12 (other summation order of the lag products); a NaN or an
    infinity among the draws comes back through the general kernel inside the same request; the oracle agrees."""
    import warnings
    import torch
    from oracle import c_oracle as co
    rng = np.random.default_rng(606)
    x = _wave_case_array(rng)
    x[3, 1, 5] = np.inf      # (parameters the path hands back)
    x[40, 0, 7] = np.nan
    shapes = [("1000x4", x), ("997x4", x[:997]), ("500x4", x[:500]), ("64x4", x[:64]), ("f32", x.astype(np.float32)),
              ("1000x8", np.concatenate([x, x[::-1]], axis=1)), ("1000x1", x[:, :1]), ("2000x2", np.concatenate([x[:, :2], x[:, 2:]], axis=0))]

    def run_all(xd, s):
        out = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = pkg.ess_rhat(xd, kind="basic", split_chains=s)
            out += [("ess", host(r.ess)), ("rhat", host(r.rhat))]
            out.append(("rhat only", host(pkg.rhat(xd, kind="basic", split_chains=s))))
            out.append(("ess only rel", host(pkg.ess(xd, kind="basic", split_chains=s, relative=True, maxlag=30))))
            out.append(("ess fft", host(pkg.ess(xd, kind="basic", split_chains=s, autocov_method=methods["fft"]))))
            out.append(("mcse", host(pkg.mcse(xd, split_chains=s))))
            out.append(("std ess", host(pkg.ess(xd, kind=np.std, split_chains=s))))        # the proxy (x - mean)^2 (src/ess_rhat.jl:640-642)
            out.append(("std mcse", host(pkg.mcse(xd, kind=np.std, split_chains=s))))      # src/mcse.jl:52-65
        return out

    for what, xs in shapes:
        for s in (2, 1, 3):
            D, C = xs.shape[:2]
            if C * s > 16 or D // s > 1024 or D // s < 8 or D * C < 256 or D * C > 8192:
                continue
            xd = dev(xs)
            monkeypatch.setenv("MDT_WAVE", "2")
            a = run_all(xd, s)
            monkeypatch.setenv("MDT_WAVE", "0")
            b = run_all(xd, s)
            monkeypatch.delenv("MDT_WAVE")
            for (ka, va), (kb, vb) in zip(a, b):
                if "rhat" in ka:
                    assert np.array_equal(va, vb, equal_nan=True), (what, s, ka, va, vb)
                elif ka.startswith("std"):  # (the pooled mean of the proxy is a thread-strided sum on the workgroup kernels: another order here)
                    assert_close(va, vb, 1e-9 if xs.dtype == np.float64 else 5e-5, f"{what} s={s} {ka}")
                else:
                    assert_close(va, vb, 1e-12 if xs.dtype == np.float64 else 2e-6, f"{what} s={s} {ka}")
            if xs.dtype == np.float64:
                assert_close(a[6][1], co.ess_estimator(xs, "std", split_chains=s), RTOL64, f"{what} s={s} oracle ess std")
                assert_close(a[7][1], co.mcse_std(xs, split_chains=s), RTOL64, f"{what} s={s} oracle mcse std")
                e, r = co.ess_rhat(xs, kind="basic", split_chains=s)
                assert_close(a[0][1], e, RTOL64, f"{what} s={s} oracle ess")
                assert_close(a[1][1], r, RTOL64, f"{what} s={s} oracle rhat")
    # MCMCChains storage (iterations, parameters, chains), read in place
    xs = x[:, :, :6]
    D, C, P = xs.shape
    xm = torch.from_numpy(np.ascontiguousarray(np.transpose(xs, (1, 2, 0)))).cuda()
    view = xm.permute(2, 0, 1)
    monkeypatch.setenv("MDT_WAVE", "2")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = [host(pkg.rhat(view, kind="basic")), host(pkg.mcse(view))]
        b = [host(pkg.rhat(dev(xs), kind="basic")), host(pkg.mcse(dev(xs)))]
    monkeypatch.delenv("MDT_WAVE")
    for u, v in zip(a, b):
        assert np.array_equal(u, v, equal_nan=True)


@pytest.mark.parametrize("name", ["cfg1_randn_1000x4x10", "ties_200x4x6", "odd_1001x3x3", "ar1_500x4x10", "onechain_400x1x3"])
def test_wave_path_on_the_golden_vectors(pkg, methods, monkeypatch, name):
    """the committed golden vectors through the wave-per-parameter kernel, forced wherever it can run (every kind x method x
    split x maxlag of the fixture; what it cannot serve -- raw draws, BDA, tiny slices -- takes the general kernels as always)"""
    import warnings
    monkeypatch.setenv("MDT_WAVE", "2")
    g = load_golden(name)
    xd = dev(g["x"])
    for ke, kr, kind, method, s, ml in _keys(g):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            S, R = pkg.ess_rhat(xd, kind=kind, autocov_method=methods[method], split_chains=s, maxlag=ml)
        assert_close(host(S), g[ke], RTOL64, f"{name} {ke}")
        assert_close(host(R), g[kr], RTOL64, f"{name} {kr}")


def test_wave_path_strided_and_host_inputs(pkg, monkeypatch):
    """MCMCChains storage read in place (chain_stride / param_stride) and host arrays through the wave-per-parameter kernel:
    the same bits as the dense device call on the same kernel."""
    import torch
    monkeypatch.setenv("MDT_WAVE", "2")
    rng = np.random.default_rng(19)
    for dt in (np.float64, np.float32):
        for shape in ((300, 4, 7), (1000, 4, 9), (640, 6, 5)):
            x = (ar1(0.6, *shape, rng) + 0.1 * rng.standard_normal(shape)).astype(dt)
            store = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 2, 0)))).cuda()  # (chains, params, draws) C-order
            view = store.permute(2, 0, 1)
            assert view.stride() == (1, shape[0] * shape[2], shape[0])
            dense = pkg.bulk_tail(dev(x))
            for u, w in zip(pkg.bulk_tail(view), dense):
                assert np.array_equal(host(u), host(w), equal_nan=True)
            for u, w in zip(pkg.bulk_tail(x), dense):                                         # host array in, host arrays out
                assert np.array_equal(np.asarray(u), host(w), equal_nan=True)
            for kind in ("bulk", "tail", "rank"):
                a, b = pkg.ess_rhat(view, kind=kind), pkg.ess_rhat(dev(x), kind=kind)
                assert np.array_equal(host(a.ess), host(b.ess)) and np.array_equal(host(a.rhat), host(b.rhat))


def test_switches_need_the_test_hooks_gate(pkg):
    """The library reads MDT_WAVE, MDT_CONFIG, MDT_RANK_VARIANT ... only when MDT_TEST_HOOKS=1 was set when it was first used
    (tests/conftest.py does that): a child process WITHOUT the gate must give the default path's bits whatever the switches say."""
    import subprocess, sys, os, json
    code = r"""
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
rng = np.random.default_rng(7)
x = rng.standard_normal((8, 4, 1000)).cumsum(axis=2) * 0.05 + rng.standard_normal((8, 4, 1000))
xd = torch.from_numpy(x).cuda().permute(2, 1, 0)
print(json.dumps([o.cpu().numpy().tolist() for o in pkg.bulk_tail(xd)]))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env_extra in ({}, {"MDT_WAVE": "0", "MDT_CONFIG": "big", "MDT_RANK_VARIANT": "1", "MDT_ONECHUNK": "0", "MDT_FOLD_MERGE": "0"}):
        env = {k: v for k, v in os.environ.items() if not k.startswith("MDT_")}
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]


def test_wave_path_is_taken_from_the_parameter_count_at_which_it_pays(pkg, monkeypatch):
    """The host takes the wave-per-parameter kernel only when the request has enough parameters to fill it (params >=
    700 + 4/3 draws x chains; below, a workgroup per parameter has the lower latency).  The two families agree in R-hat bit for
    bit and in ESS to ~1e-15 only, so the bits tell which one ran: 300 x 4 slices (threshold 2300 parameters) -- 200 parameters
    must be the bits of MDT_WAVE=0, 3000 those of MDT_WAVE=2 (and not those of MDT_WAVE=0: the ESS of some parameter differs in
    its last bits)."""
    rng = np.random.default_rng(2300)
    x = ar1(0.5, 300, 4, 3000, rng)
    xd = dev(x)

    def run(P, wave):
        if wave is None:
            monkeypatch.delenv("MDT_WAVE", raising=False)
        else:
            monkeypatch.setenv("MDT_WAVE", wave)
        return [o.cpu().numpy() for o in pkg.bulk_tail(xd[:, :, :P])]  # (the first P parameters: a dense block of the same array)

    few_default, few_never, few_forced = run(200, None), run(200, "0"), run(200, "2")
    many_default, many_never, many_forced = run(3000, None), run(3000, "0"), run(3000, "2")
    for a, b in zip(few_default, few_never):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(many_default, many_forced):
        np.testing.assert_array_equal(a, b)
    # R-hat: the same bits whichever kernel; ESS: the families sum the lag products in another order
    for i in (2, 3):
        np.testing.assert_array_equal(many_forced[i], many_never[i])
        np.testing.assert_array_equal(few_forced[i], few_never[i])
    assert any(not np.array_equal(many_forced[i], many_never[i]) for i in (0, 1)), "the forced paths gave identical ESS bits: the test cannot tell them apart"
    for i in (0, 1):
        np.testing.assert_allclose(many_forced[i], many_never[i], rtol=1e-12)


def test_reserve_changes_no_bits_and_validates_like_a_request(pkg):
    """mdt_ctx_reserve (C-ABI 0.3.1; include/mdt_hip.h): announcing the shape moves the allocations and the score table out of the
    first request -- and nothing else: a request on a fresh stream (= a fresh context) with and without the announcement gives
    the same bits, for a shape of each kernel family, Float64 and Float32; a shape nobody can serve is refused as a request would
    refuse it."""
    import torch
    rng = np.random.default_rng(31)
    for (D, C, P), dt in (((1000, 4, 40), np.float64), ((10000, 4, 6), np.float64), ((500, 8, 24), np.float32)):
        x = ar1(0.4, D, C, P, rng).astype(dt)
        outs = []
        for announce in (False, True):
            st = torch.cuda.Stream()  # the context cache is keyed by (device, stream): nothing allocated yet on this one
            with torch.cuda.stream(st):
                xd = dev(x)
                if announce:
                    pkg.reserve(D, C, P, dtype=dt)
                    pkg.reserve(D, C, P, dtype=dt)  # (idempotent)
                outs.append([o.cpu().numpy() for o in pkg.bulk_tail(xd)])
            st.synchronize()
        for a, b in zip(*outs):
            np.testing.assert_array_equal(a, b)
    with pytest.raises(pkg.ArgumentError):
        pkg.reserve(100, 4, 10, dtype=np.int32)
    with pytest.raises(pkg.DimensionMismatch):
        pkg.reserve(0, 4, 10)
    with pytest.raises(Exception):
        pkg.reserve(1 << 20, 1 << 11, 1)  # draws x chains beyond the ranking's index range
    with pytest.raises(Exception):
        pkg.reserve(1 << 40, 1 << 40, 1)  # ... and a product that would overflow int64 (checked before it is formed)


def test_reserve_like_resolves_the_context_of_the_request(pkg):
    """reserve(like=array) warms exactly the context the request on that array resolves to (round-5 advisor: without it a host request
    -- context (0, 0) -- or a request under another stream found a cold context), announces its layout (MCMCChains storage: the
    per-workgroup gather buffer) and its location (host arrays: the staging buffers), and changes no bits."""
    import torch
    rng = np.random.default_rng(32)
    D, C, P = 600, 4, 30
    x = ar1(0.3, D, C, P, rng)
    # a host request: the context of (device 0, stream 0), whatever the current stream is
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        pkg.reserve(0, 0, 0, like=x)
        got_host = [np.asarray(o) for o in pkg.bulk_tail(x)]
    assert pkg.api._ctx_key(pkg.api._describe(x)) == (0, 0)
    # MCMCChains storage (iterations, parameters, chains) on the device, read in place
    xm = torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 2, 0)))).cuda()  # memory: chains outermost, draws fastest
    view = xm.permute(2, 0, 1)                                                      # (draws, chains, params), strided
    d = pkg.api._describe(view)
    assert (d.D, d.C, d.P) == (D, C, P) and d.cs == D * P and d.ps == D
    st2 = torch.cuda.Stream()
    with torch.cuda.stream(st2):
        pkg.reserve(0, 0, 0, like=view)
        got_str = [o.cpu().numpy() for o in pkg.bulk_tail(view)]
    st2.synchronize()
    ref = [o.cpu().numpy() for o in pkg.bulk_tail(dev(x))]
    for a, b, c in zip(got_host, got_str, ref):
        np.testing.assert_array_equal(a, c)
        np.testing.assert_array_equal(b, c)


def test_chain_lengths_between_the_lds_resident_and_the_windowed_series(pkg, methods):
    """Split chains of SZ - 80 ... SZ - 41 draws (SZ = 6400 doubles of chain buffer in the 512-thread geometry) were neither
    `zfast` (chain resident in LDS) nor `long_chain` (windowed series) until round 4 -- two margins, 80 and 40 -- and took the
    slow double pass; both predicates now derive from one constant.  6340 draws per split chain: against the oracle, and
    rhat(x) == ess_rhat(x).rhat bit for bit as everywhere."""
    import warnings
    from oracle import c_oracle as co
    rng = np.random.default_rng(77)
    for niter in (6340, 6321, 6359):
        x = ar1(0.5, 2 * niter, 2, 5, rng)
        x[:, :, 1] = rng.standard_normal((2 * niter, 2))
        xd = dev(x)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = [host(o) for o in pkg.bulk_tail(xd)]
            want = co.bulk_tail(x)
            for a, b, nm in zip(got, want, ("ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail")):
                assert_close(a, b, RTOL64, f"{nm} niter {niter}")
            assert np.array_equal(host(pkg.rhat(xd, kind="bulk")), got[2])
            assert np.array_equal(host(pkg.rhat(xd, kind="tail")), got[3])
            assert_close(host(pkg.ess(xd, kind="bulk", autocov_method=methods["fft"])), co.ess_rhat(x, kind="bulk", autocov="fft")[0], RTOL64, f"fft niter {niter}")
