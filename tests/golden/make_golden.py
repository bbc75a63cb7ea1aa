"""Mint the golden vectors in tests/golden/*.npz.

Run from the repo root:  python tests/golden/make_golden.py

The reference (Julia) cannot run in this image and its own tests hold no numeric vectors for
this path, so the expected values here come from TWO independent restatements
that must agree to <= 1e-10 relative before anything is written: oracle/mdt_oracle.c (C) and
oracle/numpy_oracle.py (scipy rankdata / ndtri / numpy.fft).  The stored value is the C one.
Inputs are stored too, so the fixtures are self-contained data (no generator needed at test
time).  Float32 cases store the "f64-internal" comparator AND the Float32-faithful variant.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co  # noqa: E402
from oracle import numpy_oracle as no  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KINDS = ["basic", "bulk", "tail", "rank"]
METHODS = ["direct", "fft", "bda"]


def ar1(phi, D, C, P, rng):
    e = rng.standard_normal((D, C, P)) * np.sqrt(1 - phi ** 2)
    x = np.empty_like(e)
    x[0] = rng.standard_normal((C, P))
    for t in range(1, D):
        x[t] = phi * x[t - 1] + e[t]
    return x


def agree(a, b, tol=1e-10, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    m = ~np.isnan(a)
    if m.any():
        err = np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300))
        assert err <= tol, (what, err)


def full_case(x, splits=(2,), maxlags=(250,), methods=METHODS, kinds=KINDS, check=True):
    """all (kind, method, split, maxlag) results for x (D, C, P) float64"""
    out = {"x": np.asfortranarray(x)}
    for kind in kinds:
        for m in methods:
            for s in splits:
                for ml in maxlags:
                    key = f"{kind}|{m}|{s}|{ml}"
                    e, r = co.ess_rhat(x, kind=kind, autocov=m, split_chains=s, maxlag=ml)
                    if check:
                        e2, r2 = no.ess_rhat(x, kind=kind, method=m, s=s, maxlag=ml)
                        agree(e, e2, what=key + " ess")
                        agree(r, r2, what=key + " rhat")
                    out["ess|" + key] = e
                    out["rhat|" + key] = r
    return out


def main():
    rng = np.random.default_rng(20261003)
    cases = {}

    # config 1 of BASELINE.json: randn(1000, 4, 10), every kind and method
    x = rng.standard_normal((1000, 4, 10))
    c = full_case(x, splits=(1, 2))
    eb, et, rb, rt = co.bulk_tail(x, autocov="fft")
    c["bulk_tail_fft"] = np.stack([eb, et, rb, rt])
    # stage outputs of parameter 0
    xp = x[:, :, 0]
    c["stage_rank"] = co.tiedrank(xp)
    c["stage_z"] = co.rank_normalize(xp)
    agree(c["stage_z"], no.rank_normalize(xp), what="z")
    c["stage_fold"] = co.fold_around_median(xp)
    c["stage_q05"] = np.array([co.quantile(xp, 0.05), co.quantile(xp, 0.95)])
    agree(c["stage_q05"], [no.quantile7(xp.ravel(order="F"), 0.05), no.quantile7(xp.ravel(order="F"), 0.95)])
    for m in METHODS:
        c["stage_acov_" + m] = co.mean_autocov(xp, 2, m, nlags=251)
    cases["cfg1_randn_1000x4x10"] = c

    # AR(1) chains over the mixing range, with chain offsets on some params
    xs = [ar1(phi, 500, 4, 2, rng) for phi in (-0.9, -0.3, 0.0, 0.5, 0.9)]
    x = np.concatenate(xs, axis=2)
    x[:, 1, 1::2] += 0.5
    cases["ar1_500x4x10"] = full_case(x, splits=(1, 2, 3), maxlags=(250, 10))

    # heavy ties: integers 1..10, and a zero-inflated continuous param
    x = rng.integers(1, 11, size=(200, 4, 4)).astype(np.float64)
    z = rng.standard_normal((200, 4, 2))
    z[rng.random((200, 4, 2)) < 0.4] = 0.0
    x = np.concatenate([x, z], axis=2)
    c = full_case(x)
    c["stage_rank"] = co.tiedrank(x[:, :, 0])
    cases["ties_200x4x6"] = c

    # constant input -> NaN everywhere (test/ess_rhat.jl:242-257)
    cases["constant_100x3x2"] = full_case(np.ones((100, 3, 2)))

    # odd draws, split 2 and 3 (row-drop rule, test/utils.jl:47-56); single chain
    cases["odd_1001x3x3"] = full_case(rng.standard_normal((1001, 3, 3)), splits=(1, 2, 3))
    cases["onechain_400x1x3"] = full_case(ar1(0.5, 400, 1, 3, rng), splits=(1, 2))

    # niter guard and maxlag edge cases
    for D in (4, 5, 8, 9, 10, 11, 12):
        cases[f"short_{D}x3x2"] = full_case(rng.standard_normal((D, 3, 2)), splits=(1, 2), methods=["direct"])
    cases["maxlag_edges_60x4x3"] = full_case(ar1(0.7, 60, 4, 3, rng), maxlags=(1, 2, 3, 10, 250))

    # signed zeros / infinities (NaN-free): +-0.0 tie, Inf ranks last
    x = rng.standard_normal((50, 2, 3))
    x[3, 0, 0] = 0.0
    x[7, 1, 0] = -0.0
    x[11, 0, 1] = np.inf
    x[13, 1, 1] = -np.inf
    c = full_case(x, methods=["direct"], check=False)  # scipy/np.quantile differ on inf arithmetic
    c["stage_rank0"] = co.tiedrank(x[:, :, 0])
    c["stage_rank1"] = co.tiedrank(x[:, :, 1])
    cases["special_50x2x3"] = c

    # antithetic AR(1): max ESS == ntotal*log10(ntotal) exactly (test/ess_rhat.jl:314-327)
    cases["antithetic_100x4x20"] = full_case(ar1(-0.9, 100, 4, 20, rng), kinds=["basic"], methods=["direct"])

    # nested R-hat: 16 chains, 4 superchains of 4 (test/rhat_nested.jl recipe)
    x = rng.standard_normal((200, 16, 4))
    x += np.repeat(rng.standard_normal((1, 4, 4)) * 0.3, 4, axis=1)
    inds = np.arange(16, dtype=np.int32).reshape(4, 4, order="F")  # column k = chains 4k..4k+3
    c = {"x": np.asfortranarray(x), "inds": inds}
    for kind in KINDS:
        for s in (1, 2):
            r = co.rhat_nested(x, inds, kind=kind, split_chains=s)
            agree(r, no.rhat_nested(x, inds, kind=kind, s=s), what="nested " + kind)
            c[f"nested|{kind}|{s}"] = r
    m = co.mcse_mean(x)
    agree(m, no.mcse_mean(x), what="mcse")
    c["mcse_mean"] = m
    cases["nested_200x16x4"] = c

    # Float32 input: f64-internal comparator and the Float32-faithful variant
    x32 = ar1(0.3, 300, 4, 6, rng).astype(np.float32)
    c = {"x": np.asfortranarray(x32)}
    for kind in KINDS:
        for m in ("direct", "fft"):
            e, r = co.ess_rhat(x32, kind=kind, autocov=m)
            ef, rf = co.ess_rhat(x32, kind=kind, autocov=m, faithful=True)
            c[f"ess|{kind}|{m}|2|250"] = e
            c[f"rhat|{kind}|{m}|2|250"] = r
            c[f"essF|{kind}|{m}|2|250"] = ef
            c[f"rhatF|{kind}|{m}|2|250"] = rf
    cases["f32_300x4x6"] = c

    for name, c in cases.items():
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **c)
    print("wrote", len(cases), "fixtures to", OUT)


ESTIMATORS = [("median", 0.5), ("std", 0.5), ("mad", 0.5), ("quantile", 0.25), ("quantile", 0.9)]


def estimators():
    """ess(x; kind = estimator) and mcse(x; kind = std) on the INPUTS of fixtures that
    already exist (so those are not regenerated): est_<name>.npz, keys est|<estimator>|<p>|<method>|<s>."""
    names = ["cfg1_randn_1000x4x10", "ar1_500x4x10", "ties_200x4x6", "odd_1001x3x3", "onechain_400x1x3",
             "constant_100x3x2", "short_5x3x2", "special_50x2x3", "f32_300x4x6"]
    n = 0
    for name in names:
        x = np.load(os.path.join(OUT, name + ".npz"))["x"]
        f32 = x.dtype == np.float32
        c = {}
        for est, p in ESTIMATORS:
            for m in (("direct",) if f32 else METHODS):
                for s in ((2,) if f32 else (1, 2, 3)):
                    key = f"est|{est}|{p}|{m}|{s}"
                    try:
                        e = co.ess_estimator(x, est, prob=p, autocov=m, split_chains=s)
                    except co.OracleError:  # quantile of data with NaN throws (Statistics.quantile)
                        c["err|" + key] = np.array([1])
                        continue
                    if not f32:
                        agree(e, no.ess_estimator(x, est, prob=p, method=m, s=s), what=name + " " + key)
                    c[key] = e
        if not np.isnan(x).any():
            c["mcse_std"] = co.mcse_std(x)
            if not f32:
                agree(c["mcse_std"], no.mcse_std(x), what=name + " mcse_std")
            for est, p in (("median", 0.5), ("quantile", 0.25), ("quantile", 0.9)):
                c[f"mcseq|{est}|{p}"] = co.mcse_quantile(x, est, prob=p)
                if not f32:
                    agree(c[f"mcseq|{est}|{p}"], no.mcse_quantile(x, est, prob=p), what=name + f" mcse {est} {p}")
        np.savez_compressed(os.path.join(OUT, "est_" + name + ".npz"), **c)
        n += 1
    print("wrote", n, "estimator fixtures to", OUT)


# the energies of the reference's own known-answer tests (test/bfmi.jl:2-41): data + the expected values there
BFMI_REF_ENERGY = [42, 44, 45, 46, 42, 43, 36, 36, 31, 36, 36, 32, 36, 31, 31, 29, 29, 30, 25, 26, 29, 29, 27, 30, 31, 29]


def diag():
    """bfmi and gelmandiag on the inputs of existing fixtures: diag_<name>.npz with
    gelman|<alpha>|psrf, gelman|<alpha>|psrfci and bfmi (every (chain, param) column read as an energy series),
    plus diag_bfmi_reference.npz = the reference's two known-answer cases."""
    names = ["cfg1_randn_1000x4x10", "ar1_500x4x10", "ties_200x4x6", "odd_1001x3x3", "f32_300x4x6", "nested_200x16x4"]
    for name in names:
        x = np.load(os.path.join(OUT, name + ".npz"))["x"]
        f32 = x.dtype == np.float32
        c = {}
        for alpha in (0.05, 0.2):
            psrf, ci = co.gelmandiag(x, alpha=alpha)
            if not f32:
                a, b = no.gelmandiag(x, alpha=alpha)
                agree(psrf, a, what=name + " psrf")
                agree(ci, b, what=name + " psrfci")
            c[f"gelman|{alpha}|psrf"], c[f"gelman|{alpha}|psrfci"] = psrf, ci
        e = x.reshape(x.shape[0], -1, order="F")
        c["bfmi"] = co.bfmi(e)
        if not f32:
            agree(c["bfmi"], no.bfmi(e), what=name + " bfmi")
        np.savez_compressed(os.path.join(OUT, "diag_" + name + ".npz"), **c)
    e = np.array(BFMI_REF_ENERGY, dtype=np.float64)
    ref = {"energy_a": np.array([1.0, 2.0, 3.0, 4.0]), "expect_a": np.array(0.6),
           "energy_b": e, "expect_b": np.array(0.2406937229)}
    assert abs(co.bfmi(ref["energy_a"]) - 0.6) < 1e-12 and abs(co.bfmi(e) - 0.2406937229) < 1e-9
    np.savez_compressed(os.path.join(OUT, "diag_bfmi_reference.npz"), **ref)
    print("wrote", len(names) + 1, "diag fixtures to", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "estimators":
        estimators()
    elif len(sys.argv) > 1 and sys.argv[1] == "diag":
        diag()
    else:
        main()
        estimators()
        diag()
