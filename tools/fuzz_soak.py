#!/usr/bin/env python3
"""Soak test (needs a GPU; not part of the pytest suite): random shapes up to ~150 000 values per slice, splits,
kinds, autocov methods, data flavours and geometries against the C oracle.  python tools/fuzz_soak.py [seconds] [seed] [long|many|order|wave]
(long: chains of 50 000 - 131 072 draws, maxlag up to 1000; many: 100 - 1500 chains, more split chains than the LDS moment tables hold;
order: 600 - 4000 short parameters per request, each request issued twice -- the second launch hands its tickets out by
the costs the first one measured; wave: slices of 256 ... 8192 values in up to 16 split chains of up to 1024 draws, 1 - 300 parameters per request, the
geometry never forced: the wave-per-parameter kernel of round 5 and, for what it hands back, the general kernel behind it.
Every mode forces the tile-major partition (MDT_PA=2) on every other request.)"""
import os, sys, time, warnings
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as g
from conftest import ar1, assert_close
from oracle import c_oracle as co

pkg = g.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LONG = len(sys.argv) > 3 and sys.argv[3] == "long"
MANY = len(sys.argv) > 3 and sys.argv[3] == "many"
ORDER = len(sys.argv) > 3 and sys.argv[3] == "order"
WAVE = len(sys.argv) > 3 and sys.argv[3] == "wave"
methods = {"direct": pkg.AutocovMethod(), "fft": pkg.FFTAutocovMethod(), "bda": pkg.BDAAutocovMethod()}


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 1, 0)))).cuda().permute(2, 1, 0)


def host(t):
    return t.cpu().numpy()


# Geyer's cut-off (src/ess_rhat.jl:555-594) is discontinuous: a pair sum rho_k + rho_{k+1} that is EXACTLY zero in exact arithmetic
# -- discrete data, indicator series of a median -- comes out as +-1e-15 depending on the order of the floating-point sums, and the
# scan stops or goes on accordingly (seen: 196 x 2 integers 0..58, BDA, rho_2 + rho_3 = 3e-15 in the oracle, 0 on the device, whose
# indicator sums are exact integers; ESS 367.4 against 372.9).  No restatement can pin that, the reference's own bits would differ from
# both.  The discrete flavours therefore tolerate a few such parameters per call (counted and printed), the others none.
ties_seen = 0


def check(got, want, tol, what, discrete, short=False):
    global ties_seen
    if not discrete:
        return assert_close(got, want, tol, what)
    got, want = np.asarray(got, dtype=np.float64).ravel(), np.asarray(want, dtype=np.float64).ravel()
    assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want)), what
    ok = np.isclose(got, want, rtol=tol, atol=0.0, equal_nan=True) | (np.isinf(got) & (got == want))
    nbad = int((~ok).sum())
    # (split chains of a dozen draws: the lag sums are ratios of very small integers and exact zeros are common -- 4-5 % of the
    # parameters of a 12-draw point-mass series, every one checked by hand to be a pair sum of +-4e-16)
    assert nbad <= max(1, got.size // (10 if short else 300)), (what, nbad, got[~ok][:4], want[~ok][:4])
    if nbad:
        ties_seen += nbad
        print(f"  (Geyer tie on discrete data, {nbad} of {got.size} parameters: {what})")


t0, it, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    it += 1
    big = rng.random() < 0.35
    D = int(rng.integers(2000, 40000)) if big else int(rng.integers(5, 3000))
    C = int(rng.integers(1, 9))
    if D * C > 150000:
        C = max(1, 150000 // D)
    P = int(rng.integers(1, 4))
    if LONG:  # long chains: the 1024-thread geometry, tiled partition over dozens of rounds, per-chain indicator ballots
        D, C, P = int(rng.integers(50000, 131072)), int(rng.integers(1, 9)), int(rng.integers(1, 3))
    if MANY:  # many short chains (one thread per split chain)
        C = int(rng.integers(100, 1500))
        D, P = int(rng.integers(6, max(7, min(400, 150000 // C)))), int(rng.integers(1, 3))
    if ORDER:  # many short parameters: more tickets than workgroups, the second launch is cost-ordered
        D, C, P = int(rng.integers(20, 400)), int(rng.integers(1, 5)), int(rng.integers(600, 4000))
    s = int(rng.integers(1, 4))
    if WAVE:
        os.environ["MDT_WAVE"] = "2"  # (the default takes the wave kernel only from a few thousand parameters on)
        C = int(rng.integers(1, 9))
        D = int(rng.integers(max(8 * s, (256 + C - 1) // C), min(1024 * s, 8192 // C) + 1))
        P = int(rng.choice([1, 3, 17, 300]))
    if it % 2:
        os.environ["MDT_PA"] = "2"
    else:
        os.environ.pop("MDT_PA", None)
    if it % 3 == 0:  # the one-chunk path wherever it can run (by default only slices of 2561 ... 4000 values take it)
        os.environ["MDT_ONECHUNK"] = "2"
    else:
        os.environ.pop("MDT_ONECHUNK", None)
    fl = ("normal", "ar", "ties", "skew", "offset", "mass")[it % 6]
    if fl == "normal":
        x = rng.standard_normal((D, C, P))
    elif fl == "ar":
        x = ar1(float(rng.uniform(-0.9, 0.97)), D, C, P, rng)
    elif fl == "ties":
        x = rng.integers(0, int(rng.integers(2, 60)), size=(D, C, P)).astype(np.float64)
    elif fl == "skew":
        x = np.exp(3.0 * rng.standard_normal((D, C, P)))
    elif fl == "offset":
        x = rng.standard_normal((D, C, P)) + 3.0 * rng.standard_normal((1, C, 1)) + 1e6
    else:  # a point mass holding a large part of the slice
        x = rng.standard_normal((D, C, P))
        x[rng.random((D, C, P)) < rng.uniform(0.05, 0.7)] = 0.25
    f32 = it % 4 == 3 and fl != "offset"  # (1e6 offsets leave Float32 with ~4 digits of the signal)
    if f32:
        x = x.astype(np.float32)
    tol = 5e-6 if f32 else 1e-8
    ml = int(rng.choice([1, 7, 100, 250, 600, 1000] if LONG else [1, 7, 100, 250]))  # (long chains: the lag tiles of the windowed series up to the cap)
    m = ("direct", "fft", "bda")[it % 3]
    cfg = "" if WAVE else ("", "", "small", "big", "tiny")[it % 5]
    if cfg:
        os.environ["MDT_CONFIG"] = cfg
    else:
        os.environ.pop("MDT_CONFIG", None)
    xd = dev(x)
    what = f"it={it} {fl} {x.dtype} D={D} C={C} P={P} s={s} ml={ml} {m} cfg={cfg or 'auto'}"
    disc = fl in ("ties", "mass")
    short = D // s < 50
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kind in ("rank", "bulk", "tail", "basic"):
            S, R = pkg.ess_rhat(xd, kind=kind, autocov_method=methods[m], split_chains=s, maxlag=ml)
            We, Wr = co.ess_rhat(x, kind=kind, autocov=m, split_chains=s, maxlag=ml)
            # (kind = basic on the 1e6-offset flavour works on the raw draws: one ulp of a value is 1.2e-10 of the signal, and
            # the ratio of lag sums amplifies that -- 2e-8 seen once in 918 short parameters; BASELINE's tolerance is 1e-6)
            tk = max(tol, 1e-6) if (fl == "offset" and kind == "basic") else tol
            check(host(S), We, tk, what + f" ess {kind}", disc or (short and kind == "tail"), short)
            assert_close(host(R), Wr, tk, what + f" rhat {kind}")
        wb = co.bulk_tail(x, autocov=m, split_chains=s, maxlag=ml)
        for rep in range(2 if ORDER else 1):
            eb, et, rb, rt = pkg.bulk_tail(xd, autocov_method=methods[m], split_chains=s, maxlag=ml)
            # Geyer's cut-off can tie exactly only where a series is discrete: every ESS of a discrete flavour, and ess_tail (an
            # indicator series) on short chains.  R-hat never goes through Geyer's scan: no slack, ever.
            check(host(eb), wb[0], tol, what + f" fused ess_bulk (launch {rep + 1})", disc, short)
            check(host(et), wb[1], tol, what + f" fused ess_tail (launch {rep + 1})", disc or short, short)
            assert_close(host(rb), wb[2], tol, what + f" fused rhat_bulk (launch {rep + 1})")
            assert_close(host(rt), wb[3], tol, what + f" fused rhat_tail (launch {rep + 1})")
        for est, kind in (("median", np.median), ("mad", "mad")):
            got = pkg.ess(xd, kind=kind, autocov_method=methods[m], split_chains=s, maxlag=ml)
            check(host(got), co.ess_estimator(x, est, autocov=m, split_chains=s, maxlag=ml), tol, what + " " + est, True, short)  # (an indicator series is discrete whatever x is)
        if D // s > 4:  # mcse(kind = median | quantile(p)), src/mcse.jl:66-118: two passes (round 6: both on the wave kernel for short slices)
            pq = float(rng.choice([0.5, 0.1, 0.83]))
            check(host(pkg.mcse(xd, kind=np.median, autocov_method=methods[m], split_chains=s, maxlag=ml)), co.mcse_quantile(x, "median", 0.5, autocov=m, split_chains=s, maxlag=ml), tol, what + " mcse median", True, short)
            check(host(pkg.mcse(xd, kind=pkg.Quantile(pq), autocov_method=methods[m], split_chains=s, maxlag=ml)), co.mcse_quantile(x, "quantile", pq, autocov=m, split_chains=s, maxlag=ml), tol, what + f" mcse quantile {pq}", True, short)
        if D // s > 4 and not f32 and fl != "offset":  # the std proxy (x - mean)^2: ess(kind = std), mcse(kind = std) (round 6: on the wave kernel too)
            check(host(pkg.ess(xd, kind=np.std, autocov_method=methods[m], split_chains=s, maxlag=ml)), co.ess_estimator(x, "std", autocov=m, split_chains=s, maxlag=ml), tol, what + " ess std", disc, short)
            check(host(pkg.mcse(xd, kind=np.std, autocov_method=methods[m], split_chains=s, maxlag=ml)), co.mcse_std(x, autocov=m, split_chains=s, maxlag=ml), tol, what + " mcse std", disc, short)
        if D // s > 4:  # mcse(kind = mean): raw draws (round 6: on short slices through the wave-per-parameter kernel too)
            check(host(pkg.mcse(xd, split_chains=s, maxlag=ml, autocov_method=methods[m])), co.mcse_mean(x, autocov=m, split_chains=s, maxlag=ml), tol, what + " mcse mean", disc, short)
        if C * s >= 2 and C % 2 == 0 and C >= 4:
            ids = np.repeat(np.arange(C // 2), 2)
            got = pkg.rhat_nested(xd, ids, kind="rank", split_chains=s)
            K, S2 = 2, C // 2
            inds = np.arange(C, dtype=np.int32).reshape(S2, K).T.copy()
            want_n = co.rhat_nested(x, inds, kind="rank", split_chains=s)
            assert_close(host(got), want_n, tol, what + " nested")
            if D // s > 4:  # the fused request: mcse(mean) + nested R-hat from one launch
                mm, rr = pkg.mcse_rhat_nested(xd, ids, kind="rank", split_chains=s, maxlag=ml, autocov_method=methods[m])
                assert_close(host(rr), want_n, tol, what + " fused nested")
                check(host(mm), co.mcse_mean(x, autocov=m, split_chains=s, maxlag=ml), tol, what + " fused mcse", disc)
print(f"fuzz soak: {it} random requests x 10-11 calls each agreed with the oracle in {time.time() - t0:.0f} s ({ties_seen} exact Geyer ties on discrete data tolerated)")
