"""Independent numpy/scipy restatement of the ess/rhat hot path -- TEST INFRASTRUCTURE ONLY.

Purpose: a second opinion for oracle/mdt_oracle.c.  It deliberately uses different building
blocks (scipy.stats.rankdata, scipy.special.ndtri, numpy.fft with a power-of-two pad,
numpy.quantile, numpy.median) so that an error in reading the reference would have to be made
twice, differently, to go unnoticed.  The two must agree to <= 1e-10 relative on the golden
inputs (tests/test_oracle_golden.py::test_numpy_oracle_matches_golden); the golden vectors in tests/golden/ are minted from both
(tests/golden/make_golden.py).  Float64 only.  Never imported by the product.

Reference lines restated (relative to /root/reference): src/ess_rhat.jl:161-213, 301-311,
350-420, 456-624, 647-659; src/utils.jl:13-41, 148-193; src/rhat_nested.jl:127-188;
src/mcse.jl:45-51.
"""
import numpy as np
from scipy.special import ndtri
from scipy.stats import rankdata


def _jl_min(a, b):
    return np.where(np.isnan(a) | np.isnan(b), np.nan, np.minimum(a, b))


def _jl_max(a, b):
    return np.where(np.isnan(a) | np.isnan(b), np.nan, np.maximum(a, b))


def split_chains_2d(x, s):
    """copyto_split! (src/utils.jl:13-41): x (D, C) -> (D//s, C*s)."""
    D, C = x.shape
    niter, rem = divmod(D, s)
    cols = []
    for c in range(C):
        off = 0
        for k in range(1, s + 1):
            cols.append(x[off:off + niter, c])
            off += niter + (1 if k <= rem else 0)
    return np.stack(cols, axis=1) if cols else np.empty((niter, 0))


def rank_normalize(xp):
    """src/utils.jl:169-193 for one (D, C) slice; pooled column-major."""
    v = xp.ravel(order="F")
    n = v.size
    r = rankdata(v, method="average")
    z = ndtri((r - 0.375) / (n + 0.25))
    return z.reshape(xp.shape, order="F")


def fold(xp):
    return np.abs(xp - np.median(xp))


def _moments(xp, s):
    y = split_chains_2d(xp, s)
    niter, m = y.shape
    mu = y.mean(axis=0)
    var = ((y - mu) ** 2).sum(axis=0) / (niter - 1)
    W = var.mean()
    vb = ((mu - mu.mean()) ** 2).sum() / (m - 1 if m > 1 else m)
    varp = (niter - 1) / niter * W + vb
    return y, mu, var, W, varp


def rhat_basic(xp, s=2):
    with np.errstate(all="ignore"):
        _, _, _, W, varp = _moments(xp, s)
        return np.sqrt(varp / W)


def _autocov_all(yc, var, method, nl):
    """mean over chains of the lag-k autocovariance, k = 0..nl-1 (src/ess_rhat.jl:161-213)."""
    niter, m = yc.shape
    if method == "direct":
        return np.array([np.mean([yc[: niter - k, j] @ yc[k:, j] for j in range(m)]) / niter
                         for k in range(nl)])
    if method == "fft":
        N = 1
        while N < 2 * niter - 1:
            N *= 2
        F = np.fft.fft(yc, n=N, axis=0)
        ac = np.fft.ifft(np.abs(F) ** 2, axis=0).real
        return (ac[:nl] / ac[0] * var).mean(axis=1) * ((niter - 1) / niter)
    if method == "bda":
        mv = var.mean()
        out = []
        for k in range(nl):
            n = niter - k
            sdiff = np.mean([((yc[:n, j] - yc[k:, j]) ** 2).sum() for j in range(m)])
            out.append(mv - sdiff / (2 * n))
        return np.array(out)
    raise ValueError(method)


def ess_rhat_basic(xp, s=2, maxlag=250, method="direct", relative=False):
    """_ess_rhat(Val(:basic)) for one (D, C) slice (src/ess_rhat.jl:456-603)."""
    D, C = xp.shape
    niter = D // s
    if not niter > 4:
        return np.nan, rhat_basic(xp, s)
    if not maxlag > 0:
        raise ValueError("DomainError")
    maxlag = min(maxlag, niter - 4)
    with np.errstate(all="ignore"):
        y, mu, var, W, varp = _moments(xp, s)
        m = y.shape[1]
        ntotal = niter * m
        rhat = np.sqrt(varp / W)
        yc = y - mu
        acov = _autocov_all(yc, var, method, maxlag + 1)
        rho = 1 - (W - acov) / varp
        rho_odd = rho[1]
        pt = 1.0 + rho_odd
        sum_pt = pt
        k = 2
        while k < maxlag - 1:
            delta = rho[k] + rho[k + 1]
            if not delta > 0:
                break
            pt = float(_jl_min(delta, pt))
            sum_pt += pt
            k += 2
        rho_even = rho[k] if maxlag > 1 else 0.0
        tau = float(_jl_max(0.0, 2 * sum_pt + float(_jl_max(0.0, rho_even)) - 1))
        ess = float(_jl_min(1.0 / tau if tau != 0 else np.inf, np.log10(ntotal)))
        if not relative:
            ess *= ntotal
    return ess, rhat


def quantile7(v, p):
    return np.quantile(v, p, method="linear")


def ess_rhat(x, kind="rank", method="direct", s=2, maxlag=250, relative=False, tail_prob=0.1):
    """x: (D, C, P). Returns (ess[P], rhat[P]) like ess_rhat(x; kind) (src/ess_rhat.jl:438-455,
    604-624)."""
    x = np.asarray(x, dtype=np.float64)
    P = x.shape[2]
    ess = np.full(P, np.nan)
    rhat = np.full(P, np.nan)
    for p in range(P):
        xp = x[:, :, p]
        if kind == "basic":
            ess[p], rhat[p] = ess_rhat_basic(xp, s, maxlag, method, relative)
        elif kind == "bulk":
            ess[p], rhat[p] = ess_rhat_basic(rank_normalize(xp), s, maxlag, method, relative)
        elif kind == "tail":
            lo = (xp <= quantile7(xp.ravel(order="F"), tail_prob / 2)).astype(np.float64)
            hi = (xp <= quantile7(xp.ravel(order="F"), 1 - tail_prob / 2)).astype(np.float64)
            e1, _ = ess_rhat_basic(lo, s, maxlag, method, relative)
            e2, _ = ess_rhat_basic(hi, s, maxlag, method, relative)
            ess[p] = float(_jl_min(e1, e2))
            rhat[p] = rhat_basic(rank_normalize(fold(xp)), s)
        elif kind == "rank":
            ess[p], rb = ess_rhat_basic(rank_normalize(xp), s, maxlag, method, relative)
            rt = rhat_basic(rank_normalize(fold(xp)), s)
            rhat[p] = float(_jl_max(rt, rb))
        else:
            raise ValueError(kind)
    return ess, rhat


def rhat_nested_basic(xp, inds, s=2):
    """src/rhat_nested.jl:127-188 for one (D, C) slice; inds (K, S) 0-based."""
    K, S = inds.shape
    within = 0.0
    sm = np.empty(S)
    with np.errstate(all="ignore"):
        for k in range(S):
            y = split_chains_2d(xp[:, inds[:, k]], s)
            niter, mk = y.shape
            mu = y.mean(axis=0)
            sm[k] = mu.mean()
            var = ((y - mu) ** 2).sum(axis=0) / (niter - 1)
            Bk = ((mu - mu.mean()) ** 2).sum() / (mk - 1 if mk > 1 else mk)
            within += var.mean() + Bk
        within /= S
        between = ((sm - sm.mean()) ** 2).sum() / (S - 1)
        return np.sqrt(1 + between / within)


def rhat_nested(x, inds, kind="rank", s=2):
    x = np.asarray(x, dtype=np.float64)
    P = x.shape[2]
    out = np.empty(P)
    for p in range(P):
        xp = x[:, :, p]
        if kind == "basic":
            out[p] = rhat_nested_basic(xp, inds, s)
        elif kind == "bulk":
            out[p] = rhat_nested_basic(rank_normalize(xp), inds, s)
        elif kind == "tail":
            out[p] = rhat_nested_basic(rank_normalize(fold(xp)), inds, s)
        elif kind == "rank":
            out[p] = float(_jl_max(rhat_nested_basic(rank_normalize(xp), inds, s),
                                   rhat_nested_basic(rank_normalize(fold(xp)), inds, s)))
    return out


def mcse_mean(x, s=2, maxlag=250, method="direct"):
    """src/mcse.jl:45-51."""
    x = np.asarray(x, dtype=np.float64)
    P = x.shape[2]
    out = np.empty(P)
    for p in range(P):
        S, _ = ess_rhat_basic(x[:, :, p], s, maxlag, method, False)
        with np.errstate(all="ignore"):
            out[p] = np.std(x[:, :, p], ddof=1) / np.sqrt(S)
    return out


def ess_estimator(x, estimator, prob=0.5, method="direct", s=2, maxlag=250, relative=False):
    """ess(x; kind=estimator) through the expectand proxies of src/ess_rhat.jl:626-659."""
    x = np.asarray(x, dtype=np.float64)
    P = x.shape[2]
    out = np.empty(P)
    for p in range(P):
        xp = x[:, :, p]
        if estimator == "median":
            y = (xp <= np.median(xp)).astype(np.float64)
        elif estimator == "std":
            y = (xp - xp.mean()) ** 2
        elif estimator == "mad":
            f = fold(xp)
            y = (f <= np.median(f)).astype(np.float64)
        elif estimator == "quantile":
            y = (xp <= quantile7(xp.ravel(order="F"), prob)).astype(np.float64)
        else:
            raise ValueError(estimator)
        out[p], _ = ess_rhat_basic(y, s, maxlag, method, relative)
    return out


def mcse_std(x, s=2, maxlag=250, method="direct"):
    """src/mcse.jl:52-65."""
    x = np.asarray(x, dtype=np.float64)
    P = x.shape[2]
    out = np.empty(P)
    for p in range(P):
        y = (x[:, :, p] - x[:, :, p].mean()) ** 2
        S, _ = ess_rhat_basic(y, s, maxlag, method, False)
        mv, m4 = y.mean(), (y ** 2).mean()
        with np.errstate(all="ignore"):
            out[p] = np.sqrt((m4 / mv - mv) / S) / 2
    return out


def mcse_quantile(x, estimator="median", prob=0.5, s=2, maxlag=250, method="direct"):
    """src/mcse.jl:66-118 (scipy.special.betaincinv stands in for StatsFuns.betainvcdf)."""
    from scipy.special import betaincinv
    x = np.asarray(x, dtype=np.float64)
    P = x.shape[2]
    p = 0.5 if estimator == "median" else prob
    S = ess_estimator(x, estimator, prob=prob, method=method, s=s, maxlag=maxlag)
    out = np.empty(P)
    for q in range(P):
        if np.isnan(S[q]):
            out[q] = np.nan
            continue
        v = np.sort(x[:, :, q].ravel(order="F"))
        n = v.size
        a, b = S[q] * p + 1, S[q] * (1 - p) + 1
        pu, pl = betaincinv(a, b, 0.8413447460685429), betaincinv(a, b, 0.15865525393145705)
        lo, up = max(int(np.floor(pl * n)), 1), min(int(np.ceil(pu * n)), n)
        out[q] = (v[up - 1] - v[lo - 1]) / 2
    return out


def bfmi(energy):
    """src/bfmi.jl:38-45"""
    e = np.asarray(energy, dtype=np.float64)
    with np.errstate(all="ignore"):
        return np.mean(np.diff(e, axis=0) ** 2, axis=0) / np.var(e, axis=0, ddof=1)


def gelmandiag(x, alpha=0.05):
    """src/gelmandiag.jl:1-60 (scipy.stats.f.ppf stands in for Distributions.quantile(FDist))."""
    from scipy.stats import f as fdist
    x = np.asarray(x, dtype=np.float64)
    n, m, P = x.shape
    rfixed, rrs = (n - 1) / n, (m + 1) / (m * n)
    psibar = x.mean(axis=0)                      # (m, P)
    s2 = x.var(axis=0, ddof=1)                   # (m, P)
    w = s2.mean(axis=0)
    b = n * psibar.var(axis=0, ddof=1)
    psibar2 = psibar.mean(axis=0)

    def cov(a, c):
        return ((a - a.mean(axis=0)) * (c - c.mean(axis=0))).sum(axis=0) / (m - 1)
    with np.errstate(all="ignore"):
        var_w = s2.var(axis=0, ddof=1) / m
        var_b = 2 / (m - 1) * b ** 2
        var_wb = n / m * (cov(s2, psibar ** 2) - 2 * psibar2 * cov(s2, psibar))
        V = rfixed * w + rrs * b
        var_V = rfixed ** 2 * var_w + rrs ** 2 * var_b + 2 * rfixed * rrs * var_wb
        df = 2 * V ** 2 / var_V
        W_df = 2 * w ** 2 / var_w
        corr = (df + 3) / (df + 1)
        rrandom = rrs * b / w
        psrf = np.sqrt(corr * (rfixed + rrandom))
        fq = np.where(np.isnan(rrandom), 1.0, fdist.ppf(1 - alpha / 2, m - 1, W_df))
        ci = np.sqrt(corr * (rfixed + rrandom * fq))
    return psrf, ci
