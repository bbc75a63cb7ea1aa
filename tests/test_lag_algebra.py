"""The two reformulations of the lag products sum_t u[t] u[t + k] (src/ess_rhat.jl:170-178) that round 4 put on the device,
restated in numpy and held to the plain sums (CPU; the kernels themselves are held to the oracle by the GPU tests):

* blocked Hankel tiles (csrc/mdt_series.inc, lag_tiles_to_lp): rows of 16 draws, G_c[i][j] = sum_r u[16 r + i] u[16 r + j + c];
  lag k of residue i sits in column k + i of the strip [G_k0 | G_k0+16 | ...] -- what v_mfma_f64_16x16x4_f64 accumulates four
  rows at a time;
* pooled products (csrc/mdt_short.inc, oc_series_phase): one sweep over all split chains laid end to end, minus the k pairs per
  boundary and lag that straddle two chains, is the sum over the chains of their own lag products; centring per chain is the
  algebra of series_z_phase.
"""
import numpy as np
import pytest


def lag_sums(u, lags):
    return np.array([float(np.dot(u[: len(u) - k], u[k:])) if k < len(u) else 0.0 for k in lags])


@pytest.mark.parametrize("n,k0,ntile", [(5000, 0, 3), (5000, 32, 6), (777, 0, 3), (64, 0, 2), (8192, 32, 15), (1000, 112, 6)])
def test_hankel_tiles_hold_the_lag_products_on_their_diagonals(n, k0, ntile):
    rng = np.random.default_rng(n + k0)
    u = rng.standard_normal(n)
    nblk = (n + 63) // 64
    pad = np.zeros(64 * nblk + k0 + 16 * ntile + 64)
    pad[:n] = u                                   # zeros behind the chain: what the shifted operand reads past its end
    strip = np.zeros((16, 16 * ntile))
    for R in range(nblk):                         # one MFMA per tile and 64 values: A = u[64 R + lane], B = the same shifted by c
        a = pad[64 * R: 64 * R + 64].reshape(4, 16)            # four rows of 16 draws
        for t in range(ntile):
            c = k0 + 16 * t
            b = pad[64 * R + c: 64 * R + c + 64].reshape(4, 16)
            strip[:, 16 * t: 16 * t + 16] += a.T @ b          # G_c[i][j] += sum_rows a[row][i] b[row][j]
    nfull = 16 * (ntile - 1)                      # lags covered by every residue
    got = np.array([sum(strip[i, kap + i] for i in range(16)) for kap in range(nfull)])
    want = lag_sums(u, range(k0, k0 + nfull))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("niter,m,maxlag", [(500, 8, 100), (64, 8, 30), (1000, 4, 250), (333, 3, 60), (2000, 1, 250)])
def test_pooled_products_minus_boundary_pairs_are_the_sum_over_chains(niter, m, maxlag):
    rng = np.random.default_rng(niter * m)
    z = rng.standard_normal((m, niter)) + 0.3 * rng.standard_normal((m, 1))    # chains with their own means
    K = z[:, 0].copy()                                                       # the shift of every chain: its first score
    u = z - K[:, None]
    pooled = u.reshape(-1)
    n = niter
    lags = np.arange(0, maxlag + 1)
    # one sweep over the pooled series, minus the pairs (e, e + k) that straddle a boundary b = j n
    P = lag_sums(pooled, lags)
    for j in range(1, m):
        b = j * n
        for k in lags:
            P[k] -= float(np.dot(pooled[b - k: b], pooled[b: b + k]))
    want_raw = sum(lag_sums(u[j], lags) for j in range(m))
    np.testing.assert_allclose(P, want_raw, rtol=0, atol=1e-9 * float(np.abs(want_raw).max()))
    # centring terms summed over the chains: sum_j sum_t (u - mu_j)(u' - mu_j) = P_k + (n - k) sum mu_j^2 - 2 sum mu_j S1_j + sum_{t<k} (EH[t] + ET[t])
    mu = u.mean(axis=1)
    S1 = u.sum(axis=1)
    EH = (mu[:, None] * u).sum(axis=0)                     # EH[t] = sum_j mu_j u_j[t]
    ET = (mu[:, None] * u[:, ::-1]).sum(axis=0)            # ET[t] = sum_j mu_j u_j[n - 1 - t]
    cen = np.array([P[k] + (n - k) * float(mu @ mu) - 2.0 * float(mu @ S1) + float(EH[:k].sum() + ET[:k].sum()) for k in lags])
    want = sum(lag_sums(u[j] - mu[j], lags) for j in range(m))
    np.testing.assert_allclose(cen[1:], want[1:], rtol=0, atol=1e-9 * float(np.abs(want).max()))
    # the shift cancels: the same centred sums from the scores themselves
    want_z = sum(lag_sums(z[j] - z[j].mean(), lags) for j in range(m))
    np.testing.assert_allclose(want, want_z, rtol=0, atol=1e-9 * float(np.abs(want_z).max()))


def test_first_sweep_prediction_is_a_statement_about_the_floor_of_rho():
    """series_win_phase takes every lag in its first sweep when 1 - W / var+ stands four noise widths clear of zero: with chains
    that disagree, rho_k = 1 - (W - acov_k) / var+ (src/ess_rhat.jl:577-586) tends to that floor and Geyer's pair sums stay positive."""
    rng = np.random.default_rng(7)
    n, m = 20000, 8
    for offs, expect in ((0.0, False), (0.2, True)):
        z = rng.standard_normal((m, n)) + offs * np.arange(m)[:, None]
        W = z.var(axis=1, ddof=1).mean()
        varp = (n - 1) / n * W + z.mean(axis=1).var(ddof=1)
        floor = 1.0 - W / varp
        assert bool(floor * np.sqrt(n * m) > 4.0) == expect
        if expect:   # the scan indeed never stops: every pair sum of the pooled autocorrelation is positive up to lag 250
            acov = np.array([np.mean([np.dot(c[: n - k] - c.mean(), c[k:] - c.mean()) / n for c in z]) for k in range(1, 251)])
            rho = 1.0 - (W - acov) / varp
            assert np.all(rho[0::2][:124] + rho[1::2][:124] > 0.0)
