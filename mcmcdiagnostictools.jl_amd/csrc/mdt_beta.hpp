// mdt_beta.hpp -- regularised incomplete beta function and its inverse, for host and device code.
//
// StatsFuns.betainvcdf as used by src/mcse.jl:105-106 (quadrature points of the Beta error distribution) and,
// through Distributions.quantile(FDist(d1, d2), q) = d2 y / (d1 (1 - y)), y = betainvcdf(d1/2, d2/2, q), by
// src/gelmandiag.jl:50.  Continued fraction of I_x(a, b) + safeguarded Halley / Newton; agrees with
// scipy.special.betaincinv to ~1e-11 (tests/test_host_api.py).
#pragma once
#include <math.h>

#define MDT_HD __host__ __device__

// Continued fraction of the incomplete beta function, modified Lentz.  Each half step needs 1/(1 + t d) and
// 1/(1 + t/c); tracking rc = 1/c makes both plain reciprocals, taken through ONE division of their product, and
// the two partial numerators of a step share one more: three divisions per step instead of six (one thread per
// parameter makes this loop latency-bound, and a Float64 division is ~35 dependent instructions on CDNA).
// (The forward Wallis recurrences need a single division per step but lose ~3 digits to cancellation when
// x(a + b)/a is close to 1; tried and dropped.)
MDT_HD static inline double mdt_betacf(double a, double b, double x) {
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0, tiny = 1e-300;
    double rc = 1.0, d = 1.0 - qab * x / qap; // rc = 1/c
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m < 2000; ++m) {
        const double m2 = 2.0 * m;
        const double d1 = (qam + m2) * (a + m2), d2 = (a + m2) * (qap + m2);
        const double r = 1.0 / (d1 * d2);
        const double t1 = m * (b - m) * x * d2 * r;          //  m (b - m) x / ((a + 2m - 1)(a + 2m))
        const double t2 = -(a + m) * (qab + m) * x * d1 * r; // -(a + m)(a + b + m) x / ((a + 2m)(a + 2m + 1))
        double u = 1.0 + t1 * d, c = 1.0 + t1 * rc;
        if (fabs(u) < tiny) u = tiny;
        if (fabs(c) < tiny) c = tiny;
        double q = 1.0 / (u * c);
        d = q * c; rc = q * u; // 1/u, 1/c
        h *= d * c;
        u = 1.0 + t2 * d; c = 1.0 + t2 * rc;
        if (fabs(u) < tiny) u = tiny;
        if (fabs(c) < tiny) c = tiny;
        q = 1.0 / (u * c);
        d = q * c; rc = q * u;
        const double de = d * c;
        h *= de;
        if (fabs(de - 1.0) < 3e-16) break;
    }
    return h;
}
// lgamma(z) - [(z - 1/2) log z - z + log(2 pi)/2] for z >= 10 (Stirling series, next term < 2e-14)
MDT_HD static inline double mdt_stirling_corr(double z) {
    const double r = 1.0 / z, r2 = r * r;
    return r * (1.0 / 12.0 - r2 * (1.0 / 360.0 - r2 * (1.0 / 1260.0 - r2 * (1.0 / 1680.0 - r2 * (1.0 / 1188.0)))));
}
// log Beta(a, b) without the cancellation of lgamma(a) + lgamma(b) - lgamma(a + b) for large arguments (three
// numbers of size b log b whose sum is of size a log b): Stirling's form regrouped around log1p
MDT_HD static inline double mdt_lbeta(double a, double b) {
    if (a > b) { const double t = a; a = b; b = t; }
    if (a >= 10.0) {
        const double ab = a + b;
        return 0.9189385332046727 - 0.5 * log(b) + (a - 0.5) * log(a / ab) + b * log1p(-a / ab) +
               (mdt_stirling_corr(a) + mdt_stirling_corr(b) - mdt_stirling_corr(ab));
    }
    if (b >= 10.0)
        return lgamma(a) + (-(a + b - 0.5) * log1p(a / b) - a * log(b) + a + (mdt_stirling_corr(b) - mdt_stirling_corr(a + b)));
    return lgamma(a) + lgamma(b) - lgamma(a + b);
}
MDT_HD static inline double mdt_ibeta_lb(double a, double b, double x, double lb) { // lb = log Beta(a, b)
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double bt = exp(a * log(x) + b * log1p(-x) - lb);
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * mdt_betacf(a, b, x) / a;
    return 1.0 - bt * mdt_betacf(b, a, 1.0 - x) / b;
}
MDT_HD static inline double mdt_ibeta(double a, double b, double x) {
    return mdt_ibeta_lb(a, b, x, mdt_lbeta(a, b));
}
// inverse in x of I_x(a, b) = p; `xstart` in (0, 1) replaces the built-in start
MDT_HD static inline double mdt_betainvcdf(double a, double b, double p, double xstart = -1.0) {
    double lo = 0.0, hi = 1.0, x = a / (a + b);
    if (xstart > 0.0 && xstart < 1.0) x = xstart;
    else { /* start from the normal approximation: mean + z(p) * sd, z by a crude inverse of the logistic fit */
        const double sd = sqrt(a * b / ((a + b) * (a + b) * (a + b + 1.0)));
        const double z = (p > 0.0 && p < 1.0) ? 0.5513288954217921 * log(p / (1.0 - p)) : 0.0; /* sqrt(3)/pi * logit */
        const double x0 = x + z * sd;
        if (x0 > 0.0 && x0 < 1.0) x = x0;
    }
    const double lb = mdt_lbeta(a, b);
    for (int it = 0; it < 200; ++it) {
        const double f = mdt_ibeta_lb(a, b, x, lb) - p;
        if (f > 0.0) hi = x; else lo = x;
        const double pdf = (x > 0.0 && x < 1.0) ? exp((a - 1.0) * log(x) + (b - 1.0) * log1p(-x) - lb) : 0.0;
        // Halley's step (the density's logarithmic derivative is two divisions): cubic convergence -- every step costs a
        // continued fraction, and this loop runs one thread per parameter; plain Newton where the correction is wild
        double xn = 0.5 * (lo + hi);
        if (pdf > 0.0) {
            const double nt = f / pdf, g = (a - 1.0) / x - (b - 1.0) / (1.0 - x), den = 1.0 - 0.5 * nt * g;
            xn = x - ((den > 0.5 && den < 2.0) ? nt / den : nt);
        }
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (fabs(xn - x) <= 4e-16 * x) { x = xn; break; } /* (two neighbouring doubles would otherwise alternate until the cap) */
        x = xn;
    }
    return x;
}
