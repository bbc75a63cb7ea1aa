/*
 * mdt_oracle.c -- CPU ORACLE for the ess / rhat / ess_rhat / rhat_nested / mcse(mean) hot path of
 * MCMCDiagnosticTools.jl v0.3.19 (reference at /root/reference, Julia).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it -- always as the checker, never as the thing that
 * is measured or shipped.  The product library (libmdt_hip.so) does not link, load or call it.
 *
 * What it is: a statement-by-statement C restatement of
 *     src/ess_rhat.jl:95-213 (autocov caches), :291-311 (ess kinds, tail), :350-420 (rhat kinds),
 *     :456-624 (ess_rhat), :647-659 (quantile proxy);
 *     src/utils.jl:13-41 (copyto_split!), :148-193 (fold, rank-normalise);
 *     src/rhat_nested.jl:83-188; src/mcse.jl:45-51,
 * plus the third-party behaviour those lines delegate to (none of it is under /root/reference:
 * StatsBase 0.34 tiedrank, StatsFuns 1-2 norminvcdf, Statistics 1.6 median/quantile/var/mean,
 * AbstractFFTs+FFTW fft; compat ranges from Project.toml:23-36, no Manifest), restated from the
 * published algorithms as listed in SURVEY.md Appendix A.
 *
 * PARITY PINNING STATUS.  The reference cannot be run here (Julia absent) and its tests hold NO
 * fixed numeric vectors for this path, so value-level parity at the third-party boundaries
 * (tiedrank / norminvcdf / quantile / FFT outputs as numbers) is "parity unpinned" by reference
 * output.  What IS pinned (tests/test_oracle_*.py):
 *   - every known-answer test the reference's own suite holds for the path: antithetic-chain
 *     ESS == ntotal*log10(ntotal) (test/ess_rhat.jl:314-327), constant input -> NaN
 *     (:242-257), niter<=4 -> NaN ESS + valid R-hat (:66-98), relative scaling (:100-113),
 *     bulk == basic o rank-normalise and monotone invariance (:329-335), FFT ~ direct,
 *     R-hat identical across autocov methods (:210-240), copyto_split! row selection
 *     (test/utils.jl:26-56), rank-normalise moments (:98-107), fold (:109-123),
 *     rhat_nested ~ sqrt(rhat^2 + 1/D), permutation invariance, :rank == max(:bulk,:tail)
 *     (test/rhat_nested.jl:113-155), mcse constant -> NaN (test/mcse.jl:86-101);
 *   - an INDEPENDENT numpy/scipy restatement (oracle/numpy_oracle.py: scipy rankdata, ndtri,
 *     numpy.fft, numpy.quantile) that must agree to <= 1e-10 on the golden inputs; the golden
 *     vectors in tests/golden/ were minted from the two together (tests/golden/make_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -shared).  Three instantiations of mdt_oracle_impl.inc:
 *     _f64           T = double, arithmetic double   (the Float64 reference path)
 *     _f32           T = float,  arithmetic double   ("f64-internal" comparator for the
 *                                                     Float32 device path: rounds where the
 *                                                     reference stores Float32, sums in double)
 *     _f32_faithful  T = float,  arithmetic float    (what the reference does for Float32;
 *                                                     reported, not gated on)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    int64_t draws, chains, params; /* dense column-major x[d + draws*(c + chains*p)] */
    int32_t kind;                  /* 0 basic, 1 bulk, 2 tail, 3 rank */
    int32_t autocov;               /* 0 AutocovMethod, 1 FFTAutocovMethod, 2 BDAAutocovMethod */
    int32_t split_chains;
    int32_t relative;
    int64_t maxlag;
    double tail_prob;
} mdt_oracle_args;

/*
 * StatsFuns.norminvcdf(p) = -erfcinv(2p)*sqrt(2) (call site src/utils.jl:182).  The upstream
 * erfcinv is a ~1 ulp rational approximation whose source is not available here; any Phi^-1
 * accurate to ~1e-16 meets the contract.  This one: Acklam's rational start (rel. err 1.2e-9)
 * polished by two Halley steps on 0.5*erfc(-z/sqrt2) with libm erfc (<1 ulp) -- converges to
 * the rounding floor.  The upper half is evaluated as -Phi^-1(1-p); 1-p is exact for p>=0.5.
 */
double mdt_oracle_norminvcdf(double p) {
    if (isnan(p) || p < 0.0 || p > 1.0) return NAN;
    if (p == 0.0) return -INFINITY;
    if (p == 1.0) return INFINITY;
    if (p > 0.5) return -mdt_oracle_norminvcdf(1.0 - p);
    static const double a[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                                1.383577518672690e+02,  -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                                6.680131188771972e+01,  -1.328068155288572e+01};
    static const double c[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                                -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
    static const double d[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                                3.754408661907416e+00};
    double z;
    if (p < 0.02425) {
        double q = sqrt(-2.0 * log(p));
        z = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
            ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    } else {
        double q = p - 0.5, r = q * q;
        z = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
            (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
    }
    for (int it = 0; it < 2; ++it) {
        double e = 0.5 * erfc(-z * M_SQRT1_2) - p;
        double u = e * 2.5066282746310002 * exp(0.5 * z * z); /* sqrt(2*pi) */
        z = z - u / (1.0 + 0.5 * z * u);
    }
    return z;
}


/* Regularised incomplete beta I_x(a, b) (continued fraction, modified Lentz) and its inverse (safeguarded
 * Newton): StatsFuns.betainvcdf of src/mcse.jl:105-106.  Agrees with scipy.special.betaincinv to ~1e-11. */
static double mdto_betacf(double a, double b, double x) {
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0, tiny = 1e-300;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m < 2000; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double de = d * c;
        h *= de;
        if (fabs(de - 1.0) < 3e-16) break;
    }
    return h;
}
static double mdto_ibeta(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double lb = lgamma(a) + lgamma(b) - lgamma(a + b);
    const double bt = exp(a * log(x) + b * log1p(-x) - lb);
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * mdto_betacf(a, b, x) / a;
    return 1.0 - bt * mdto_betacf(b, a, 1.0 - x) / b;
}
static double mdto_betainvcdf(double a, double b, double p) {
    double lo = 0.0, hi = 1.0, x = a / (a + b);
    { /* start from the normal approximation: mean + z(p) * sd, z by a crude inverse of the logistic fit */
        const double sd = sqrt(a * b / ((a + b) * (a + b) * (a + b + 1.0)));
        const double z = (p > 0.0 && p < 1.0) ? 0.5513288954217921 * log(p / (1.0 - p)) * 1.0 : 0.0; /* sqrt(3)/pi * logit */
        const double x0 = x + z * sd;
        if (x0 > 0.0 && x0 < 1.0) x = x0;
    }
    const double lb = lgamma(a) + lgamma(b) - lgamma(a + b);
    for (int it = 0; it < 200; ++it) {
        const double f = mdto_ibeta(a, b, x) - p;
        if (f > 0.0) hi = x; else lo = x;
        const double pdf = (x > 0.0 && x < 1.0) ? exp((a - 1.0) * log(x) + (b - 1.0) * log1p(-x) - lb) : 0.0;
        double xn = (pdf > 0.0) ? x - f / pdf : 0.5 * (lo + hi);
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (fabs(xn - x) <= 4e-16 * x) { x = xn; break; } /* (two neighbouring doubles would otherwise alternate until the cap) */
        x = xn;
    }
    return x;
}

#define T double
#define A double
#define SFX _f64
#define T_IS_DOUBLE 1
#include "mdt_oracle_impl.inc"
#undef T
#undef A
#undef SFX
#undef T_IS_DOUBLE

#define T float
#define A double
#define SFX _f32
#define T_IS_DOUBLE 0
#include "mdt_oracle_impl.inc"
#undef T
#undef A
#undef SFX
#undef T_IS_DOUBLE

#define T float
#define A float
#define SFX _f32_faithful
#define T_IS_DOUBLE 0
#include "mdt_oracle_impl.inc"
#undef T
#undef A
#undef SFX
#undef T_IS_DOUBLE


/*
 * The same oracle on all host cores (bench.py's `cpu_baseline_all_cores`): the parameters are
 * independent (the `for` over _eachparam, src/ess_rhat.jl:517), so blocks of `chunk` parameters go to the threads of an
 * OpenMP team, each block through the single-threaded functions above on its own sub-array (the params axis is the
 * slowest, a block is contiguous).  The reference itself is single-threaded: this is the generous baseline, the
 * single-core figure stays the contractual one.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <malloc.h>
int mdt_oracle_bulk_tail_omp_f64(const mdt_oracle_args* a, const double* x, double* ess_bulk, double* ess_tail,
                                 double* rhat_bulk, double* rhat_tail, int nthreads, int64_t chunk) {
    const int64_t P = a->params, n = a->draws * a->chains;
    if (chunk < 1) chunk = 1;
    const int64_t nblk = (P + chunk - 1) / chunk;
    int rc_all = 0;
    /* every block allocates and frees slice-sized temporaries: keep them in the thread arenas instead of mapping and
     * unmapping pages on every call (hundreds of threads faulting fresh pages serialise in the kernel) */
    mallopt(M_MMAP_THRESHOLD, 32 << 20); /* (glibc's ceiling: HEAP_MAX_SIZE / 2; a block's largest temporary is a few MB) */
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    /* process-wide settings: put back to glibc's DEFAULT VALUES after the parallel region (below).  Not the allocator the process
     * would have had without this call: setting any of the three switches glibc's DYNAMIC mmap threshold off for good, so
     * whatever the host process measures on the CPU afterwards runs with a fixed 128 KiB threshold.  bench.py runs this
     * leg after the single-thread baseline of the line (cpu_baseline), which is the contractual one. */
#ifdef _OPENMP
    if (nthreads < 1) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t p0 = b * chunk, cnt = (p0 + chunk <= P) ? chunk : P - p0;
        mdt_oracle_args s = *a;
        s.params = cnt;
        const int rc = mdt_oracle_bulk_tail_f64(&s, x + p0 * n, ess_bulk + p0, ess_tail + p0, rhat_bulk + p0, rhat_tail + p0);
        if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
        }
    }
    mallopt(M_MMAP_THRESHOLD, 128 * 1024); /* glibc defaults (DEFAULT_MMAP_THRESHOLD, DEFAULT_TRIM_THRESHOLD, DEFAULT_TOP_PAD); */
    mallopt(M_TRIM_THRESHOLD, 128 * 1024); /* setting them explicitly also switches the dynamic mmap threshold off, as the   */
    mallopt(M_TOP_PAD, 128 * 1024);        /* first three calls already did                                                  */
    malloc_trim(0);
    return rc_all;
}

int mdt_oracle_omp_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
