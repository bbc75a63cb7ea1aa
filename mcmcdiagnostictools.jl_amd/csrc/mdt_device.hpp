// mdt_device.hpp -- device code of the MI355X (gfx950) ESS / R-hat engine.
//
// One 1024-thread workgroup (16 wave64) owns one parameter at a time and walks the params axis
// persistently (grid = #CUs; the 150 KB LDS working set allows one workgroup per CU).  For a
// parameter slice x[0..n), n = draws*chains (column-major, draws fastest) it runs, all on chip:
//
//   rank phase    pooled tied ranks of x (or of |x - median|) WITHOUT a sort: a 1024-element
//                 sample is bitonic-sorted, 256 sample quantiles become coarse splitters, each
//                 coarse bucket is cut linearly into 31 fine bins (+1 bin reserved for values
//                 equal to the splitter), an LDS histogram + prefix gives every bin its rank
//                 base, and the rank inside a bin (~5 elements) is an all-pairs count against
//                 the bin's members staged in LDS, ~14K elements per round.  Exact for any
//                 input: binning only has to be monotone, ties/NaN/Inf/-0.0 follow
//                 StatsBase.tiedrank + isless (src/utils.jl:180).  Order statistics (median,
//                 type-7 quantile neighbours) drop out of the same counting step.
//   series phase  split-chain means, variances, lagged products (src/ess_rhat.jl:526-551,
//                 161-179) from an LDS-staged, centred tile of the series; lags are computed
//                 lazily in blocks and Geyer's initial monotone sequence (:555-594) runs on one
//                 lane between blocks, so well-mixed chains stop after 16 lags.
//   indicator     tail-ESS (src/ess_rhat.jl:301-311, 647-659): the quantile indicators are
//                 bit-vectors, their autocovariance is popcount(bits & bits>>k), exact integers.
//
// Everything is deterministic (no floating-point atomics; fixed reduction trees), which the
// reference's tests demand (test/ess_rhat.jl:167-204: results identical wherever a parameter
// sits in the batch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdt {

constexpr int NT = 1024;            // threads per workgroup
constexpr int NWAVE = NT / 64;      // 16 wave64
constexpr int K1 = 256;             // coarse buckets (sample quantiles)
constexpr int FINE = 32;            // bins per coarse bucket: 1 reserved (== splitter) + 31 linear
constexpr int NREG = K1 * FINE;     // regular bins are 1..NREG
constexpr int BIN_UNDER = 0;        // below the smallest sample value
constexpr int BIN_TOPEQ = NREG + 1; // equal to the largest sample value (reserved)
constexpr int BIN_OVER = NREG + 2;  // above the largest sample value
constexpr int BIN_NAN = NREG + 3;   // NaN: ranked last, in index order (isless / stable sortperm)
constexpr int NBIN = NREG + 4;
constexpr int CH = 13824;           // elements of bin-ordered data staged in LDS per round
constexpr int MAXROUNDS = 320;      // => n <= ~4.4M elements per parameter
constexpr int TL = 8192;            // series tile (elements of one split chain resident in LDS)
constexpr int HALO = 1040;          // >= MAXLAG_CAP + 1 + 32 (one lag block past the last lag)
constexpr int MAXLAG_CAP = 1000;
constexpr int MAXM = 256;           // max number of split chains chains*split_chains
constexpr int NLAGTAB = 1024;

// ------------------------------------------------------------------ LDS carve (bytes)
constexpr int OFF_RED = 0;                                // double red[32]
constexpr int OFF_OSTAT = OFF_RED + 32 * 8;               // double ostat[8]
constexpr int OFF_SCAL = OFF_OSTAT + 8 * 8;               // double scal[24]
constexpr int OFF_TPOS = OFF_SCAL + 24 * 8;               // uint32 tpos[8]
constexpr int OFF_INTS = OFF_TPOS + 8 * 4;                // int ints[24]
constexpr int OFF_WTOT = OFF_INTS + 24 * 4;               // uint32 wtot[16]
constexpr int OFF_RLO = OFF_WTOT + 16 * 4;                // int rlo[MAXROUNDS]
constexpr int OFF_RHI = OFF_RLO + MAXROUNDS * 4;          // int rhi[MAXROUNDS]
constexpr int OFF_RBASE = OFF_RHI + MAXROUNDS * 4;        // uint32 rbase[MAXROUNDS]
constexpr int OFF_MEAN = OFF_RBASE + MAXROUNDS * 4;       // double mean_j[MAXM]
constexpr int OFF_VAR = OFF_MEAN + MAXM * 8;              // double var_j[MAXM]
constexpr int OFF_BIG = ((OFF_VAR + MAXM * 8 + 15) / 16) * 16;
// rank phase view of the big region
constexpr int OFF_SPLT = OFF_BIG;                         // double t[K1+2]
constexpr int OFF_SPLS = OFF_SPLT + (K1 + 2) * 8;         // double scale[K1]
constexpr int OFF_CUR = OFF_SPLS + K1 * 8;                // uint32 cur[NBIN+4]
constexpr int OFF_CHUNK = ((OFF_CUR + (NBIN + 4) * 4 + 15) / 16) * 16; // double chunk[CH]
constexpr int LDS_RANK_END = OFF_CHUNK + CH * 8;
// series phase view of the big region
constexpr int OFF_SER = OFF_BIG;                          // double ser[TL+HALO]
constexpr int OFF_LAGA = OFF_SER + (TL + HALO) * 8;       // double A[NLAGTAB]   sum over chains of lag sums
constexpr int OFF_PART = OFF_LAGA + NLAGTAB * 8;          // double part[NWAVE*8]
constexpr int OFF_AUX = OFF_PART + NWAVE * 8 * 8;         // double aux[2*NLAGTAB] (bda head/tail squares)
constexpr int LDS_SER_END = OFF_AUX + 2 * NLAGTAB * 8;
// indicator phase view of the big region
constexpr int OFF_BITS = OFF_BIG;                         // uint64 bits[2][BITW]  raw ballots of one chain
constexpr int BITW = 2048;                                // => draws <= 131072
constexpr int OFF_SBITS = OFF_BITS + 2 * BITW * 8;        // uint64 sbits[2][BITW] split-aligned words
constexpr int OFF_ICNT = OFF_SBITS + 2 * BITW * 8;        // double icov[2][NLAGTAB] sum over chains
constexpr int OFF_IDIF = OFF_ICNT + 2 * NLAGTAB * 8;      // double idif[2][NLAGTAB] sum over chains of #(I_t != I_{t+k})
constexpr int OFF_IVAR = OFF_IDIF + 2 * NLAGTAB * 8;      // double imean[2][MAXM]
constexpr int LDS_IND_END = OFF_IVAR + 2 * MAXM * 8;
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int LDS_BYTES = cmax(LDS_RANK_END, cmax(LDS_SER_END, LDS_IND_END));
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget (160 KiB per CU on gfx950)");

// ------------------------------------------------------------------ kernel parameters
struct KP {
    const void* x;
    long long D, C, P, n;       // draws, chains, params, n = D*C
    long long niter;            // D / s
    int s, rem, m;              // split_chains, D % s, m = C*s
    int mode;                   // 0 ess_rhat / bulk_tail, 1 rhat_nested, 2 mcse_mean, 3 rank_normalize stage
    int basic;                  // kind == basic (no ranking); outputs via out_eb / out_rb
    int need_eb, need_rb, need_et, need_rt; // bulk ESS / bulk Rhat / tail ESS / tail Rhat
    int method;                 // 0 direct, 1 fft, 2 bda
    int ess_ok;                 // niter > 4
    int relative;
    int fold;                   // rank_normalize stage: rank |x - median| instead of x
    long long maxlag;           // already min(maxlag, niter - 4)
    double rel_ess_max;         // log10(ntotal) evaluated in T on the host (src/ess_rhat.jl:514)
    double ntotal;              // niter * m
    double cf;                  // (niter - 1) / niter
    long long q_j[2];           // type-7 quantile: 1-based lower order statistic j (lo, hi)
    double q_g[2];              // and gamma, both evaluated in T on the host
    const double* ztab;         // normal scores by doubled rank, [2n+1]
    unsigned int* r2buf;        // per-workgroup scratch: doubled ranks, stride n_pad
    unsigned short* binbuf;     // per-workgroup scratch: bin ids, stride n_pad
    long long n_pad;
    void* out_eb; void* out_rb; void* out_et; void* out_rt; // [P] of T, or null
    void* out_rrank;            // [P] max(rhat_tail, rhat_bulk) (kind = rank), or null
    double* rank_out;           // stage: [P*n] tied ranks
    void* z_out;                // stage: [P*n] normal scores (T)
    const int* chain_inds;      // nested: K x S (device)
    int K, S;
    int* status;                // device status word (bit0: rounds overflow, bit1: NaN hit quantile)
};

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ double jl_min(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }
__device__ __forceinline__ double jl_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// deterministic workgroup sum; every thread returns the same value
__device__ __forceinline__ double wg_sum(double v, double* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) s += red[i];
    __syncthreads();
    return s;
}

// Phi^-1 to the rounding floor: Acklam's rational start polished by two Halley steps on erfc.
// Only used to build the normal-score table (once per (n, dtype)), never per element.
__device__ inline double norminv_acc(double p) {
    if (!(p > 0.0) || !(p < 1.0)) return (p == 0.0) ? -__builtin_inf() : (p == 1.0 ? __builtin_inf() : __builtin_nan(""));
    bool upper = p > 0.5;
    if (upper) p = 1.0 - p; // exact for p >= 0.5
    double z;
    if (p < 0.02425) {
        double q = sqrt(-2.0 * log(p));
        z = (((((-7.784894002430293e-03 * q + -3.223964580411365e-01) * q + -2.400758277161838e+00) * q + -2.549732539343734e+00) * q + 4.374664141464968e+00) * q + 2.938163982698783e+00) /
            ((((7.784695709041462e-03 * q + 3.224671290700398e-01) * q + 2.445134137142996e+00) * q + 3.754408661907416e+00) * q + 1.0);
    } else {
        double q = p - 0.5, r = q * q;
        z = (((((-3.969683028665376e+01 * r + 2.209460984245205e+02) * r + -2.759285104469687e+02) * r + 1.383577518672690e+02) * r + -3.066479806614716e+01) * r + 2.506628277459239e+00) * q /
            (((((-5.447609879822406e+01 * r + 1.615858368580409e+02) * r + -1.556989798598866e+02) * r + 6.680131188771972e+01) * r + -1.328068155288572e+01) * r + 1.0);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        double e = 0.5 * erfc(-z * 0.70710678118654752440) - p;
        double u = e * 2.5066282746310002 * exp(0.5 * z * z);
        z = z - u / (1.0 + 0.5 * z * u);
    }
    return upper ? -z : z;
}

// normal scores by doubled rank i = 2*rank: Phi^-1((i/2 - 3/8) / (n + 1/4))  (src/utils.jl:189-193,
// :182).  For Float32 data the quotient is rounded to Float32 before Phi^-1 and the score after.
__global__ void ztab_kernel(double* tab, long long n, int f32) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 2 * n) return;
    double z = __builtin_nan("");
    if (i >= 2) {
        double q = ((double)i * 0.5 - 0.375) / ((double)(4 * n + 1) / 4.0);
        if (f32) q = (double)(float)q;
        z = norminv_acc(q);
        if (f32) z = (double)(float)z;
    }
    tab[i] = z;
}

// ------------------------------------------------------------------ rank phase
// bitonic sort of one value per thread across the 1024-thread workgroup (ascending by thread id)
__device__ inline double wg_bitonic_sort(double v, double* buf) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= NT; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            double o;
            if (j >= 64) {
                buf[tid] = v;
                __syncthreads();
                o = buf[tid ^ j];
                __syncthreads();
            } else {
                o = __shfl_xor(v, j);
            }
            const bool up = (tid & k) == 0, lower = (tid & j) == 0;
            v = (lower == up) ? (o < v ? o : v) : (o > v ? o : v);
        }
    }
    return v;
}

__device__ __forceinline__ int bin_of(double v, const double* __restrict__ t, const double* __restrict__ scale) {
    if (v != v) return BIN_NAN;
    if (v < t[0]) return BIN_UNDER;
    const double top = t[K1];
    if (v >= top) return (v == top) ? BIN_TOPEQ : BIN_OVER;
    int lo = 0, hi = K1; // invariant: t[lo] <= v < t[hi]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int mid = (lo + hi) >> 1;
        if (t[mid] <= v) lo = mid; else hi = mid;
    }
    const double tk = t[lo];
    if (v == tk) return 1 + lo * FINE;
    double f = (v - tk) * scale[lo];
    f = f < (double)(FINE - 2) ? f : (double)(FINE - 2);
    return 2 + lo * FINE + (int)f;
}

__device__ __forceinline__ bool bin_reserved(int b) {
    return (b >= 1 && b <= NREG && ((b - 1) & (FINE - 1)) == 0) || b == BIN_TOPEQ;
}

// exclusive prefix over cur[0..NBIN); cur[NBIN] = total
__device__ inline void wg_exscan_bins(unsigned* cur, unsigned* wtot, unsigned total) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int PER = (NBIN + NT - 1) / NT;
    unsigned loc[PER];
    unsigned s = 0;
    const int b0 = tid * PER;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = b0 + i;
        const unsigned c = (b < NBIN) ? cur[b] : 0u;
        loc[i] = s;
        s += c;
    }
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int i = 0; i < w; ++i) woff += wtot[i];
    const unsigned base = woff + incl - s;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = b0 + i;
        if (b < NBIN) cur[b] = base + loc[i];
    }
    if (tid == 0) cur[NBIN] = total;
    __syncthreads();
}

// Pooled tied ranks of V(0..n) as doubled ranks r2 = 2*rank (exact integers) into r2out.
// tpos[0..ntgt): 0-based sorted positions whose values are captured into ostat[].
template <class VF>
__device__ void rank_phase(VF V, long long n, unsigned char* smem, unsigned* __restrict__ r2out,
                           unsigned short* __restrict__ binbuf, int ntgt, int* status) {
    const int tid = threadIdx.x;
    double* red = (double*)(smem + OFF_RED);
    double* ostat = (double*)(smem + OFF_OSTAT);
    unsigned* tpos = (unsigned*)(smem + OFF_TPOS);
    int* ints = (int*)(smem + OFF_INTS);
    unsigned* wtot = (unsigned*)(smem + OFF_WTOT);
    int* rlo = (int*)(smem + OFF_RLO);
    int* rhi = (int*)(smem + OFF_RHI);
    unsigned* rbase = (unsigned*)(smem + OFF_RBASE);
    double* spl_t = (double*)(smem + OFF_SPLT);
    double* spl_s = (double*)(smem + OFF_SPLS);
    unsigned* cur = (unsigned*)(smem + OFF_CUR);
    double* chunk = (double*)(smem + OFF_CHUNK);
    (void)red;

    // ---- sample -> sorted -> splitters
    {
        const long long items = (n + NT - 1) / NT;
        const long long j = tid % items;
        long long e = j * NT + tid;
        if (e >= n) e = (j > 0) ? (j - 1) * NT + tid : (long long)(tid % n);
        double v = V(e);
        if (v != v) v = __builtin_inf();
        v = wg_bitonic_sort(v, chunk);
        if ((tid & 3) == 0) spl_t[tid >> 2] = v;
        if (tid == NT - 1) spl_t[K1] = v;
        for (int b = tid; b < NBIN + 1; b += NT) cur[b] = 0u;
        __syncthreads();
        if (tid < K1) {
            const double w = spl_t[tid + 1] - spl_t[tid];
            double sc = (w > 0.0 && w < __builtin_inf()) ? (double)(FINE - 1) / w : 0.0;
            if (!(sc < __builtin_inf())) sc = 0.0;
            spl_s[tid] = sc;
        }
        __syncthreads();
    }
    // ---- histogram
    for (long long e = tid; e < n; e += NT) {
        const int b = bin_of(V(e), spl_t, spl_s);
        binbuf[e] = (unsigned short)b;
        atomicAdd(&cur[b], 1u);
    }
    __syncthreads();
    wg_exscan_bins(cur, wtot, (unsigned)n);
    // ---- rounds: maximal runs of consecutive bins that fit the LDS chunk
    if (tid == 0) {
        int nr = 0, b = 0;
        const unsigned nn = cur[BIN_NAN]; // number of non-NaN elements
        while (b < BIN_NAN && cur[b] < nn) {
            if (nr >= MAXROUNDS) { atomicOr(status, 1); break; }
            const unsigned base = cur[b];
            int lo = b, hi = BIN_NAN;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cur[mid] - base <= (unsigned)CH) lo = mid; else hi = mid - 1;
            }
            rbase[nr] = base;
            rlo[nr] = b;
            if (lo == b) { rhi[nr] = -1; b = b + 1; }  // a single bin larger than the chunk
            else { rhi[nr] = lo; b = lo; }
            ++nr;
        }
        ints[0] = nr;
    }
    __syncthreads();
    const int nr = ints[0];
    for (int r = 0; r < nr; ++r) {
        const int lo = rlo[r], hi = rhi[r];
        const unsigned base = rbase[r];
        if (hi >= 0) {
            // place the members of bins [lo, hi) into the chunk, bin by bin
            for (long long e = tid; e < n; e += NT) {
                const int b = binbuf[e];
                if (b >= lo && b < hi) {
                    const unsigned pos = atomicAdd(&cur[b], 1u);
                    chunk[pos - base] = V(e);
                }
            }
            __syncthreads();
            // count: rank inside the bin by all-pairs comparison
            for (long long e = tid; e < n; e += NT) {
                const int b = binbuf[e];
                if (b >= lo && b < hi) {
                    const unsigned st = (b > 0) ? cur[b - 1] : 0u;
                    const unsigned en = cur[b];
                    const double v = V(e);
                    unsigned less = st, eq;
                    if (bin_reserved(b)) {
                        eq = en - st;
                    } else {
                        eq = 0;
                        for (unsigned i = st; i < en; ++i) {
                            const double y = chunk[i - base];
                            less += (y < v) ? 1u : 0u;
                            eq += (y == v) ? 1u : 0u;
                        }
                    }
                    r2out[e] = 2u * less + eq + 1u;
                    for (int i = 0; i < ntgt; ++i)
                        if (tpos[i] - less < eq) ostat[i] = v;
                }
            }
            __syncthreads();
        } else {
            // oversize bin: all-equal by construction if reserved, brute force otherwise
            const int b = lo;
            const unsigned st = base;
            const unsigned sz = cur[b + 1] - st;
            const bool res = bin_reserved(b);
            for (long long e = tid; e < n; e += NT) {
                if (binbuf[e] == b) {
                    const double v = V(e);
                    unsigned less = st, eq = sz;
                    if (!res) {
                        eq = 0;
                        for (long long i = 0; i < n; ++i) {
                            if (binbuf[i] != b) continue;
                            const double y = V(i);
                            less += (y < v) ? 1u : 0u;
                            eq += (y == v) ? 1u : 0u;
                        }
                    }
                    r2out[e] = 2u * less + eq + 1u;
                    for (int i = 0; i < ntgt; ++i)
                        if (tpos[i] - less < eq) ostat[i] = v;
                }
            }
            __syncthreads();
            if (tid == 0) cur[b] = st + sz;
            __syncthreads();
        }
    }
    // ---- NaNs rank last, in index order, never tied
    const unsigned nn = cur[BIN_NAN];
    if (nn < (unsigned)n) {
        for (long long e = tid; e < n; e += NT) {
            if (binbuf[e] == BIN_NAN) {
                unsigned less = nn;
                for (long long i = 0; i < e; ++i) less += (binbuf[i] == BIN_NAN) ? 1u : 0u;
                r2out[e] = 2u * less + 2u;
                for (int i = 0; i < ntgt; ++i)
                    if (tpos[i] == less) ostat[i] = __builtin_nan("");
            }
        }
    }
    if (tid == 0) ints[9] = (nn < (unsigned)n) ? 1 : 0; // any NaN in this slice
    __syncthreads();
}

// ------------------------------------------------------------------ series phase
__device__ __forceinline__ long long split_row0(int k, long long niter, int rem) {
    return (long long)k * niter + (k < rem ? k : rem); // src/utils.jl:29-36
}

// Accumulate sum_t ser[t]*ser[t+kb+i], i < NL, over this wave's t-segment of [0, tlen).
// Waves are arranged as G lag groups x (NWAVE/G) t-segments.
template <int NL>
__device__ __forceinline__ void acc_lags(const double* __restrict__ ser, int tlen, int k0, int G, double (&acc)[8]) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = w % G, q = w / G, nseg = NWAVE / G;
    const int seglen = (tlen + nseg - 1) / nseg;
    const int t0 = q * seglen;
    const int t1 = (t0 + seglen < tlen) ? t0 + seglen : tlen;
    const int kb = k0 + 8 * g;
    for (int t = t0 + lane; t < t1; t += 64) {
        const double a = ser[t];
#pragma unroll
        for (int i = 0; i < NL; ++i) acc[i] = fma(a, ser[t + kb + i], acc[i]);
    }
}

struct SeriesOut {
    double rhat;
    double ess;
};

// Geyer scan state kept in LDS scal[] between lag blocks
struct Geyer {
    double pt, sum;
    long long k;
    int started, done;
};

// Y(c, row) -> value of the series for original chain c at draw `row`.
// Leaves mean_j / var_j of every split chain in LDS; returns rhat (and ess if want_acov).
template <class YF>
__device__ SeriesOut series_phase(const KP& kp, YF Y, bool want_acov, unsigned char* smem) {
    const int tid = threadIdx.x;
    double* red = (double*)(smem + OFF_RED);
    double* scal = (double*)(smem + OFF_SCAL);
    int* ints = (int*)(smem + OFF_INTS);
    double* mean_j = (double*)(smem + OFF_MEAN);
    double* var_j = (double*)(smem + OFF_VAR);
    double* ser = (double*)(smem + OFF_SER);
    double* A = (double*)(smem + OFF_LAGA);
    double* part = (double*)(smem + OFF_PART);
    double* aux = (double*)(smem + OFF_AUX);
    const long long niter = kp.niter;
    const int m = kp.m, s = kp.s;
    const long long maxlag = kp.maxlag;
    const int method = kp.method;
    const int ntile = (int)((niter + TL - 1) / TL);
    SeriesOut out;
    out.rhat = __builtin_nan("");
    out.ess = __builtin_nan("");

    for (int k = tid; k < NLAGTAB; k += NT) { A[k] = 0.0; aux[k] = 0.0; aux[NLAGTAB + k] = 0.0; }
    if (tid == 0) { ints[1] = 0; ints[2] = 0; } // geyer started / done
    __syncthreads();

    long long k0 = 0;   // first lag of the current block
    int block = 0;
    while (true) {
        // block layout: block 0 = lags [0, 16) (or [0,1) when only moments are wanted)
        int G;          // lag groups of 8
        long long kend; // one past the last lag of the block
        if (!want_acov) { G = 1; kend = 1; }
        else if (block == 0) { G = 2; kend = 16; }
        else { G = 4; kend = k0 + 32; }
        if (kend > maxlag + 1) kend = maxlag + 1;
        if (!want_acov) kend = 1;
        const int nseg = NWAVE / G;

        for (int j = 0; j < m; ++j) {
            const int c = j / s, kk = j % s;
            const long long r0 = split_row0(kk, niter, kp.rem);
            double mu;
            if (block == 0) {
                // pass 1: mean of the split chain (Statistics.mean!, src/ess_rhat.jl:529)
                double sum = 0.0;
                for (int ti = 0; ti < ntile; ++ti) {
                    const long long tau = (long long)ti * TL;
                    const int tlen = (int)((niter - tau < TL) ? niter - tau : TL);
                    double loc = 0.0;
                    for (int i = tid; i < tlen; i += NT) {
                        const double v = Y(c, r0 + tau + i);
                        if (ntile == 1) ser[i] = v;
                        loc += v;
                    }
                    sum += wg_sum(loc, red);
                }
                mu = sum / (double)niter;
                if (tid == 0) mean_j[j] = mu;
            } else {
                mu = mean_j[j];
            }
            // pass 2: centred lag products
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int halo = want_acov ? (int)k0 + 8 * G : 8; // ser[t + k] is read for k < k0 + 8G
            for (int ti = 0; ti < ntile; ++ti) {
                const long long tau = (long long)ti * TL;
                const int tlen = (int)((niter - tau < TL) ? niter - tau : TL);
                if (ntile == 1 && block == 0) {
                    for (int i = tid; i < tlen + halo; i += NT) ser[i] = (i < tlen) ? ser[i] - mu : 0.0;
                } else {
                    __syncthreads();
                    for (int i = tid; i < tlen + halo; i += NT) {
                        const long long t = tau + i;
                        ser[i] = (t < niter) ? Y(c, r0 + t) - mu : 0.0;
                    }
                }
                __syncthreads();
                if (!want_acov) acc_lags<1>(ser, tlen, (int)k0, G, acc);
                else acc_lags<8>(ser, tlen, (int)k0, G, acc);
                if (method == 2 && block == 0 && want_acov) {
                    // BDA variogram needs the squares of the first / last k centred values
                    // (src/ess_rhat.jl:204-212): sum_{t<n-k}(c_t - c_{t+k})^2 =
                    //   2*S0 - head_k - tail_k - 2*P_k, head_k = sum_{t<k} c_t^2, tail_k = sum_{t>=n-k} c_t^2
                    const int kq = tid;
                    if (kq <= maxlag) {
                        if (ti == 0) {
                            double h = 0.0;
                            for (int t = 0; t < kq && t < tlen; ++t) h = fma(ser[t], ser[t], h);
                            aux[kq] += h;
                        }
                        // tail: elements t in [niter - kq, niter) that live in this tile
                        double tl = 0.0;
                        for (long long t = niter - kq; t < niter; ++t) {
                            const long long i = t - tau;
                            if (i >= 0 && i < tlen) tl = fma(ser[i], ser[i], tl);
                        }
                        aux[NLAGTAB + kq] += tl;
                    }
                }
            }
            // reduce the wave partials in a fixed order
            {
                const int w = tid >> 6, lane = tid & 63;
                const int g = w % G, q = w / G;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double v = wave_sum(acc[i]);
                    if (lane == 0) part[(g * 8 + i) * nseg + q] = v;
                }
                __syncthreads();
                const long long kq = k0 + tid;
                if (tid < 8 * G && kq < kend) {
                    double sm = 0.0;
                    for (int q2 = 0; q2 < nseg; ++q2) sm += part[tid * nseg + q2];
                    if (kq == 0) var_j[j] = sm / (double)(niter - 1); // Statistics.var(...; mean, corrected=true)
                    A[kq] += sm;
                }
                __syncthreads();
            }
        }
        // ---- combine on one lane (src/ess_rhat.jl:537-594)
        if (tid == 0) {
            if (block == 0) {
                double W = 0.0, mb = 0.0;
                for (int j = 0; j < m; ++j) { W += var_j[j]; mb += mean_j[j]; }
                W /= (double)m;
                mb /= (double)m;
                double vb = 0.0;
                for (int j = 0; j < m; ++j) { const double d = mean_j[j] - mb; vb = fma(d, d, vb); }
                vb /= (double)(m > 1 ? m - 1 : m);
                const double varp = kp.cf * W + vb;
                scal[0] = W;
                scal[1] = varp;
                scal[2] = sqrt(varp / W);
            }
            if (want_acov) {
                const double W = scal[0], inv_varp = 1.0 / scal[1];
                const double dm = (double)m, dn = (double)niter;
                auto rho = [&](long long k) -> double {
                    double ma;
                    if (method == 2) {
                        const double n2 = (double)(niter - k);
                        const double sdiff = (2.0 * A[0] - aux[k] - aux[NLAGTAB + k] - 2.0 * A[k]) / dm;
                        ma = W - sdiff / (2.0 * n2);
                    } else {
                        ma = A[k] / dm / dn;
                    }
                    return 1.0 - inv_varp * (W - ma);
                };
                double pt = scal[4], sum = scal[5];
                long long k = (long long)scal[6];
                int done = 0;
                const long long avail = kend - 1; // largest lag available
                if (!ints[1]) {
                    const double rho_odd = rho(1);
                    pt = 1.0 + rho_odd;
                    sum = pt;
                    k = 2;
                    ints[1] = 1;
                }
                int stop = 0; // loop finished (break or bound)
                while (k < maxlag - 1) {
                    if (k + 1 > avail) break;
                    const double delta = rho(k) + rho(k + 1);
                    if (!(delta > 0.0)) { stop = 1; break; }
                    pt = jl_min(delta, pt);
                    sum += pt;
                    k += 2;
                }
                if (!(k < maxlag - 1)) stop = 1;
                if (stop && (k <= avail || !(maxlag > 1))) {
                    const double rho_even = (maxlag > 1) ? rho(k) : 0.0;
                    const double tau = jl_max(0.0, 2.0 * sum + jl_max(0.0, rho_even) - 1.0);
                    double e = jl_min(1.0 / tau, kp.rel_ess_max);
                    if (!kp.relative) e *= kp.ntotal;
                    scal[3] = e;
                    done = 1;
                }
                scal[4] = pt; scal[5] = sum; scal[6] = (double)k;
                ints[2] = done;
            } else {
                ints[2] = 1;
            }
        }
        __syncthreads();
        if (ints[2]) break;
        k0 = kend;
        ++block;
    }
    out.rhat = scal[2];
    if (want_acov) out.ess = scal[3];
    __syncthreads();
    return out;
}

// ------------------------------------------------------------------ indicator (tail ESS) phase
__device__ __forceinline__ unsigned long long bits_at(const unsigned long long* __restrict__ w, long long pos, int nw) {
    // 64 bits starting at bit `pos` of the little-endian bit stream w[0..nw)
    const long long a = pos >> 6;
    const int r = (int)(pos & 63);
    unsigned long long lo = (a < nw) ? w[a] : 0ull;
    if (r == 0) return lo;
    unsigned long long hi = (a + 1 < nw) ? w[a + 1] : 0ull;
    return (lo >> r) | (hi << (64 - r));
}

// ESS of the two indicators 1[x <= q_lo], 1[x <= q_hi] (src/ess_rhat.jl:301-311): returns
// min(ess_lo, ess_hi) with Julia's NaN-propagating min.
template <typename T>
__device__ double indicator_phase(const KP& kp, const T* __restrict__ xp, double qlo, double qhi, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* scal = (double*)(smem + OFF_SCAL);
    int* ints = (int*)(smem + OFF_INTS);
    unsigned long long* bits = (unsigned long long*)(smem + OFF_BITS);
    unsigned long long* sbits = (unsigned long long*)(smem + OFF_SBITS);
    double* icov = (double*)(smem + OFF_ICNT);
    double* imean = (double*)(smem + OFF_IVAR);
    double* idif = (double*)(smem + OFF_IDIF);
    const long long niter = kp.niter, D = kp.D, maxlag = kp.maxlag;
    const int s = kp.s, m = kp.m;
    const int nwD = (int)((D + 63) >> 6);
    const int nwS = (int)((niter + 63) >> 6);
    const T tq[2] = {(T)qlo, (T)qhi};
    const double dn = (double)niter;

    long long k0 = 0;
    for (int k = tid; k < 2 * NLAGTAB; k += NT) { icov[k] = 0.0; idif[k] = 0.0; }
    if (tid < 2) { ints[4 + tid] = 0; ints[6 + tid] = 0; } // started, done per indicator
    __syncthreads();
    while (true) {
        long long kend = k0 + 64;
        if (kend > maxlag + 1) kend = maxlag + 1;
        for (int c = 0; c < kp.C; ++c) {
            // raw ballots of chain c
            for (int wi = w; wi < nwD; wi += NWAVE) {
                const long long t = ((long long)wi << 6) + lane;
                const T v = (t < D) ? xp[(long long)c * D + t] : (T)0;
                const unsigned long long b0 = __ballot(t < D && v <= tq[0]);
                const unsigned long long b1 = __ballot(t < D && v <= tq[1]);
                if (lane == 0) { bits[wi] = b0; bits[BITW + wi] = b1; }
            }
            __syncthreads();
            for (int kk = 0; kk < s; ++kk) {
                const int j = c * s + kk;
                const long long r0 = split_row0(kk, niter, kp.rem);
                // split-aligned words, tail masked
                for (int i = tid; i < 2 * nwS; i += NT) {
                    const int ind = i / nwS, wi = i % nwS;
                    unsigned long long v = bits_at(bits + ind * BITW, r0 + ((long long)wi << 6), nwD);
                    const long long left = niter - ((long long)wi << 6);
                    if (left < 64) v &= (left <= 0) ? 0ull : ((1ull << left) - 1ull);
                    sbits[ind * BITW + wi] = v;
                }
                __syncthreads();
                // wave -> (indicator, lag); lanes own lags
                if (w < 2) {
                    const int ind = w;
                    const unsigned long long* sb = sbits + ind * BITW;
                    const long long k = k0 + lane;
                    long long Sx = 0, Ak = 0, Hk = 0, Tk = 0;
                    if (k < kend) {
                        for (int i = 0; i < nwS; ++i) {
                            const unsigned long long v = sb[i];
                            const unsigned long long vs = bits_at(sb, ((long long)i << 6) + k, nwS);
                            long long cnt = niter - (((long long)i << 6) + k); // positions t+k < niter
                            const unsigned long long ms = cnt >= 64 ? ~0ull : (cnt <= 0 ? 0ull : ((1ull << cnt) - 1ull));
                            Sx += __popcll(v);
                            Ak += __popcll(v & vs);
                            Tk += __popcll(vs);
                            Hk += __popcll(v & ms);
                        }
                        // n^2 * sum_t (I_t - mu)(I_{t+k} - mu) with mu = S/n, exact in int64
                        const long long N = niter * niter * Ak - niter * Sx * (Hk + Tk) + (niter - k) * Sx * Sx;
                        const double ck = (double)N / (dn * dn);
                        icov[ind * NLAGTAB + k] += ck;
                        // variogram numerator (BDA, src/ess_rhat.jl:204-212): the mean cancels in
                        // c_t - c_{t+k}, so sum_t (I_t - I_{t+k})^2 = #(I_t != I_{t+k}) = H + T - 2A
                        idif[ind * NLAGTAB + k] += (double)(Hk + Tk - 2 * Ak);
                        if (k == 0) imean[ind * MAXM + j] = (double)Sx / dn;
                    }
                }
                __syncthreads();
            }
        }
        // Geyer on lanes 0 (lo) and 1 (hi)
        if (tid < 2) {
            const int ind = tid;
            const double* cv = icov + ind * NLAGTAB;
            const double dm = (double)m;
            // var_j = c_{j,0}/(niter-1); W = mean_j var_j = cv[0]/m/(niter-1)
            const double W = cv[0] / dm / (double)(niter - 1);
            double mb = 0.0;
            for (int j = 0; j < m; ++j) mb += imean[ind * MAXM + j];
            mb /= dm;
            double vb = 0.0;
            for (int j = 0; j < m; ++j) { const double d = imean[ind * MAXM + j] - mb; vb = fma(d, d, vb); }
            vb /= (double)(m > 1 ? m - 1 : m);
            const double varp = kp.cf * W + vb;
            const double inv_varp = 1.0 / varp;
            const double* dv = idif + ind * NLAGTAB;
            const int bda = kp.method == 2;
            auto rho = [&](long long k) -> double {
                const double ma = bda ? W - dv[k] / dm / (2.0 * (double)(niter - k)) : cv[k] / dm / dn;
                return 1.0 - inv_varp * (W - ma);
            };
            double pt = scal[8 + 4 * ind], sum = scal[9 + 4 * ind];
            long long k = (long long)scal[10 + 4 * ind];
            const long long avail = kend - 1;
            if (!ints[6 + ind]) {
                if (!ints[4 + ind]) {
                    pt = 1.0 + rho(1);
                    sum = pt;
                    k = 2;
                    ints[4 + ind] = 1;
                }
                int stop = 0;
                while (k < maxlag - 1) {
                    if (k + 1 > avail) break;
                    const double delta = rho(k) + rho(k + 1);
                    if (!(delta > 0.0)) { stop = 1; break; }
                    pt = jl_min(delta, pt);
                    sum += pt;
                    k += 2;
                }
                if (!(k < maxlag - 1)) stop = 1;
                if (stop && (k <= avail || !(maxlag > 1))) {
                    const double rho_even = (maxlag > 1) ? rho(k) : 0.0;
                    const double tau = jl_max(0.0, 2.0 * sum + jl_max(0.0, rho_even) - 1.0);
                    double e = jl_min(1.0 / tau, kp.rel_ess_max);
                    if (!kp.relative) e *= kp.ntotal;
                    scal[11 + 4 * ind] = e;
                    ints[6 + ind] = 1;
                }
                scal[8 + 4 * ind] = pt; scal[9 + 4 * ind] = sum; scal[10 + 4 * ind] = (double)k;
            }
        }
        __syncthreads();
        if (ints[6] && ints[7]) break;
        k0 = kend;
    }
    const double r = jl_min(scal[11], scal[15]);
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------ the kernel
template <typename T>
__device__ __forceinline__ void store_out(void* p, long long i, double v) {
    if (p) ((T*)p)[i] = (T)v;
}

template <typename T>
__global__ __launch_bounds__(NT) void mdt_kernel(KP kp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    double* ostat = (double*)(smem + OFF_OSTAT);
    double* scal = (double*)(smem + OFF_SCAL);
    unsigned* tpos = (unsigned*)(smem + OFF_TPOS);
    double* red = (double*)(smem + OFF_RED);
    double* mean_j = (double*)(smem + OFF_MEAN);
    double* var_j = (double*)(smem + OFF_VAR);
    const long long n = kp.n, D = kp.D;
    unsigned* r2 = kp.r2buf + (long long)blockIdx.x * kp.n_pad;
    unsigned short* bb = kp.binbuf + (long long)blockIdx.x * kp.n_pad;
    const double* __restrict__ ztab = kp.ztab;
    const double nan = __builtin_nan("");

    for (long long p = blockIdx.x; p < kp.P; p += gridDim.x) {
        const T* __restrict__ xp = (const T*)kp.x + p * n;
        auto VX = [&](long long e) -> double { return (double)xp[e]; };
        auto YX = [&](int c, long long row) -> double { return (double)xp[(long long)c * D + row]; };
        auto YZ = [&](int c, long long row) -> double { return ztab[r2[(long long)c * D + row]]; };

        // nested R-hat combine from mean_j / var_j (src/rhat_nested.jl:152-184); lane 0 only
        auto nested_combine = [&]() -> double {
            const int K = kp.K, S = kp.S, s = kp.s;
            const int mk = K * s;
            double within = 0.0, smean_sum = 0.0;
            // superchain means are kept in scal[16..] when S is small, else recomputed
            for (int k = 0; k < S; ++k) {
                double ms = 0.0, Wk = 0.0;
                for (int i = 0; i < K; ++i) {
                    const int c = kp.chain_inds[k * K + i];
                    for (int q = 0; q < s; ++q) { ms += mean_j[c * s + q]; Wk += var_j[c * s + q]; }
                }
                ms /= (double)mk;
                Wk /= (double)mk;
                double Bk = 0.0;
                for (int i = 0; i < K; ++i) {
                    const int c = kp.chain_inds[k * K + i];
                    for (int q = 0; q < s; ++q) { const double d = mean_j[c * s + q] - ms; Bk = fma(d, d, Bk); }
                }
                Bk /= (double)(mk > 1 ? mk - 1 : mk);
                within += Wk + Bk;
                smean_sum += ms;
            }
            within /= (double)S;
            const double sm = smean_sum / (double)S;
            double between = 0.0;
            for (int k = 0; k < S; ++k) {
                double ms = 0.0;
                for (int i = 0; i < K; ++i) {
                    const int c = kp.chain_inds[k * K + i];
                    for (int q = 0; q < s; ++q) ms += mean_j[c * s + q];
                }
                ms /= (double)mk;
                const double d = ms - sm;
                between = fma(d, d, between);
            }
            between /= (double)(S - 1);
            return sqrt(1.0 + between / within);
        };

        if (kp.niter == 0) { // mean of nothing: everything is NaN (src/ess_rhat.jl:391-406 with niter = 0)
            if (tid == 0) {
                store_out<T>(kp.out_eb, p, nan); store_out<T>(kp.out_rb, p, nan);
                store_out<T>(kp.out_et, p, nan); store_out<T>(kp.out_rt, p, nan);
                store_out<T>(kp.out_rrank, p, nan);
            }
            continue;
        }

        if (kp.basic && kp.mode != 3) {
            const bool acov = kp.need_eb && kp.ess_ok;
            const SeriesOut so = series_phase(kp, YX, acov, smem);
            double ess = acov ? so.ess : nan;
            if (kp.mode == 1) {
                if (tid == 0) store_out<T>(kp.out_rb, p, nested_combine());
                __syncthreads();
            } else if (kp.mode == 2) {
                // pooled corrected std over all draws and chains (src/mcse.jl:50)
                double loc = 0.0;
                for (long long e = tid; e < n; e += NT) loc += (double)xp[e];
                const double mu = wg_sum(loc, red) / (double)n;
                loc = 0.0;
                for (long long e = tid; e < n; e += NT) { const double d = (double)xp[e] - mu; loc = fma(d, d, loc); }
                const double var = wg_sum(loc, red) / (double)(n - 1);
                if (tid == 0) store_out<T>(kp.out_eb, p, sqrt(var) / sqrt(ess));
            } else if (tid == 0) {
                store_out<T>(kp.out_eb, p, ess);
                store_out<T>(kp.out_rb, p, so.rhat);
            }
            continue;
        }

        // ---- ranking of x; capture median / quantile neighbours on the way
        const bool need_med = kp.need_rt || (kp.mode == 3 && kp.fold);
        const bool need_q = kp.need_et && kp.ess_ok;
        int ntgt = 0;
        if (tid == 0) {
            int t = 0;
            if (need_med) { tpos[t++] = (unsigned)((n - 1) / 2); tpos[t++] = (unsigned)(n / 2); }
            if (need_q) {
                for (int i = 0; i < 2; ++i) {
                    tpos[t++] = (unsigned)(kp.q_j[i] - 1);
                    tpos[t++] = (unsigned)((n == 1) ? 0 : kp.q_j[i]);
                }
            }
            ((int*)(smem + OFF_INTS))[8] = t;
        }
        __syncthreads();
        ntgt = ((int*)(smem + OFF_INTS))[8];
        const bool need_rank_x = kp.need_eb || kp.need_rb || need_med || need_q || (kp.mode == 3 && !kp.fold);
        if (need_rank_x) rank_phase(VX, n, smem, r2, bb, ntgt, kp.status);

        // order statistics -> median / quantiles, in T arithmetic like the reference
        T medT = (T)0, qloT = (T)0, qhiT = (T)0;
        const bool has_nan = need_rank_x && ((int*)(smem + OFF_INTS))[9] != 0;
        if (has_nan) { // Statistics.median -> NaN; Statistics.quantile -> ArgumentError (status bit 1)
            medT = (T)nan; qloT = (T)nan; qhiT = (T)nan;
        } else {
            int t = 0;
            if (need_med) {
                const T a = (T)ostat[0], b = (T)ostat[1];
                medT = (n & 1) ? a : (T)(a / (T)2 + b / (T)2); // Statistics.median: a/2 + b/2
                t = 2;
            }
            if (need_q) {
                T qq[2];
                for (int i = 0; i < 2; ++i) {
                    const T a = (T)ostat[t + 2 * i], b = (T)ostat[t + 2 * i + 1];
                    const T g = (T)kp.q_g[i];
                    const bool fin = (a - a == (T)0) && (b - b == (T)0); // isfinite(a) && isfinite(b)
                    qq[i] = fin ? (T)(a + (T)(g * (T)(b - a))) : (T)((T)(((T)1 - g) * a) + (T)(g * b));
                }
                qloT = qq[0];
                qhiT = qq[1];
            }
        }
        __syncthreads();

        if (kp.mode == 3) { // stage output: _rank_normalize of x or of the folded x
            if (kp.fold) {
                auto VF = [&](long long e) -> double { const T d = (T)(xp[e] - medT); return (double)(d < (T)0 ? -d : d); };
                rank_phase(VF, n, smem, r2, bb, 0, kp.status);
            }
            for (long long e = tid; e < n; e += NT) {
                const unsigned r = r2[e];
                if (kp.rank_out) kp.rank_out[p * n + e] = (double)r * 0.5;
                if (kp.z_out) ((T*)kp.z_out)[p * n + e] = (T)ztab[r];
            }
            __syncthreads();
            continue;
        }

        double eb = nan, rb = nan, et = nan, rt = nan;
        if (kp.need_eb || kp.need_rb) {
            const bool acov = kp.need_eb && kp.ess_ok;
            const SeriesOut so = series_phase(kp, YZ, acov, smem);
            rb = (kp.mode == 1) ? 0.0 : so.rhat;
            if (acov) eb = so.ess;
            if (kp.mode == 1) {
                if (tid == 0) scal[20] = nested_combine();
                __syncthreads();
                rb = scal[20];
                __syncthreads();
            }
        }
        if (kp.need_rt) {
            auto VF = [&](long long e) -> double { const T d = (T)(xp[e] - medT); return (double)(d < (T)0 ? -d : d); };
            rank_phase(VF, n, smem, r2, bb, 0, kp.status);
            const SeriesOut so = series_phase(kp, YZ, false, smem);
            rt = so.rhat;
            if (kp.mode == 1) {
                if (tid == 0) scal[20] = nested_combine();
                __syncthreads();
                rt = scal[20];
                __syncthreads();
            }
        }
        if (kp.need_et && kp.ess_ok) {
            if (qloT != qloT || qhiT != qhiT) { et = nan; if (tid == 0) atomicOr(kp.status, 2); }
            else et = indicator_phase<T>(kp, xp, (double)qloT, (double)qhiT, smem);
        }
        if (tid == 0) {
            store_out<T>(kp.out_eb, p, eb);
            store_out<T>(kp.out_rb, p, rb);
            store_out<T>(kp.out_et, p, et);
            store_out<T>(kp.out_rt, p, rt);
            store_out<T>(kp.out_rrank, p, jl_max(rt, rb)); // map(max, Rtail, Rbulk), src/ess_rhat.jl:419,622
        }
        __syncthreads();
    }
}

} // namespace mdt
