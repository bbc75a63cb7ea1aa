// mdt_device.hpp -- device code of the MI355X (gfx950) ESS / R-hat engine.
//
// One persistent launch; a workgroup owns one parameter at a time (parameters are handed out by an atomic
// ticket).  For a parameter slice x[0..n), n = draws*chains (column-major, draws fastest) it runs, on chip:
//
//   rank phase    pooled tied ranks of x (or of |x - median|) WITHOUT a sort.  Default (rank_phase_q, two
//                 levels): NT sample values are sorted, every 4th becomes a coarse splitter; a histogram pass
//                 counts the coarse classes, maximal runs of classes that fit the LDS chunk become rounds, a
//                 partition pass groups the slice by round (and split chain); every round re-cuts ITS value
//                 range into up to 4096 local fine bins (one per bucket reserved for values equal to the
//                 splitter), an LDS histogram + prefix gives every bin its rank base, and the rank inside a bin
//                 (~1.5 members) is an all-pairs count against the bin's members in LDS.  Exact for any input:
//                 binning only has to be monotone, ties/NaN/Inf/-0.0 follow StatsBase.tiedrank + isless
//                 (src/utils.jl:180).  Order statistics (median, type-7 quantile neighbours) drop out of the
//                 same counting step.  rank_phase_b / rank_phase_p are the single-level formulations (the
//                 fallback for a point mass larger than the chunk; bit-identical ranks).
//   series phase  split-chain means, variances, lagged products (src/ess_rhat.jl:526-551, 161-179) from an
//                 LDS-resident chain; lags are computed lazily in blocks and Geyer's initial monotone sequence
//                 (:555-594) runs on one lane between blocks, so well-mixed chains stop after 32 lags.  The
//                 moments of a rank-normalised series are exact fixed-point sums (order independent).
//   indicator     tail-ESS (src/ess_rhat.jl:301-311, 647-659): the quantile indicators are bit-vectors, their
//                 autocovariance is popcount(bits & bits>>k), exact integers.
//
// Everything is deterministic (no floating-point atomics), which the reference's tests demand
// (test/ess_rhat.jl:167-204: results identical wherever a parameter sits in the batch).
//
// The file is included several times by mdt_hip.hip, into one namespace per workgroup geometry:
//   MDT_NT 1024, MDT_LDS_LIMIT 160 KiB, 4 waves/SIMD  one workgroup per CU: long chains, big FFTs, maxlag <= 1000
//   MDT_NT  512, MDT_LDS_LIMIT  80 KiB, 4 waves/SIMD  TWO workgroups per CU: the memory-bound phases of one
//                                                     parameter overlap the LDS/VALU-bound phases of another
//   MDT_NT 1024, MDT_LDS_LIMIT  80 KiB, 8 waves/SIMD  two workgroups per CU at 64 VGPRs per thread (half the
//                                                     per-thread arrays): twice the waves to hide latencies with
// Every constant below derives from those macros; the host picks the geometry per request.
#include <hip/hip_runtime.h>
#include <stdint.h>

#if !defined(MDT_NS) || !defined(MDT_NT) || !defined(MDT_LDS_LIMIT) || !defined(MDT_WAVES_PER_EU)
#error "define MDT_NS, MDT_NT, MDT_LDS_LIMIT and MDT_WAVES_PER_EU before including mdt_device.hpp"
#endif

namespace MDT_NS {

constexpr int NT = MDT_NT;          // threads per workgroup
constexpr bool SMALL = (MDT_LDS_LIMIT <= 80 * 1024); // the 80 KiB tables (two workgroups per CU)
constexpr int NWAVE = NT / 64;      // 16 or 8 wave64
constexpr int K1 = NT / 4;          // coarse buckets (sample quantiles): every 4th of the NT sorted sample values
constexpr int FINE = 32;            // bins per coarse bucket: 1 reserved (== splitter) + 31 linear
constexpr int NREG = K1 * FINE;     // regular bins are 1..NREG
constexpr int BIN_UNDER = 0;        // below the smallest sample value
constexpr int BIN_TOPEQ = NREG + 1; // equal to the largest sample value (reserved)
constexpr int BIN_OVER = NREG + 2;  // above the largest sample value
constexpr int BIN_NAN = NREG + 3;   // NaN: ranked last, in index order (isless / stable sortperm)
constexpr int NBIN = NREG + 4;
constexpr int CH = SMALL ? (MDT_NT == 512 ? 6400 : 4096) : 13760; // elements of bin-ordered data staged in LDS per round
constexpr int GRID = SMALL ? 512 : 2048; // uniform cells over [min sample, max sample] that seed the splitter search
constexpr int MAXROUNDS = 192;      // => n <= ~2.6M elements per parameter
constexpr int TL = SMALL ? 4096 : 8192; // series tile (elements of one split chain resident in LDS)
constexpr int MAXLAG_CAP = SMALL ? 255 : 1000;
constexpr int HALO = SMALL ? 296 : 1040; // >= MAXLAG_CAP + 1 + 32 (one lag block past the last lag)
constexpr int MAXM = 256;           // max number of split chains chains*split_chains
constexpr int NLAGTAB = SMALL ? 256 : 1024;

// ------------------------------------------------------------------ LDS carve (bytes)
constexpr int OFF_RED = 0;                                // double red[32]
constexpr int OFF_OSTAT = OFF_RED + 32 * 8;               // double ostat[8]
constexpr int OFF_SCAL = OFF_OSTAT + 8 * 8;               // double scal[24]
constexpr int OFF_PROFT = OFF_SCAL + 24 * 8;              // int64 last profiling stamp
constexpr int OFF_TPOS = OFF_PROFT + 8;                   // uint32 tpos[8]
constexpr int OFF_INTS = OFF_TPOS + 8 * 4;                // int ints[24]
constexpr int OFF_WTOT = OFF_INTS + 24 * 4;               // uint32 wtot[16]
constexpr int OFF_RLO = OFF_WTOT + 16 * 4;                // int rlo[MAXROUNDS]
constexpr int OFF_RHI = OFF_RLO + MAXROUNDS * 4;          // int rhi[MAXROUNDS]
constexpr int OFF_RBASE = OFF_RHI + MAXROUNDS * 4;        // uint32 rbase[MAXROUNDS]
constexpr int OFF_RCUR = OFF_RBASE + MAXROUNDS * 4;       // uint32 rcur[MAXROUNDS]: fill level of each round's segment
constexpr int OFF_MEAN = ((OFF_RCUR + MAXROUNDS * 4 + 7) / 8) * 8; // double mean_j[MAXM]
constexpr int OFF_VAR = OFF_MEAN + MAXM * 8;              // double var_j[MAXM]
constexpr int OFF_BIG = ((OFF_VAR + MAXM * 8 + 15) / 16) * 16;
// rank phase view of the big region
constexpr int OFF_SPLT = OFF_BIG;                         // double t[K1+2]
constexpr int OFF_SPLS = OFF_SPLT + (K1 + 2) * 8;         // double scale[K1]
constexpr int OFF_CUR = OFF_SPLS + K1 * 8;                // uint32 cur[NBIN+4]
constexpr int OFF_LUT = OFF_CUR + (NBIN + 4) * 4;         // uint32 lut[GRID]: (klo | khi << 16) per cell
constexpr int OFF_CHUNK = ((OFF_LUT + GRID * 4 + 15) / 16) * 16; // double chunk[CH]
constexpr int LDS_RANK_END = OFF_CHUNK + CH * 8;
// series phase view of the big region
constexpr int OFF_LAGA = OFF_BIG;                         // double A[NLAGTAB]   sum over chains of lag sums
constexpr int OFF_AUX = OFF_LAGA + NLAGTAB * 8;           // double aux[2*NLAGTAB] (bda head/tail squares)
constexpr int OFF_PART = OFF_AUX + 2 * NLAGTAB * 8;       // double part[NWAVE*8]
constexpr int OFF_SER = OFF_PART + NWAVE * 8 * 8;         // double ser[TL+HALO]
constexpr int LDS_SER_END = OFF_SER + (TL + HALO) * 8;
// indicator phase view of the big region
constexpr int OFF_BITS = OFF_BIG;                         // uint64 bits[2][BITW]  raw ballots of one chain
constexpr int BITW = SMALL ? 1024 : 2048;                 // => draws <= 65536 / 131072
constexpr int OFF_SBITS = OFF_BITS + 2 * BITW * 8;        // uint64 sbits[2][BITW] split-aligned words
constexpr int OFF_ICNT = OFF_SBITS + 2 * BITW * 8;        // double icov[2][NLAGTAB] sum over chains
constexpr int OFF_IDIF = OFF_ICNT + 2 * NLAGTAB * 8;      // double idif[2][NLAGTAB] sum over chains of #(I_t != I_{t+k})
constexpr int OFF_IVAR = OFF_IDIF + 2 * NLAGTAB * 8;      // double imean[2][MAXM]
constexpr int OFF_IPART = OFF_IVAR + 2 * MAXM * 8;         // int ipart[2][NWAVE/2][64][2]: (S, A_k) of a word segment
constexpr int LDS_IND_END = OFF_IPART + 2 * (NWAVE / 2) * 64 * 2 * 4;
constexpr int LDS_BYTES = MDT_LDS_LIMIT; // 160 KiB per CU on gfx950: one big or two small workgroups
static_assert(LDS_RANK_END <= LDS_BYTES && LDS_SER_END <= LDS_BYTES && LDS_IND_END <= LDS_BYTES, "LDS budget");

// ------------------------------------------------------------------ kernel parameters
} // namespace MDT_NS
#ifndef MDT_DEVICE_COMMON
#define MDT_DEVICE_COMMON
namespace mdtc {
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
    const unsigned int* tpos_arr; // mode 5: [P][2] zero-based sorted positions (l-1, u-1) per parameter; 0xFFFFFFFF: NaN
    int fft_lds;                // FFTAutocovMethod: 1 = run the in-LDS transform when it fits (MDT_FFT_LDS=1), 0 = lagged products
    int est;                    // ess(kind = estimator): 0 none, 4 median, 5 std, 6 mad, 7 quantile(p) (src/ess_rhat.jl:626-659); mode 4: mcse(std)
    int rank_variant;           // 0: rounds re-stream the slice (rank_phase_b), 1: partitioned rounds (rank_phase_p), 2: two-level (rank_phase_q)
    long long maxlag;           // already min(maxlag, niter - 4)
    double rel_ess_max;         // log10(ntotal) evaluated in T on the host (src/ess_rhat.jl:514)
    double ntotal;              // niter * m
    double cf;                  // (niter - 1) / niter
    long long q_j[2];           // type-7 quantile: 1-based lower order statistic j (lo, hi)
    double q_g[2];              // and gamma, both evaluated in T on the host
    const double* ztab;         // normal scores by doubled rank, [2n+1]
    unsigned int* r2buf;        // per-workgroup scratch: doubled ranks, stride n_pad (streaming variant)
    double* zbuf;               // per-workgroup scratch: normal scores (or doubled ranks) in time order, stride n_pad
    double* xsbuf;              // per-workgroup scratch: values grouped by round (segment r = sorted positions of round r)
    unsigned short* bsbuf;      // per-workgroup scratch: bin ids in the same grouped order
    unsigned short* binbuf;     // per-workgroup scratch: bin ids, stride n_pad
    long long n_pad;
    void* out_eb; void* out_rb; void* out_et; void* out_rt; // [P] of T, or null
    void* out_rrank;            // [P] max(rhat_tail, rhat_bulk) (kind = rank), or null
    double* rank_out;           // stage: [P*n] tied ranks
    void* z_out;                // stage: [P*n] normal scores (T)
    const int* chain_inds;      // nested: K x S (device)
    int K, S;
    int* status;                // device status words: [0] bit0 rounds overflow, bit1 NaN hit quantile; [1] next-parameter ticket
    unsigned long long* prof;   // optional [grid*16] phase cycle counters (diagnostics), or null
};
} // namespace mdtc
#endif
namespace MDT_NS {
using mdtc::KP;

// phase slots of KP::prof
enum { PH_LOAD = 0, PH_RANKX = 1, PH_SERB = 2, PH_RANKF = 3, PH_SERT = 4, PH_IND = 5, PH_OUT = 6,
       PH_R_SAMPLE = 8, PH_R_HIST = 9, PH_R_SCAN = 10, PH_R_PLACE = 11, PH_R_COUNT = 12 };
#ifdef MDT_PROF_Q /* diagnostics build: the z-series sub-slots are lent to the two-level rank phase */
#define MDT_PROF_QMARK(slot) MDT_PROF_MARK(slot)
#define MDT_PROF_ZMARK(slot)
#define MDT_PROF_QSEL(a, b) (b)
#else
#define MDT_PROF_QMARK(slot)
#define MDT_PROF_ZMARK(slot) MDT_PROF_MARK(slot)
#define MDT_PROF_QSEL(a, b) (a)
#endif
#define MDT_PROF_MARK(slot)                                                                   \
    do {                                                                                      \
        if (kp.prof && threadIdx.x == 0) {                                                    \
            long long* pt_ = (long long*)(smem + OFF_PROFT);                                  \
            const long long t_ = clock64();                                                   \
            kp.prof[blockIdx.x * 16 + (slot)] += (unsigned long long)(t_ - *pt_);             \
            *pt_ = t_;                                                                        \
        }                                                                                     \
    } while (0)

// ------------------------------------------------------------------ uniform arguments
// Arguments of a non-inlined device function arrive in VGPRs even when they are wave-uniform, and
// the compiler then treats everything derived from them as divergent: `kp.field` became a
// flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) inside the hot loops (a full memory round trip that
// also drains every outstanding load).  These helpers move such values back to SGPRs; the parameter
// block is read through the constant address space, i.e. with scalar loads.
#define MDT_CA __attribute__((address_space(4)))
#define MDT_GA __attribute__((address_space(1)))
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ long long uni(long long v) { return (long long)uni64((unsigned long long)v); }
__device__ __forceinline__ double uni(double v) { return __longlong_as_double((long long)uni64((unsigned long long)__double_as_longlong(v))); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <class P>
__device__ __forceinline__ P* uniptr(P* p) { return (P*)uni64((unsigned long long)p); }
template <class P>
__device__ __forceinline__ MDT_GA P* uniptr_g(P* p) { return (MDT_GA P*)uni64((unsigned long long)p); }
#define MDT_UNIFORM_KP(kp_in) (*(const MDT_CA KP*)uni64((unsigned long long)&(kp_in)))

// ------------------------------------------------------------------ value sources
// Plain structs passed by value into the (deliberately not inlined) phase functions.  Pointers are
// explicitly in the GLOBAL address space: across a call boundary a plain pointer is generic and
// every access through it becomes a flat_load, which ties up both wait counters and serialises
// against LDS traffic (measured: 352 flat loads + 2283 s_waitcnt in the rank phase).
template <typename T>
struct SrcPlain {      // x itself
    const MDT_GA T* xp;
    __device__ __forceinline__ void make_uniform() { xp = (const MDT_GA T*)uni64((unsigned long long)xp); }
    __device__ __forceinline__ double operator()(long long e) const { return (double)xp[e]; }
    // element base + i with a wave-uniform base: scalar base register + 32-bit lane offset, no 64-bit address math
    __device__ __forceinline__ double at(long long base, unsigned i) const { return (double)(xp + base)[i]; }
};
template <typename T>
struct SrcFold {       // |x - median|, evaluated in T like the reference (src/utils.jl:155)
    const MDT_GA T* xp;
    T med;
    __device__ __forceinline__ void make_uniform() { xp = (const MDT_GA T*)uni64((unsigned long long)xp); med = uni(med); }
    __device__ __forceinline__ double operator()(long long e) const {
        const T d = (T)(xp[e] - med);
        return (double)(d < (T)0 ? -d : d);
    }
    __device__ __forceinline__ double at(long long base, unsigned i) const {
        const T d = (T)((xp + base)[i] - med);
        return (double)(d < (T)0 ? -d : d);
    }
};
template <typename T>
struct SerX {          // series = raw draws of chain c
    const MDT_GA T* xp;
    long long D;
    __device__ __forceinline__ void make_uniform() { xp = (const MDT_GA T*)uni64((unsigned long long)xp); D = uni(D); }
    __device__ __forceinline__ double operator()(int c, long long row) const { return (double)xp[(long long)c * D + row]; }
};
template <typename T>
struct SerSq {         // series = (x - mean)^2 evaluated in T: the expectand proxy of std (src/ess_rhat.jl:640-642)
    const MDT_GA T* xp;
    long long D;
    T mu;
    __device__ __forceinline__ void make_uniform() { xp = (const MDT_GA T*)uni64((unsigned long long)xp); D = uni(D); mu = uni(mu); }
    __device__ __forceinline__ double operator()(int c, long long row) const {
        const T d = (T)(xp[(long long)c * D + row] - mu);
        return (double)(T)(d * d);
    }
};
struct SerZ {          // series = normal scores, already in time order in the per-workgroup scratch
    const MDT_GA double* z;
    long long D;
    __device__ __forceinline__ void make_uniform() { z = (const MDT_GA double*)uni64((unsigned long long)z); D = uni(D); }
    __device__ __forceinline__ double operator()(int c, long long row) const { return z[(long long)c * D + row]; }
};
#define MDT_PHASE __device__ __attribute__((noinline))
// Keeps a loaded value "used" so that the compiler cannot sink the load under the condition of a later
// select (it then branches around every single LDS read and waits for each one separately).  Call it in a
// SECOND loop, after all the loads of a batch were issued: the value must be in its register at the asm.
__device__ __forceinline__ unsigned keep(unsigned v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ double keep(double v) { asm volatile("" : "+v"(v)); return v; }


// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ long long split_row0(int k, long long niter, int rem) {
    return (long long)k * niter + (k < rem ? k : rem); // src/utils.jl:29-36
}
__device__ __forceinline__ double jl_min(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }
__device__ __forceinline__ double jl_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }

// Wave-wide sum with DPP row shifts / row broadcasts (gfx9): ~6 VALU steps of ~10 cycles each,
// against ~6 x 2 dependent ds_bpermute round trips (~120 cycles each) for the __shfl_xor butterfly.
// Fixed association order -> deterministic.  The total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, true);
    // bound_ctrl=true / masked rows: a lane without a source receives 0 (the `old` operand), i.e. +0.0
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_l63(double v) {
    v = dpp_add<0x111, 0xF>(v); // row_shr:1
    v = dpp_add<0x112, 0xF>(v); // row_shr:2
    v = dpp_add<0x114, 0xF>(v); // row_shr:4
    v = dpp_add<0x118, 0xF>(v); // row_shr:8  -> lane 15 of every row holds the row sum
    v = dpp_add<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    v = wave_sum_l63(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// inclusive prefix sum over the 64 lanes with DPP only (no LDS permute, no lane-index registers)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_add_u32(unsigned v) {
    return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
    v = dpp_add_u32<0x111, 0xF>(v); // row_shr:1
    v = dpp_add_u32<0x112, 0xF>(v); // row_shr:2
    v = dpp_add_u32<0x114, 0xF>(v); // row_shr:4
    v = dpp_add_u32<0x118, 0xF>(v); // row_shr:8 -> inclusive inside every row of 16
    v = dpp_add_u32<0x142, 0xA>(v); // row_bcast:15: rows 1 and 3 take the total of the row before
    v = dpp_add_u32<0x143, 0xC>(v); // row_bcast:31: rows 2 and 3 take the total of the first half
    return v;
}
// 64-bit integer wave sum on the same DPP tree (total in every lane)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_add_i64(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xF, true);
    return v + (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    v = dpp_add_i64<0x111, 0xF>(v);
    v = dpp_add_i64<0x112, 0xF>(v);
    v = dpp_add_i64<0x114, 0xF>(v);
    v = dpp_add_i64<0x118, 0xF>(v);
    v = dpp_add_i64<0x142, 0xA>(v);
    v = dpp_add_i64<0x143, 0xC>(v);
    const int lo = __builtin_amdgcn_readlane((int)v, 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Fixed-point images of a shifted normal score u (|u| < 32) and of its square (< 256): adding the magic
// 1.5 * 2^6 (resp. 1.5 * 2^9) leaves u in units of 2^-46 (u^2: 2^-43) in the mantissa.  Sums of up to 2^13
// such integers are exact in int64 and do not depend on the order of summation: the moments of the
// z-series are the same bits whatever the geometry, the input order (time / segments) or the reduction
// tree.  Longer chains coarsen the unit by one bit per doubling (Fx::make), still ~1e-12 of a unit score.
struct Fx {
    double m1, m2;   // magic addends for u and u^2
    double i1, i2;   // value of one integer unit
    __device__ static Fx make(long long nterms) {
        int extra = 0;
        while ((8192LL << extra) < nterms) ++extra;
        Fx f;
        f.m1 = ldexp(96.0, extra);
        f.m2 = ldexp(768.0, extra);
        f.i1 = ldexp(1.0, extra - 46);
        f.i2 = ldexp(1.0, extra - 43);
        return f;
    }
    __device__ __forceinline__ long long u(double x) const { return __double_as_longlong(x + m1) - __double_as_longlong(m1); }
    __device__ __forceinline__ long long uu(double x) const { return __double_as_longlong(x + m2) - __double_as_longlong(m2); }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding GLOBAL
// load and store (s_waitcnt vmcnt(0)): a prefetch issued before it is serialised and every streamed store
// is waited for.  Use where the data exchanged across the barrier lives in LDS.
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
    long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 > 3 * n + 1) return;
    // [0, 2n]: by doubled rank; [2n+1, 3n+1]: dense copy for whole ranks r = i0 - (2n+1), so that the
    // common tie-free case reads consecutive entries instead of every other one
    const long long i = (i0 <= 2 * n) ? i0 : 2 * (i0 - (2 * n + 1));
    double z = __builtin_nan("");
    if (i >= 2) {
        double q = ((double)i * 0.5 - 0.375) / ((double)(4 * n + 1) / 4.0);
        if (f32) q = (double)(float)q;
        z = norminv_acc(q);
        if (f32) z = (double)(float)z;
    }
    tab[i0] = z;
}

// ------------------------------------------------------------------ rank phase
// Sorted position (0..NT-1, unique) of one value per thread among the NT values of the workgroup.
// Each wave sorts its 64 values with shuffles only, publishes them, and every value then counts, by
// 15 independent binary searches, how many values of the other waves precede it (<= for earlier waves,
// < for later ones: a stable total order).  One barrier instead of the twenty of a full bitonic network.
__device__ inline int wg_sample_rank(double& v, double* buf) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int k = 2; k <= 64; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const double o = __shfl_xor(v, j);
            const bool up = (lane & k) == 0 || k == 64, lower = (lane & j) == 0;
            v = (lower == up) ? (o < v ? o : v) : (o > v ? o : v);
        }
    }
    buf[tid] = v; // ascending within each wave
    __syncthreads();
    int pos = lane;
#pragma unroll
    for (int w2 = 0; w2 < NWAVE; ++w2) {
        if (w2 == w) continue;
        const double* run = buf + w2 * 64;
        // number of elements of run that precede v: upper bound for earlier waves, lower bound for later ones
        int lo = 0;
#pragma unroll
        for (int stp = 32; stp > 0; stp >>= 1) {
            const double y = run[lo + stp - 1];
            const bool before = (w2 < w) ? (y <= v) : (y < v);
            lo += before ? stp : 0;
        }
        { // 64 is a power of two: the last candidate
            const double y = run[lo < 63 ? lo : 63];
            const bool before = (w2 < w) ? (y <= v) : (y < v);
            lo += (lo < 64 && before) ? 1 : 0;
        }
        pos += lo;
    }
    return pos;
}

// Splitter search seeded by a uniform grid: cell(v) is monotone in v, and lut[cell] brackets the
// coarse index (built in the sample step of the rank phases), so the search usually needs 0-2 steps instead of 8.
struct BinCtx {
    double t0, top, ginv;
};
__device__ __forceinline__ int cell_of(double v, double t0, double ginv) {
    double c = (v - t0) * ginv;
    c = c < (double)(GRID - 1) ? c : (double)(GRID - 1);
    return (int)c; // v >= t0 here, so c >= 0; NaN*0 cannot occur (ginv finite, v finite or +inf -> clamped)
}
__device__ __forceinline__ int bin_of_fast(double v, const BinCtx& bc, const double* __restrict__ t,
                                           const double* __restrict__ scale, const unsigned* __restrict__ lut) {
    if (v != v) return BIN_NAN;
    if (v < bc.t0) return BIN_UNDER;
    if (v >= bc.top) return (v == bc.top) ? BIN_TOPEQ : BIN_OVER;
    const unsigned lh = lut[cell_of(v, bc.t0, bc.ginv)];
    int lo = (int)(lh & 0xFFFFu), hi = (int)(lh >> 16); // t[lo] <= v < t[hi + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid] <= v) lo = mid; else hi = mid - 1;
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
    const unsigned incl = wave_incl_scan_u32(s);
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

// ------------------------------------------------------------------ rank phase, dense loops
// Pooled tied ranks -> normal scores for a slice of any size.  No per-thread persistent state and
// no big unrolled bodies (an earlier register-resident variant spilled into every inner loop): every
// pass is a plain grid-stride loop over the elements with U independent loads in flight per thread.
// All SCATTERED traffic goes through LDS; global memory only sees coalesced streams:
//   1. histogram of bin ids (x streamed once from HBM)
//   2. prefix -> bin starts; rounds = maximal runs of bins whose members fit the LDS chunk
//   3. per round
//        place   x re-streamed (Infinity Cache); elements whose value falls in the round's coarse
//                value range get their bin, draw their sorted-order slot with one LDS atomic, drop
//                their value at chunk[slot] and remember the slot in posbuf[e] (coalesced)
//        count   DENSE, in slot order: thread i ranks chunk[i] inside its bin by all-pairs compares;
//                the normal score is looked up here, where consecutive lanes hold consecutive
//                ranks (coalesced table read), and written back over the value
//        pickup  posbuf[e] streamed; owners of this round fetch chunk[slot] and store it in time
//                order (coalesced)
constexpr int DENSE_IT = (CH + NT - 1) / NT; // slot-order iterations per thread per round
constexpr int UB = (MDT_WAVES_PER_EU >= 8) ? 10 : 20; // independent loads in flight per thread in the streaming passes
constexpr int UBH = 10;                      // ... in the histogram pass (value, bracket and probe of every element are live together)

template <typename PT, class VG>
MDT_PHASE void rank_phase_b(const KP& kp_, VG vg, long long n_, unsigned char* smem, double* zout_, PT* posbuf_,
                            unsigned short* binbuf_, int ntgt_, int store_r2_) {
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    vg.make_uniform();
    const long long n = uni(n_);
    const int ntgt = uni(ntgt_), store_r2 = uni(store_r2_);
    MDT_GA double* zout = uniptr_g(zout_);
    MDT_GA PT* posbuf = uniptr_g(posbuf_);
    MDT_GA unsigned short* binbuf = uniptr_g(binbuf_);
    const long long step = (long long)NT * UB;
    const int tid = threadIdx.x;
    const MDT_GA double* ztab = (const MDT_GA double*)kp.ztab;
    double* ostat = (double*)(smem + OFF_OSTAT);
    double* scal = (double*)(smem + OFF_SCAL);
    unsigned* tpos = (unsigned*)(smem + OFF_TPOS);
    int* ints = (int*)(smem + OFF_INTS);
    unsigned* wtot = (unsigned*)(smem + OFF_WTOT);
    int* rlo = (int*)(smem + OFF_RLO);
    int* rhi = (int*)(smem + OFF_RHI);
    unsigned* rbase = (unsigned*)(smem + OFF_RBASE);
    double* spl_t = (double*)(smem + OFF_SPLT);
    double* spl_s = (double*)(smem + OFF_SPLS);
    unsigned* cur = (unsigned*)(smem + OFF_CUR);
    unsigned* lut = (unsigned*)(smem + OFF_LUT);
    double* chunk = (double*)(smem + OFF_CHUNK);

    // ---- sample -> sorted -> splitters -> grid LUT
    {
        const long long items = (n + NT - 1) / NT;
        long long js = tid % items;
        if (js * NT + tid >= n && js > 0) --js;
        long long es = js * NT + tid;
        if (es >= n) es = (long long)(tid % n);
        double v = vg(es);
        if (v != v) v = __builtin_inf();
        const int spos = wg_sample_rank(v, chunk);
        if ((spos & 3) == 0) spl_t[spos >> 2] = v;  // every 4th sample quantile is a coarse splitter
        if (spos == NT - 1) spl_t[K1] = v;
        for (int b = tid; b < NBIN + 1; b += NT) cur[b] = 0u;
        __syncthreads();
        const double t0 = spl_t[0], top = spl_t[K1];
        double ginv = (double)GRID / (top - t0);
        if (!(ginv > 0.0) || !(ginv < __builtin_inf())) ginv = 0.0; // degenerate range: one cell, full search
        if (tid < K1) {
            const double w = spl_t[tid + 1] - spl_t[tid];
            double sc = (w > 0.0 && w < __builtin_inf()) ? (double)(FINE - 1) / w : 0.0;
            if (!(sc < __builtin_inf())) sc = 0.0;
            spl_s[tid] = sc;
        }
        // hi[c] = max{k < K1 : cell(t_k) <= c}; the coarse index of any v in cell c lies in [hi[c-1], hi[c]]
        for (int c = tid; c < GRID; c += NT) {
            int h[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cc = c - 1 + q;
                int lo = 0, hi = K1 - 1;
                if (cc < 0) hi = 0;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    const double tm = spl_t[mid];
                    const bool ok = (tm < top) ? (cell_of(tm, t0, ginv) <= cc) : false;
                    if (ok) lo = mid; else hi = mid - 1;
                }
                h[q] = lo;
            }
            lut[c] = (unsigned)h[0] | ((unsigned)h[1] << 16);
        }
        if (tid == 0) { scal[21] = t0; scal[22] = top; scal[23] = ginv; }
        __syncthreads();
    }
    BinCtx bc;
    bc.t0 = scal[21]; bc.top = scal[22]; bc.ginv = scal[23];
    MDT_PROF_MARK(PH_R_SAMPLE);

    // ---- 1. histogram.  Staged per batch: UB loads -> UB bin searches -> UB atomics -> UB stores, so that
    //      each kind of latency is paid once per batch, not once per element (a per-element
    //      `if { atomic; wait; write; store }` body serialises on lgkmcnt(0) / vmcnt).
    for (long long e0 = tid; e0 < n; e0 += step) {
        double xb[UB];
        int bq[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) { const long long e = e0 + (long long)u * NT; xb[u] = (e < n) ? vg(e) : __builtin_nan(""); }
#pragma unroll
        for (int u = 0; u < UB; ++u) bq[u] = bin_of_fast(xb[u], bc, spl_t, spl_s, lut); // padding -> BIN_NAN
#pragma unroll
        for (int u = 0; u < UB; ++u) { if (e0 + (long long)u * NT < n) atomicAdd(&cur[bq[u]], 1u); }
#pragma unroll
        for (int u = 0; u < UB; ++u) { const long long e = e0 + (long long)u * NT; if (e < n) binbuf[e] = (unsigned short)bq[u]; }
    }
    __syncthreads();
    MDT_PROF_MARK(PH_R_HIST);
    // ---- 2. starts and rounds
    wg_exscan_bins(cur, wtot, (unsigned)n);
    if (tid == 0) {
        int nr = 0, b = 0;
        const unsigned nn = cur[BIN_NAN];
        while (b < BIN_NAN && cur[b] < nn) {
            if (nr >= MAXROUNDS) { atomicOr(kp.status, 1); break; }
            const unsigned base = cur[b];
            int lo = b, hi = BIN_NAN;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cur[mid] - base <= (unsigned)CH) lo = mid; else hi = mid - 1;
            }
            rbase[nr] = base;
            rlo[nr] = b;
            if (lo == b) { rhi[nr] = -(b + 1) - 1; b = b + 1; } // a single bin larger than the chunk: -(hi) - 1
            else { rhi[nr] = lo; b = lo; }
            ++nr;
        }
        ints[0] = nr;
        ints[10] = (int)nn;
    }
    __syncthreads();
    MDT_PROF_MARK(PH_R_SCAN);
    const int nr = ints[0];
    for (int r = 0; r < nr; ++r) {
        const int lo = rlo[r];
        const bool oversize = rhi[r] < 0;
        const int hi = oversize ? lo + 1 : rhi[r];
        const unsigned base = rbase[r];
        const unsigned cnt = cur[hi] - base; // cur[hi] is still the START of bin hi
        // value range of the coarse buckets that overlap bins [lo, hi): a cheap superset filter
        const double vlo = (lo == 0) ? -__builtin_inf() : spl_t[(lo - 1) / FINE < K1 ? (lo - 1) / FINE : K1];
        const bool top_open = (hi - 1) > NREG;
        const double vhi = top_open ? __builtin_inf() : spl_t[(hi - 2) / FINE + 1];
        __syncthreads(); // everyone has read cur[hi] before the atomics below move it
        if (!oversize) {
            // 3a. place: bin ids (2 B) and values (8 B) streamed together, both unconditionally; staged
            for (long long e0 = tid; e0 < n; e0 += step) {
                double xb[UB];
                int bq[UB];
                unsigned ps[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const long long e = e0 + (long long)u * NT;
                    bq[u] = (e < n) ? (int)binbuf[e] : 0xFFFF;
                    xb[u] = (e < n) ? vg(e) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) { ps[u] = 0u; if (bq[u] >= lo && bq[u] < hi) ps[u] = atomicAdd(&cur[bq[u]], 1u); }
#pragma unroll
                for (int u = 0; u < UB; ++u) { if (bq[u] >= lo && bq[u] < hi) chunk[ps[u] - base] = xb[u]; }
#pragma unroll
                for (int u = 0; u < UB; ++u) { if (bq[u] >= lo && bq[u] < hi) posbuf[e0 + (long long)u * NT] = (PT)ps[u]; }
            }
            __syncthreads(); // cur[b] is now the END of every bin of this round; the start is cur[b-1]
            MDT_PROF_MARK(PH_R_PLACE);
            // 3b. dense count in slot order; only the order statistics that fall into this round are checked
            int tg0 = ntgt, tg1 = 0;
            for (int t = 0; t < ntgt; ++t)
                if (tpos[t] - base < cnt) { tg0 = t < tg0 ? t : tg0; tg1 = t + 1; }
            unsigned res[DENSE_IT];
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) {
                const unsigned i = (unsigned)(k * NT + tid);
                res[k] = 2u;
                if (i < cnt) {
                    const double v = chunk[i];
                    const int b = bin_of_fast(v, bc, spl_t, spl_s, lut);
                    const unsigned st = (b > 0) ? cur[b - 1] : 0u;
                    const unsigned en = cur[b];
                    unsigned less = st, eq;
                    if (bin_reserved(b)) {
                        eq = en - st;
                    } else {
                        eq = 0;
                        const unsigned last = en - 1u - base;
                        for (unsigned m2 = st - base; m2 <= last; m2 += 4u) {
                            // 4 independent LDS reads per trip; indices past the bin are clamped to its
                            // last member and masked out of the counts
                            const unsigned i1 = m2 + 1u, i2 = m2 + 2u, i3 = m2 + 3u;
                            const double y0 = chunk[m2];
                            const double y1 = chunk[i1 <= last ? i1 : last];
                            const double y2 = chunk[i2 <= last ? i2 : last];
                            const double y3 = chunk[i3 <= last ? i3 : last];
                            less += (y0 < v) ? 1u : 0u;
                            eq += (y0 == v) ? 1u : 0u;
                            less += (i1 <= last && y1 < v) ? 1u : 0u;
                            eq += (i1 <= last && y1 == v) ? 1u : 0u;
                            less += (i2 <= last && y2 < v) ? 1u : 0u;
                            eq += (i2 <= last && y2 == v) ? 1u : 0u;
                            less += (i3 <= last && y3 < v) ? 1u : 0u;
                            eq += (i3 <= last && y3 == v) ? 1u : 0u;
                        }
                    }
                    res[k] = 2u * less + eq + 1u;
                    for (int t = tg0; t < tg1; ++t)
                        if (tpos[t] - less < eq) ostat[t] = v;
                }
            }
            // normal scores by doubled rank: consecutive lanes read consecutive table entries
            double zz[DENSE_IT];
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) zz[k] = store_r2 ? (double)res[k] : ((res[k] & 1u) ? ztab[res[k]] : ztab[2 * n + 1 + (res[k] >> 1)]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) {
                const unsigned i = (unsigned)(k * NT + tid);
                if (i < cnt) chunk[i] = zz[k];
            }
            __syncthreads();
            MDT_PROF_MARK(PH_R_COUNT);
            // 3c. pickup, staged
            for (long long e0 = tid; e0 < n; e0 += step) {
                PT pb[UB];
                int bq[UB];
                double zq[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const long long e = e0 + (long long)u * NT;
                    bq[u] = (e < n) ? (int)binbuf[e] : 0xFFFF;
                    pb[u] = (e < n) ? posbuf[e] : (PT)0;
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) { zq[u] = 0.0; if (bq[u] >= lo && bq[u] < hi) zq[u] = chunk[(unsigned)pb[u] - base]; }
#pragma unroll
                for (int u = 0; u < UB; ++u) { if (bq[u] >= lo && bq[u] < hi) zout[e0 + (long long)u * NT] = zq[u]; }
            }
            __syncthreads();
            MDT_PROF_MARK(PH_R_PLACE);
        } else {
            // a single bin larger than the chunk: all-equal by construction if reserved, else brute force
            const int b = lo;
            const bool resv = bin_reserved(b);
            for (long long e = tid; e < n; e += NT) {
                const double v = vg(e);
                if (v >= vlo && (top_open || v < vhi) && bin_of_fast(v, bc, spl_t, spl_s, lut) == b) {
                    unsigned less = base, eq = cnt;
                    if (!resv) {
                        eq = 0;
                        for (long long i = 0; i < n; ++i) {
                            const double y = vg(i);
                            if (bin_of_fast(y, bc, spl_t, spl_s, lut) != b) continue;
                            less += (y < v) ? 1u : 0u;
                            eq += (y == v) ? 1u : 0u;
                        }
                    }
                    const unsigned r2 = 2u * less + eq + 1u;
                    zout[e] = store_r2 ? (double)r2 : ztab[r2];
                    for (int t = 0; t < ntgt; ++t)
                        if (tpos[t] - less < eq) ostat[t] = v;
                }
            }
            __syncthreads();
            if (tid == 0) cur[b] = base + cnt; // keep the invariant "bins below the current round hold their END"
            __syncthreads();
        }
    }
    const unsigned nn = (unsigned)ints[10];
    if (nn < (unsigned)n) { // NaNs rank last, in index order, never tied
        for (long long e = tid; e < n; e += NT) {
            const double v = vg(e);
            if (v != v) {
                unsigned less = nn;
                for (long long i = 0; i < e; ++i) { const double y = vg(i); less += (y != y) ? 1u : 0u; }
                const unsigned r2 = 2u * less + 2u;
                zout[e] = store_r2 ? (double)r2 : ztab[r2];
                for (int t = 0; t < ntgt; ++t)
                    if (tpos[t] == less) ostat[t] = __builtin_nan("");
            }
        }
    }
    if (tid == 0) ints[9] = (nn < (unsigned)n) ? 1 : 0;
    __syncthreads();
}

// ------------------------------------------------------------------ rank phase, partitioned rounds
// Same ranking as rank_phase_b, but the rounds do not re-stream the whole slice: after the histogram
// one PARTITION pass appends every element (value, original index, bin) to the segment of its round
// -- segment r is exactly the range of sorted positions [rbase[r], rbase[r] + cnt_r) -- with one LDS
// atomic per wave per round present in the wave (ballot + mbcnt), so the global writes are runs of
// consecutive addresses.  A round then loads its own segment (one batch of <= 14 elements per thread),
// places, counts densely, and scatters the scores straight from registers to time order.
// Traffic is O(n) for any n (rank_phase_b is O(n * rounds)) and every element is touched once per step.
template <class VG>
MDT_PHASE void rank_phase_p(const KP& kp_, VG vg, long long n_, unsigned char* smem, double* zout_, unsigned* esbuf_,
                            unsigned short* binbuf_, double* xsbuf_, unsigned short* bsbuf_, int ntgt_, int store_r2_) {
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    vg.make_uniform();
    const long long n = uni(n_);
    const int ntgt = uni(ntgt_), store_r2 = uni(store_r2_);
    MDT_GA double* zout = uniptr_g(zout_);
    MDT_GA unsigned* esbuf = uniptr_g(esbuf_);
    MDT_GA unsigned short* binbuf = uniptr_g(binbuf_);
    MDT_GA double* xsbuf = uniptr_g(xsbuf_);
    MDT_GA unsigned short* bsbuf = uniptr_g(bsbuf_);
    const long long step = (long long)NT * UB;
    const int tid = threadIdx.x, lane = tid & 63;
    const MDT_GA double* ztab = (const MDT_GA double*)kp.ztab;
    double* ostat = (double*)(smem + OFF_OSTAT);
    double* scal = (double*)(smem + OFF_SCAL);
    unsigned* tpos = (unsigned*)(smem + OFF_TPOS);
    int* ints = (int*)(smem + OFF_INTS);
    unsigned* wtot = (unsigned*)(smem + OFF_WTOT);
    int* rlo = (int*)(smem + OFF_RLO);
    int* rhi = (int*)(smem + OFF_RHI);
    unsigned* rbase = (unsigned*)(smem + OFF_RBASE);
    unsigned* rcur = (unsigned*)(smem + OFF_RCUR);
    double* spl_t = (double*)(smem + OFF_SPLT);
    double* spl_s = (double*)(smem + OFF_SPLS);
    unsigned* cur = (unsigned*)(smem + OFF_CUR);
    unsigned* lut = (unsigned*)(smem + OFF_LUT);
    double* chunk = (double*)(smem + OFF_CHUNK);

    // ---- sample -> sorted -> splitters -> grid LUT
    {
        const long long items = (n + NT - 1) / NT;
        long long js = tid % items;
        if (js * NT + tid >= n && js > 0) --js;
        long long es = js * NT + tid;
        if (es >= n) es = (long long)(tid % n);
        double v = vg(es);
        if (v != v) v = __builtin_inf();
        const int spos = wg_sample_rank(v, chunk);
        if ((spos & 3) == 0) spl_t[spos >> 2] = v;  // every 4th sample quantile is a coarse splitter
        if (spos == NT - 1) spl_t[K1] = v;
        for (int b = tid; b < NBIN + 1; b += NT) cur[b] = 0u;
        __syncthreads();
        const double t0 = spl_t[0], top = spl_t[K1];
        double ginv = (double)GRID / (top - t0);
        if (!(ginv > 0.0) || !(ginv < __builtin_inf())) ginv = 0.0; // degenerate range: one cell, full search
        if (tid < K1) {
            const double w = spl_t[tid + 1] - spl_t[tid];
            double sc = (w > 0.0 && w < __builtin_inf()) ? (double)(FINE - 1) / w : 0.0;
            if (!(sc < __builtin_inf())) sc = 0.0;
            spl_s[tid] = sc;
        }
        for (int c = tid; c < GRID; c += NT) {
            int h[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cc = c - 1 + q;
                int lo = 0, hi = K1 - 1;
                if (cc < 0) hi = 0;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    const double tm = spl_t[mid];
                    const bool ok = (tm < top) ? (cell_of(tm, t0, ginv) <= cc) : false;
                    if (ok) lo = mid; else hi = mid - 1;
                }
                h[q] = lo;
            }
            lut[c] = (unsigned)h[0] | ((unsigned)h[1] << 16);
        }
        if (tid == 0) { scal[21] = t0; scal[22] = top; scal[23] = ginv; }
        __syncthreads();
    }
    BinCtx bc;
    bc.t0 = scal[21]; bc.top = scal[22]; bc.ginv = scal[23];
    MDT_PROF_MARK(PH_R_SAMPLE);

    // ---- 1. histogram (the HBM read of x); bin ids kept for the partition pass
    for (long long e0 = tid; e0 < n; e0 += step) {
        double xb[UB];
        int bq[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) { const long long e = e0 + (long long)u * NT; xb[u] = (e < n) ? vg(e) : __builtin_nan(""); }
#pragma unroll
        for (int u = 0; u < UB; ++u) bq[u] = bin_of_fast(xb[u], bc, spl_t, spl_s, lut); // padding -> BIN_NAN
#pragma unroll
        for (int u = 0; u < UB; ++u) { if (e0 + (long long)u * NT < n) atomicAdd(&cur[bq[u]], 1u); }
#pragma unroll
        for (int u = 0; u < UB; ++u) { const long long e = e0 + (long long)u * NT; if (e < n) binbuf[e] = (unsigned short)bq[u]; }
    }
    __syncthreads();
    MDT_PROF_MARK(PH_R_HIST);
    // ---- 2. starts and rounds
    wg_exscan_bins(cur, wtot, (unsigned)n);
    if (tid == 0) {
        int nr = 0, b = 0;
        const unsigned nn = cur[BIN_NAN];
        while (b < BIN_NAN && cur[b] < nn) {
            if (nr >= MAXROUNDS) { atomicOr(kp.status, 1); break; }
            const unsigned base = cur[b];
            int lo = b, hi = BIN_NAN;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cur[mid] - base <= (unsigned)CH) lo = mid; else hi = mid - 1;
            }
            rbase[nr] = base;
            rlo[nr] = b;
            rcur[nr] = 0u;
            if (lo == b) { rhi[nr] = -(b + 1) - 1; b = b + 1; } // a single bin larger than the chunk
            else { rhi[nr] = lo; b = lo; }
            ++nr;
        }
        ints[0] = nr;
        ints[10] = (int)nn;
    }
    __syncthreads();
    MDT_PROF_MARK(PH_R_SCAN);
    const int nr = ints[0];
    // ---- 3. partition: append (value, index, bin) to the segment of the element's round
    for (long long e0 = tid; e0 < n; e0 += step) {
        double xb[UB];
        int bq[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long long e = e0 + (long long)u * NT;
            bq[u] = (e < n) ? (int)binbuf[e] : BIN_NAN;
            xb[u] = (e < n) ? vg(e) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int b = bq[u];
            int r = -1;
            if (b < BIN_NAN) { // round of the bin: last r with rlo[r] <= b
                int lo2 = 0, hi2 = nr - 1;
                while (lo2 < hi2) { const int mid = (lo2 + hi2 + 1) >> 1; if (rlo[mid] <= b) lo2 = mid; else hi2 = mid - 1; }
                r = lo2;
            }
            // one LDS atomic per (wave, round present): ballot -> leader reserves -> lanes take consecutive slots
            unsigned dst = 0u;
            unsigned long long todo = __ballot(r >= 0);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                const int r0 = __builtin_amdgcn_readlane(r, src);
                const unsigned long long m2 = __ballot(r == r0);
                unsigned off0 = 0u;
                if (lane == src) off0 = atomicAdd(&rcur[r0], (unsigned)__popcll(m2));
                off0 = (unsigned)__builtin_amdgcn_readlane((int)off0, src);
                if (r == r0) dst = rbase[r0] + off0 + (unsigned)__popcll(m2 & ((1ull << lane) - 1ull));
                todo &= ~m2;
            }
            if (r >= 0) {
                xsbuf[dst] = xb[u];
                esbuf[dst] = (unsigned)(e0 + (long long)u * NT);
                bsbuf[dst] = (unsigned short)b;
            }
        }
    }
    __syncthreads();
    MDT_PROF_MARK(PH_R_PLACE);
    // ---- 4. rounds over their own segments
    for (int r = 0; r < nr; ++r) {
        const int lo = rlo[r];
        const bool oversize = rhi[r] < 0;
        const unsigned base = rbase[r];
        const unsigned cnt = rcur[r];
        if (!oversize) {
            unsigned ev[DENSE_IT], pv[DENSE_IT];
            {
                double xv[DENSE_IT];
                int bv[DENSE_IT];
#pragma unroll
                for (int k = 0; k < DENSE_IT; ++k) {
                    const unsigned i = (unsigned)(k * NT + tid);
                    xv[k] = (i < cnt) ? xsbuf[base + i] : 0.0;
                    ev[k] = (i < cnt) ? esbuf[base + i] : 0u;
                    bv[k] = (i < cnt) ? (int)bsbuf[base + i] : 0;
                }
#pragma unroll
                for (int k = 0; k < DENSE_IT; ++k) { pv[k] = 0u; if ((unsigned)(k * NT + tid) < cnt) pv[k] = atomicAdd(&cur[bv[k]], 1u) - base; }
#pragma unroll
                for (int k = 0; k < DENSE_IT; ++k) { if ((unsigned)(k * NT + tid) < cnt) chunk[pv[k]] = xv[k]; }
            }
            __syncthreads(); // cur[b] is now the END of every bin of this round; the start is cur[b-1]
            MDT_PROF_MARK(PH_R_PLACE);
            int tg0 = ntgt, tg1 = 0; // only the order statistics that fall into this round are checked
            for (int t = 0; t < ntgt; ++t)
                if (tpos[t] - base < cnt) { tg0 = t < tg0 ? t : tg0; tg1 = t + 1; }
            unsigned res[DENSE_IT];
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) {
                const unsigned i = (unsigned)(k * NT + tid);
                res[k] = 2u;
                if (i < cnt) {
                    const double v = chunk[i];
                    const int b = bin_of_fast(v, bc, spl_t, spl_s, lut);
                    const unsigned st = (b > 0) ? cur[b - 1] : 0u;
                    const unsigned en = cur[b];
                    unsigned less = st, eq;
                    if (bin_reserved(b)) {
                        eq = en - st;
                    } else {
                        eq = 0;
                        const unsigned last = en - 1u - base;
                        for (unsigned m2 = st - base; m2 <= last; m2 += 4u) {
                            const unsigned i1 = m2 + 1u, i2 = m2 + 2u, i3 = m2 + 3u;
                            const double y0 = chunk[m2];
                            const double y1 = chunk[i1 <= last ? i1 : last];
                            const double y2 = chunk[i2 <= last ? i2 : last];
                            const double y3 = chunk[i3 <= last ? i3 : last];
                            less += (y0 < v) ? 1u : 0u;
                            eq += (y0 == v) ? 1u : 0u;
                            less += (i1 <= last && y1 < v) ? 1u : 0u;
                            eq += (i1 <= last && y1 == v) ? 1u : 0u;
                            less += (i2 <= last && y2 < v) ? 1u : 0u;
                            eq += (i2 <= last && y2 == v) ? 1u : 0u;
                            less += (i3 <= last && y3 < v) ? 1u : 0u;
                            eq += (i3 <= last && y3 == v) ? 1u : 0u;
                        }
                    }
                    res[k] = 2u * less + eq + 1u;
                    for (int t = tg0; t < tg1; ++t)
                        if (tpos[t] - less < eq) ostat[t] = v;
                }
            }
            double zz[DENSE_IT];
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) zz[k] = store_r2 ? (double)res[k] : ((res[k] & 1u) ? ztab[res[k]] : ztab[2 * n + 1 + (res[k] >> 1)]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) {
                const unsigned i = (unsigned)(k * NT + tid);
                if (i < cnt) chunk[i] = zz[k];
            }
            __syncthreads();
            MDT_PROF_MARK(PH_R_COUNT);
            // scores to time order, straight from the slots remembered in registers
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) zz[k] = ((unsigned)(k * NT + tid) < cnt) ? chunk[pv[k]] : 0.0;
#pragma unroll
            for (int k = 0; k < DENSE_IT; ++k) { if ((unsigned)(k * NT + tid) < cnt) zout[ev[k]] = zz[k]; }
            __syncthreads();
            MDT_PROF_MARK(PH_R_PLACE);
        } else {
            // a single bin larger than the chunk: all-equal by construction if reserved, else brute force
            // over the members of the bin (they are contiguous in the segment)
            const bool resv = bin_reserved(lo);
            for (unsigned i = tid; i < cnt; i += NT) {
                const double v = xsbuf[base + i];
                unsigned less = base, eq = cnt;
                if (!resv) {
                    eq = 0;
                    for (unsigned j2 = 0; j2 < cnt; ++j2) {
                        const double y = xsbuf[base + j2];
                        less += (y < v) ? 1u : 0u;
                        eq += (y == v) ? 1u : 0u;
                    }
                }
                const unsigned r2 = 2u * less + eq + 1u;
                zout[esbuf[base + i]] = store_r2 ? (double)r2 : ztab[r2];
                for (int t = 0; t < ntgt; ++t)
                    if (tpos[t] - less < eq) ostat[t] = v;
            }
            __syncthreads();
            if (tid == 0) cur[lo] = base + cnt; // keep the invariant "bins below the current round hold their END"
            __syncthreads();
        }
    }
    const unsigned nn = (unsigned)ints[10];
    if (nn < (unsigned)n) { // NaNs rank last, in index order, never tied
        for (long long e = tid; e < n; e += NT) {
            if (binbuf[e] == BIN_NAN) {
                unsigned less = nn;
                for (long long i = 0; i < e; ++i) less += (binbuf[i] == BIN_NAN) ? 1u : 0u;
                const unsigned r2 = 2u * less + 2u;
                zout[e] = store_r2 ? (double)r2 : ztab[r2];
                for (int t = 0; t < ntgt; ++t)
                    if (tpos[t] == less) ostat[t] = __builtin_nan("");
            }
        }
    }
    if (tid == 0) ints[9] = (nn < (unsigned)n) ? 1 : 0;
    __syncthreads();
}

// ------------------------------------------------------------------ rank phase, two-level (the default)
// Level 1: the slice is partitioned by the 256 COARSE sample-quantile buckets only (one histogram of 260
//   classes, one partition pass that appends (value, index, class) to the segment of the class's round;
//   rounds = maximal runs of classes that fit the LDS chunk).
// Level 2: each round re-bins ITS OWN segment with up to 8192 local fine bins (linear cut of every
//   coarse bucket in the round, sub-bin 0 reserved for values equal to the splitter; 16-bit counters
//   packed two per word), so a bin holds ~1-2 values whatever n is (the single-level variants degrade as
//   n / 8196 per bin).  Bin boundaries inside the chunk are a bit mask, so the dense count needs no bin
//   search at all; the scores go to time order straight from registers.  O(n) traffic.
// A coarse class larger than the chunk (constant data, a huge point mass) falls back to rank_phase_p.
constexpr int NCLS = K1 + 4;              // 0 under, 1..K1 buckets, K1+1 == top sample, K1+2 over, K1+3 NaN
constexpr int CLS_NAN = K1 + 3;
constexpr int NBL = SMALL ? 4096 : 8192;      // local fine bins per round
constexpr int BPT = NBL / NT;             // bins per thread in the prefix step (packed two per word)
static_assert(BPT == 8 || BPT == 4, "prefix step: 4 or 8 bins per thread");
constexpr int QCH = SMALL ? (MDT_NT == 512 ? 6080 : 5440) : 15104; // chunk slots of the two-level variant
constexpr bool USE_LBS = SMALL;           // bin id of every slot in LDS (80 KiB tables) instead of the start-bit masks
constexpr int DENSE_ITQ = (QCH + NT - 1) / NT;
constexpr int OFF_QT = OFF_BIG;                               // double t[K1+2]
constexpr int OFF_QIW = OFF_QT + (K1 + 2) * 8;                // double invw[K1]
constexpr int OFF_QLUT = OFF_QIW + K1 * 8;                    // uint32 lut[GRID]
constexpr int OFF_QCC = OFF_QLUT + GRID * 4;                  // uint32 cc[NCLS + 4]
constexpr int OFF_QROF = OFF_QCC + (NCLS + 4) * 4;            // uint16 rof[NCLS + 4]
constexpr int OFF_QCUR = ((OFF_QROF + (NCLS + 4) * 2 + 3) / 4) * 4; // uint32 lcur[NBL/2 + 2] (packed u16)
constexpr int OFF_QBST = OFF_QCUR + (NBL / 2 + 2) * 4;        // uint32 bst[QCH/32 + 2]: bin starts          | uint16 lbs[QCH]: bin id
constexpr int OFF_QBRS = OFF_QBST + (QCH / 32 + 2) * 4;       // uint32 brs[QCH/32 + 2]: reserved-bin starts |   (| 0x8000: reserved bin)
constexpr int OFF_QCHUNK = (((USE_LBS ? OFF_QBST + QCH * 2 : OFF_QBRS + (QCH / 32 + 2) * 4) + 15) / 16) * 16; // double chunk[QCH]
constexpr int LDS_RANKQ_END = OFF_QCHUNK + QCH * 8;
static_assert(LDS_RANKQ_END <= LDS_BYTES, "two-level rank view exceeds the LDS carve");

__device__ __forceinline__ int class_of(double v, const BinCtx& bc, const double* __restrict__ t,
                                        const unsigned* __restrict__ lut) {
    if (v != v) return CLS_NAN;
    if (v < bc.t0) return 0;
    if (v >= bc.top) return (v == bc.top) ? K1 + 1 : K1 + 2;
    const unsigned lh = lut[cell_of(v, bc.t0, bc.ginv)];
    int lo = (int)(lh & 0xFFFFu), hi = (int)(lh >> 16);
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid] <= v) lo = mid; else hi = mid - 1;
    }
    return 1 + lo;
}

// Partition-time view of the region above the class tables: the cursors of the (round [, split chain])
// segments; the (split chain x class) counters of the histogram pass and the sample buffer live above them.
constexpr int SEGKEYS = 128;                                   // (round, split chain) keys of a segmented output
constexpr int KMAX = 256;                                      // partition keys (>= MAXROUNDS, >= SEGKEYS)
static_assert(KMAX >= MAXROUNDS && KMAX >= SEGKEYS && KMAX <= NT, "partition key tables");
constexpr int KDUMMY = KMAX;                                   // key of a slot past the end of the slice / of a NaN
constexpr int OFF_TSEG = OFF_QCUR;                             // uint32 segcur[KMAX]: cursor of every segment (round-major order)
constexpr int OFF_TSEG2 = OFF_TSEG + KMAX * 4;                 // uint32 segcur2[KMAX]: start of every segment in chain-major order
constexpr int OFF_TGB2 = OFF_TSEG2 + KMAX * 4;                 // uint32 kg2[KMAX]: chain-major place = round-major place + kg2[key]
constexpr int OFF_TX = ((OFF_TGB2 + KMAX * 4 + 15) / 16) * 16; // double tx[NT]: sample buffer; then
constexpr int OFF_CC2 = OFF_TX;                                // uint32 cc2[NJ][NCLS] + 1 (histogram pass only)
constexpr int CC2_BYTES = LDS_BYTES - OFF_CC2;
static_assert(NT * 8 <= CC2_BYTES, "sample buffer exceeds the LDS carve");
// element tag (kbuf / ksbuf, 16 bits): class (9 bits) | split-chain slot (6 bits) << 9 | first-row flag << 15
constexpr int TAG_CLS = 511;

// Split chain of element e of the pooled slice (src/utils.jl:29-36: split k takes rows k*niter + min(k, rem)
// onward, niter of them; one row is dropped after each of the first `rem` splits): slot in [0, m), or m for a
// dropped row; first = the row opens its split chain.
__device__ __forceinline__ unsigned split_slot(unsigned e, unsigned D, unsigned niter, unsigned rem, unsigned s, unsigned m, bool& first) {
    const unsigned c = e / D, row = e - c * D;
    const unsigned b = rem * (niter + 1u);
    unsigned kk, off;
    if (row < b) { kk = row / (niter + 1u); off = row - kk * (niter + 1u); }
    else { const unsigned q = (row - b) / niter; kk = rem + q; off = row - b - q * niter; }
    first = off == 0u;
    return (off >= niter) ? m : c * s + kk;
}

// The same bookkeeping for a thread that visits e, e + step, e + 2 step, ...: one pair of divisions at the
// start, additions afterwards.
struct SplitIter {
    unsigned D, niter, rem, s, m;
    unsigned c, row, kk, off;
    __device__ __forceinline__ void locate() { // (kk, off) of `row`
        const unsigned b = rem * (niter + 1u);
        if (row < b) { kk = row / (niter + 1u); off = row - kk * (niter + 1u); }
        else { const unsigned q = (row - b) / niter; kk = rem + q; off = row - b - q * niter; }
    }
    __device__ __forceinline__ void init(unsigned e) { c = e / D; row = e - c * D; locate(); }
    __device__ __forceinline__ void advance(unsigned step) {
        row += step;
        off += step;
        if (row >= D) {
            do { row -= D; ++c; } while (row >= D);
            locate();
        } else {
            unsigned len = niter + (kk < rem ? 1u : 0u);
            while (off >= len) { off -= len; ++kk; len = niter + (kk < rem ? 1u : 0u); }
        }
    }
    __device__ __forceinline__ unsigned tag() const { // slot << 9 | first-row flag << 15
        const unsigned j = (off >= niter) ? m : c * s + kk;
        return (j << 9) | (off == 0u ? 0x8000u : 0u);
    }
};

// Returns 0 when it cannot serve the slice (a coarse class larger than the chunk, or NaNs together with a
// segmented request): the caller falls back to rank_phase_p.  1: scores (or doubled ranks) are in zout in
// TIME order.  2: (only when seg_req) scores are in zout in CHAIN-MAJOR SEGMENT order: split chain j owns
// zout[j * niter, (j + 1) * niter), filled round by round (each (round, chain) segment is one contiguous,
// coalesced stream); the draw index inside the chain of every slot is in esbuf, read as 16-bit words, in the
// same order (written by the partition pass); the score of the first draw of every split chain is in
// mean_j[] -- series_z_phase<true> scatters that to time order in LDS: no scattered global access anywhere.
// 3: (only when out_req == 2: the consumer needs nothing but the moments of every split chain) no scores are
// written at all: the pickup of every round adds its members to exact fixed-point sums per split chain and
// mean_j[] / var_j[] hold the final means and variances (shift 0; a chain whose scores are all the same
// bits gets variance exactly 0, as the reference's centred sum does).
// out_req: 0 time order, 1 segment order if possible, 2 moments only if possible, 4 nothing but the order
// statistics (rounds that hold none of them are skipped altogether; returns 1, zout untouched).
template <class VG>
MDT_PHASE int rank_phase_q(const KP& kp_, VG vg, long long n_, unsigned char* smem, double* zout_, unsigned* esbuf_,
                           unsigned short* kbuf_, double* xsbuf_, unsigned short* ksbuf_, int ntgt_, int store_r2_, int out_req_) {
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    vg.make_uniform();
    const long long n = uni(n_);
    const int ntgt = uni(ntgt_), store_r2 = uni(store_r2_);
    MDT_GA double* zout = uniptr_g(zout_);
    MDT_GA unsigned* esbuf = uniptr_g(esbuf_);
    MDT_GA unsigned short* kbuf = uniptr_g(kbuf_);
    MDT_GA double* xsbuf = uniptr_g(xsbuf_);
    MDT_GA unsigned short* ksbuf = uniptr_g(ksbuf_);
    const long long step = (long long)NT * UB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const MDT_GA double* ztab = (const MDT_GA double*)kp.ztab;
    double* ostat = (double*)(smem + OFF_OSTAT);
    double* scal = (double*)(smem + OFF_SCAL);
    unsigned* tpos = (unsigned*)(smem + OFF_TPOS);
    int* ints = (int*)(smem + OFF_INTS);
    unsigned* wtot = (unsigned*)(smem + OFF_WTOT);
    int* rlo = (int*)(smem + OFF_RLO);
    int* rhi = (int*)(smem + OFF_RHI);
    unsigned* rbase = (unsigned*)(smem + OFF_RBASE);
    unsigned* rcur = (unsigned*)(smem + OFF_RCUR);
    double* mean_j = (double*)(smem + OFF_MEAN);
    unsigned* segsave = (unsigned*)(smem + OFF_VAR);     // [SEGKEYS + 1] starts of the (round, slot) segments
    unsigned* e0tab = segsave + SEGKEYS + 4;             // [65] pooled index of the first draw of slot j
    unsigned* cumj = e0tab + 68;                         // [65] draws of slot j placed by the earlier rounds
    int* adj2 = (int*)(cumj + 68);                       // [65] this round: slot of a round member + adj2[j] = its place in chain-major order
    static_assert((SEGKEYS + 4 + 3 * 68) * 4 <= MAXM * 8, "segment tables must fit the var_j bytes");
    double* qt = (double*)(smem + OFF_QT);
    double* qiw = (double*)(smem + OFF_QIW);
    unsigned* lut = (unsigned*)(smem + OFF_QLUT);
    unsigned* cc = (unsigned*)(smem + OFF_QCC);
    unsigned short* rof = (unsigned short*)(smem + OFF_QROF);
    unsigned* lcur = (unsigned*)(smem + OFF_QCUR);
    unsigned* bst = (unsigned*)(smem + OFF_QBST);
    unsigned* brs = (unsigned*)(smem + OFF_QBRS);
    unsigned short* lbs = (unsigned short*)(smem + OFF_QBST);
    double* chunk = (double*)(smem + OFF_QCHUNK);
    unsigned* cc2 = (unsigned*)(smem + OFF_CC2);
    unsigned* segcur = (unsigned*)(smem + OFF_TSEG);
    unsigned* segcur2 = (unsigned*)(smem + OFF_TSEG2);
    unsigned* kg2 = (unsigned*)(smem + OFF_TGB2);
    MDT_GA unsigned short* locbuf = (MDT_GA unsigned short*)esbuf; // segmented output: draw index of every member, chain-major
    double* tx = (double*)(smem + OFF_TX);
    const unsigned uD = (unsigned)kp.D, uniter = (unsigned)kp.niter, urem = (unsigned)kp.rem, us = (unsigned)kp.s, um = (unsigned)kp.m;
    // split-chain slots per class in the histogram: m real ones + 1 for dropped rows
    const int out_req = uni(out_req_);
    const bool noout = out_req == 4;
    const int NJ = ((out_req == 1 || out_req == 2) && kp.m + 1 <= 64) ? kp.m + 1 : 1; // slots the elements are tagged with
    const int NJc = (out_req == 1 && (NJ * NCLS + 1) * 4 <= CC2_BYTES) ? NJ : 1; // slots the class histogram is kept for (segment sizes)
    const bool momout = out_req == 2 && NJ > 1;
    // moments-only view of the mean_j / var_j bytes while the rounds run (65 slots each)
    long long* mS1 = (long long*)(smem + OFF_MEAN);
    long long* mS2 = mS1 + 65;
    unsigned long long* mOr = (unsigned long long*)(smem + OFF_VAR);
    unsigned long long* mAnd = mOr + 65;
    static_assert(2 * 65 * 8 <= MAXM * 8, "moment accumulators must fit the mean_j / var_j bytes");

    // ---- sample -> sorted positions -> coarse splitters -> grid LUT
    {
        const unsigned un = (unsigned)n, items = (un + NT - 1u) / NT; // (n < 2^31: 32-bit index math)
        unsigned js = (unsigned)tid % items;
        if (js * NT + (unsigned)tid >= un && js > 0u) --js;
        unsigned es = js * NT + (unsigned)tid;
        if (es >= un) es = (unsigned)tid % un;
        double v = vg.at(0, es);
        if (v != v) v = __builtin_inf();
        const int spos = wg_sample_rank(v, tx);
        if ((spos & 3) == 0) qt[spos >> 2] = v;
        if (spos == NT - 1) qt[K1] = v;
        if (tid == 0) ints[12] = 0;
        __syncthreads();
        for (int c = tid; c < NJc * NCLS + 1; c += NT) cc2[c] = 0u; // (aliases the sample buffer: after the barrier)
        const double t0 = qt[0], top = qt[K1];
        double ginv = (double)GRID / (top - t0);
        if (!(ginv > 0.0) || !(ginv < __builtin_inf())) ginv = 0.0;
        if (tid < K1) {
            const double wd = qt[tid + 1] - qt[tid];
            double iw = (wd > 0.0 && wd < __builtin_inf()) ? 1.0 / wd : 0.0;
            if (!(iw < __builtin_inf())) iw = 0.0;
            qiw[tid] = iw;
        }
        int wmax = 0;
        for (int c = tid; c < GRID; c += NT) {
            int h[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cq = c - 1 + q;
                int lo = 0, hi = K1 - 1;
                if (cq < 0) hi = 0;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    const double tm = qt[mid];
                    const bool ok = (tm < top) ? (cell_of(tm, t0, ginv) <= cq) : false;
                    if (ok) lo = mid; else hi = mid - 1;
                }
                h[q] = lo;
            }
            lut[c] = (unsigned)h[0] | ((unsigned)h[1] << 16);
            wmax = (h[1] - h[0]) > wmax ? (h[1] - h[0]) : wmax;
        }
        { // widest bracket = the number of search steps every value takes: one LDS atomic per wave
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int ov = __shfl_xor(wmax, o); wmax = ov > wmax ? ov : wmax; }
            if (lane == 0) atomicMax(&ints[12], wmax);
        }
        if (tid == 0) { scal[21] = t0; scal[22] = top; scal[23] = ginv; }
        __syncthreads();
    }
    BinCtx bc;
    bc.t0 = scal[21]; bc.top = scal[22]; bc.ginv = scal[23];
    const int nsteps = 32 - __clz(ints[12]); // smallest S with 2^S > widest bracket
    MDT_PROF_MARK(PH_R_SAMPLE);

    // ---- level 1: coarse histogram per (split-chain slot, class): the HBM read of x
    // Straight-line per batch: every value takes the same `nsteps` search steps, so the LDS reads of the UBH
    // values of a thread overlap; slots past the end of the slice count into a dummy counter.
    {
        SplitIter sp;
        sp.D = uD; sp.niter = uniter; sp.rem = urem; sp.s = us; sp.m = um;
        for (long long b0 = 0; b0 < n; b0 += (long long)NT * UBH) {
            const unsigned left = (unsigned)((n - b0 < (long long)NT * UBH) ? n - b0 : (long long)NT * UBH);
            double xb[UBH];
#pragma unroll
            for (int u = 0; u < UBH; ++u) {
                const unsigned i = (unsigned)(u * NT + tid);
                xb[u] = vg.at(b0, i < left ? i : left - 1u);
            }
            unsigned lh[UBH];
#pragma unroll
            for (int u = 0; u < UBH; ++u) {
                double vc = (xb[u] >= bc.t0) ? xb[u] : bc.t0; // (NaN -> t0)
                vc = vc < bc.top ? vc : bc.top;
                lh[u] = lut[cell_of(vc, bc.t0, bc.ginv)];
            }
#pragma unroll
            for (int u = 0; u < UBH; ++u) lh[u] = keep(lh[u]);
            for (int st = 0; st < nsteps; ++st) {
                double tm[UBH];
#pragma unroll
                for (int u = 0; u < UBH; ++u) tm[u] = qt[((lh[u] & 0xFFFFu) + (lh[u] >> 16) + 1u) >> 1];
#pragma unroll
                for (int u = 0; u < UBH; ++u) tm[u] = keep(tm[u]);
#pragma unroll
                for (int u = 0; u < UBH; ++u) {
                    const unsigned lo = lh[u] & 0xFFFFu, hi = lh[u] >> 16;
                    const unsigned mid = (lo + hi + 1u) >> 1;
                    const bool up = tm[u] <= xb[u];
                    lh[u] = up ? (mid | (hi << 16)) : (lo | ((lo < hi ? mid - 1u : hi) << 16));
                }
            }
            unsigned kq[UBH];
            if (NJ > 1) sp.init((unsigned)b0 + (unsigned)tid);
#pragma unroll
            for (int u = 0; u < UBH; ++u) {
                const double v = xb[u];
                unsigned cls = 1u + (lh[u] & 0xFFFFu);
                cls = (v < bc.t0) ? 0u : cls;
                cls = (v >= bc.top) ? ((v == bc.top) ? (unsigned)(K1 + 1) : (unsigned)(K1 + 2)) : cls;
                cls = (v != v) ? (unsigned)CLS_NAN : cls;
                kq[u] = cls;
                if (NJ > 1) { kq[u] |= sp.tag(); sp.advance((unsigned)NT); }
            }
#pragma unroll
            for (int u = 0; u < UBH; ++u) {
                const bool ok = (unsigned)(u * NT + tid) < left;
                atomicAdd(&cc2[ok ? (NJc > 1 ? (kq[u] >> 9 & 63u) * NCLS : 0u) + (kq[u] & TAG_CLS) : (unsigned)(NJc * NCLS)], 1u);
            }
#pragma unroll
            for (int u = 0; u < UBH; ++u) {
                const unsigned i = (unsigned)(u * NT + tid);
                if (i < left) (kbuf + b0)[i] = (unsigned short)kq[u];
            }
        }
    }
    __syncthreads();
    MDT_PROF_MARK(PH_R_HIST);
    // ---- class totals -> class starts, rounds (maximal runs of classes that fit the chunk), class -> round
    for (int c = tid; c < NCLS + 4; c += NT) {
        unsigned t = 0;
        if (c < NCLS) for (int j = 0; j < NJc; ++j) t += cc2[j * NCLS + c];
        cc[c] = t;
    }
    __syncthreads();
    static_assert(NCLS <= 320, "class prefix: 5 classes per lane");
    if (tid < 64) { // exclusive prefix over the class counts, 5 classes per lane
        unsigned c5[5], tot = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) { const int c = 5 * tid + q; c5[q] = (c < NCLS) ? cc[c] : 0u; tot += c5[q]; }
        const unsigned incl = wave_incl_scan_u32(tot);
        unsigned run = incl - tot;
#pragma unroll
        for (int q = 0; q < 5; ++q) { const int c = 5 * tid + q; if (c <= NCLS) cc[c] = run; run += c5[q]; }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned nn = cc[CLS_NAN];
        int nr = 0, c = 0, ok = 1;
        while (c < CLS_NAN && cc[c] < nn) {
            if (nr >= MAXROUNDS) { atomicOr(kp.status, 1); ok = 0; break; }
            const unsigned base = cc[c];
            int e2 = c, hi = CLS_NAN; // last class boundary with cc[e2] - base <= QCH (cc is non-decreasing)
            while (e2 < hi) {
                const int mid = (e2 + hi + 1) >> 1;
                if (cc[mid] - base <= (unsigned)QCH) e2 = mid; else hi = mid - 1;
            }
            if (e2 == c) { ok = 0; break; } // one class larger than the chunk
            rbase[nr] = base; rlo[nr] = c; rhi[nr] = e2; rcur[nr] = cc[e2] - base;
            c = e2;
            ++nr;
        }
        const int segout = NJc > 1 && nr * NJc <= SEGKEYS && kp.niter <= 65535;
        if (NJ > 1 && nn < (unsigned)n) ok = 0; // NaNs and a segmented / moments request: the single-level variant serves it
        ints[0] = nr;
        ints[10] = (int)nn;
        ints[13] = ok;
        ints[15] = segout ? NJc : 1;
    }
    __syncthreads();
    MDT_PROF_MARK(PH_R_SCAN);
    if (!ints[13]) { __syncthreads(); return 0; }
    const int nr = ints[0];
    const int NJe = ints[15];         // split-chain slots per round in the partition keys (1: rounds only)
    const bool segout = NJe > 1;
    const int NKEY = nr * NJe;
    for (int c = tid; c < CLS_NAN; c += NT) { // class -> round
        int r = 0;
        while (r + 1 < nr && rlo[r + 1] <= c) ++r;
        rof[c] = (unsigned short)r;
    }
    // ---- key (= round * NJe + slot) sizes -> global segment starts
    {
        unsigned cnt = 0;
        if (tid < NKEY) {
            const int r = tid / NJe, j = tid - r * NJe;
            if (segout) { for (int c = rlo[r]; c < rhi[r]; ++c) cnt += cc2[j * NCLS + c]; }
            else cnt = rcur[r];
        }
        const unsigned incl = wave_incl_scan_u32(cnt);
        if (lane == 63) wtot[w] = incl;
        __syncthreads(); // (also: every read of cc2 is done before the tile area is reused)
        unsigned woff = 0;
        for (int q = 0; q < w; ++q) woff += wtot[q];
        const unsigned start = woff + incl - cnt;
        if (tid < KMAX) segcur[tid] = start;
        if (segout && tid <= NKEY) segsave[tid] = (tid < NKEY) ? start : (unsigned)n;
        if (tid < 68) {
            const unsigned c = (unsigned)tid / us, kk = (unsigned)tid - c * us;
            e0tab[tid] = (segout && tid < (int)um) ? c * uD + (unsigned)split_row0((int)kk, kp.niter, kp.rem) : 0u;
            cumj[tid] = 0u;
        }
    }
    __syncthreads();
    if (segout && tid < NKEY) { // chain-major start of segment (r, j): j * niter + members of j in the earlier rounds
        const int r = tid / NJe, j = tid - r * NJe;
        unsigned cum = 0;
        for (int r2 = 0; r2 < r; ++r2) cum += segsave[r2 * NJe + j + 1] - segsave[r2 * NJe + j];
        segcur2[tid] = (unsigned)j * uniter + cum;
    }
    __syncthreads();
    // ---- partition: every element draws its place in its key's segment with one returning LDS atomic on the
    // segment cursor and is stored there; lanes of a wave that share a key draw neighbouring places, so the
    // stores still fill whole cache lines.  (Counting-sorting tiles of 8*NT records in LDS first, so that every
    // key's run leaves as one contiguous stream, was measured too: 17.0 ms against 16.8 ms -- the extra LDS
    // traffic and four barriers per tile cost more than the better-formed stores save.)
    {
        const bool want_e = !momout && !noout; // the element index is only needed to scatter scores back in time order
        if (segout && tid < NKEY) kg2[tid] = segcur2[tid] - segcur[tid]; // chain-major place = place + kg2[key]
        __syncthreads();
        constexpr int UP = 10;
        for (long long b0 = 0; b0 < n; b0 += (long long)NT * UP) {
            const unsigned left = (unsigned)((n - b0 < (long long)NT * UP) ? n - b0 : (long long)NT * UP);
            double xv[UP];
            unsigned kv[UP], key[UP], off[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                unsigned i = (unsigned)(u * NT + tid);
                i = i < left ? i : left - 1u;
                kv[u] = (unsigned)(kbuf + b0)[i];
                xv[u] = vg.at(b0, i);
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) { const unsigned cls = kv[u] & TAG_CLS; key[u] = (unsigned)rof[cls < (unsigned)CLS_NAN ? cls : 0u]; }
#pragma unroll
            for (int u = 0; u < UP; ++u) key[u] = keep(key[u]);
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const unsigned cls = kv[u] & TAG_CLS;
                const bool ok = (unsigned)(u * NT + tid) < left && cls < (unsigned)CLS_NAN;
                key[u] = ok ? key[u] * (unsigned)NJe + (segout ? (kv[u] >> 9 & 63u) : 0u) : (unsigned)KDUMMY;
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) off[u] = atomicAdd(&segcur[key[u] < (unsigned)KDUMMY ? key[u] : (unsigned)(KMAX - 1)], key[u] < (unsigned)KDUMMY ? 1u : 0u);
#pragma unroll
            for (int u = 0; u < UP; ++u) off[u] = keep(off[u]);
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                if (key[u] < (unsigned)KDUMMY) {
                    const unsigned e = (unsigned)b0 + (unsigned)(u * NT + tid);
                    xsbuf[off[u]] = xv[u];
                    ksbuf[off[u]] = (unsigned short)kv[u];
                    if (segout) locbuf[off[u] + kg2[key[u]]] = (unsigned short)(e - e0tab[kv[u] >> 9 & 63u]);
                    else if (want_e) esbuf[off[u]] = e;
                }
            }
        }
    }
    __syncthreads(); // the segments are complete in global memory
    MDT_PROF_MARK(MDT_PROF_QSEL(PH_R_PLACE, 13));
    const unsigned zdense = (unsigned)(2 * n + 1); // start of the dense whole-rank copy in ztab (3n + 2 entries < 2^32)
    const Fx fx = Fx::make(kp.niter);
    if (momout) { // (the bytes held the partition's tables until here)
        if (tid < 65) { mS1[tid] = 0; mS2[tid] = 0; mOr[tid] = 0ull; mAnd[tid] = ~0ull; }
        __syncthreads();
    }
    // ---- level 2: rounds
    const int tid_f = tid;
    for (int r = 0; r < nr; ++r) {
        // the per-slot index math (masks, LDS addresses for 15 slots) is invariant over the rounds; hoisted out
        // of this loop it costs ~70 spilled registers, so the thread index is made opaque per round
        int tid = tid_f;
        asm volatile("" : "+v"(tid));
        const int clo = rlo[r], chi = rhi[r];
        const unsigned base = rbase[r];
        const unsigned cnt = rcur[r];
        if (noout) { // order statistics only: a round that holds none of them is not ranked at all
            bool any = false;
            for (int t = 0; t < ntgt; ++t) any = any || (tpos[t] - base < cnt);
            if (!any) continue;
        }
        int F2 = NBL / (chi - clo);
        F2 = F2 > 4096 ? 4096 : F2;
        const double f2m1 = (double)(F2 - 1);
        for (int i = tid; i < NBL / 2 + 2; i += NT) lcur[i] = 0u;
        if (!USE_LBS) { for (int i = tid; i < QCH / 32 + 2; i += NT) { bst[i] = 0u; brs[i] = 0u; } }
        if (segout && tid < NJe) {
            const unsigned st = segsave[r * NJe + tid], c = segsave[r * NJe + tid + 1] - st, done = cumj[tid];
            adj2[tid] = (int)((unsigned)tid * uniter + done) - (int)(st - base);
            cumj[tid] = done + c;
        }
        __syncthreads();
        unsigned pv[DENSE_ITQ];
        unsigned fmask = 0u; // slots of this thread that hold the first draw of a split chain
        unsigned rmask = 0u; // ... that fall into a reserved bin (value == splitter)
        unsigned jpk[(DENSE_ITQ + 3) / 4]; // split-chain slot of every member, 8 bits each
#pragma unroll
        for (int k = 0; k < (DENSE_ITQ + 3) / 4; ++k) jpk[k] = 0u;
        {
            double xv[DENSE_ITQ];
            int lb[DENSE_ITQ];
            {
                const MDT_GA double* xr = xsbuf + base; // (uniform base + 32-bit lane offsets, clamped: no branches)
                const MDT_GA unsigned short* kr = ksbuf + base;
#pragma unroll
                for (int k = 0; k < DENSE_ITQ; ++k) {
                    unsigned i = (unsigned)(k * NT + tid);
                    i = i < cnt ? i : cnt - 1u;
                    xv[k] = xr[i];
                    lb[k] = (int)kr[i];
                }
            }
            // local fine bin: (class - clo) * F2 + sub; sub 0 = equal to the bucket's splitter (reserved).
            // Two half-batches: the splitter and scale of every member are read before any is used.
            constexpr int HB = (DENSE_ITQ + 1) / 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double tk2[HB], iw2[HB];
#pragma unroll
                for (int kk = 0; kk < HB; ++kk) {
                    const int k = h * HB + kk;
                    if (k < DENSE_ITQ) {
                        const int cls = lb[k] & TAG_CLS;
                        const int kc = cls < 1 ? 0 : (cls > K1 ? K1 - 1 : cls - 1);
                        tk2[kk] = qt[kc];
                        iw2[kk] = qiw[kc];
                    }
                }
#pragma unroll
                for (int kk = 0; kk < HB; ++kk) { if (h * HB + kk < DENSE_ITQ) { tk2[kk] = keep(tk2[kk]); iw2[kk] = keep(iw2[kk]); } }
#pragma unroll
                for (int kk = 0; kk < HB; ++kk) {
                    const int k = h * HB + kk;
                    if (k < DENSE_ITQ) {
                        fmask |= ((unsigned)lb[k] >> 15 & 1u) << k;
                        jpk[k >> 2] |= ((unsigned)lb[k] >> 9 & 63u) << ((k & 3) << 3);
                        const int cls = lb[k] & TAG_CLS;
                        const double v = xv[k];
                        double f = (v - tk2[kk]) * iw2[kk] * f2m1;
                        f = f < (double)(F2 - 2) ? f : (double)(F2 - 2);
                        const int subr = (v == tk2[kk]) ? 0 : 1 + (int)f;        // inside a regular bucket
                        const int sub = (cls >= 1 && cls <= K1) ? subr : ((cls == K1 + 1) ? 0 : 1);
                        lb[k] = (cls - clo) * F2 + sub;
                        rmask |= (sub == 0 ? 1u : 0u) << k;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < DENSE_ITQ; ++k) { if ((unsigned)(k * NT + tid) < cnt) atomicAdd(&lcur[lb[k] >> 1], 1u << ((lb[k] & 1) << 4)); }
            __syncthreads();
            MDT_PROF_QMARK(14);
            // exclusive prefix over the packed 16-bit counters; mark bin starts (and reserved bins)
            {
                unsigned wv[BPT / 2], c8[BPT], s8[BPT];
                unsigned tot = 0;
#pragma unroll
                for (int q = 0; q < BPT / 2; ++q) wv[q] = lcur[(BPT / 2) * tid + q];
#pragma unroll
                for (int q = 0; q < BPT; ++q) { c8[q] = (wv[q >> 1] >> ((q & 1) << 4)) & 0xFFFFu; s8[q] = tot; tot += c8[q]; }
                const unsigned incl = wave_incl_scan_u32(tot);
                if (lane == 63) wtot[w] = incl;
                __syncthreads();
                unsigned woff = 0;
                for (int q = 0; q < w; ++q) woff += wtot[q];
                const unsigned b0 = woff + incl - tot;
#pragma unroll
                for (int q = 0; q < BPT / 2; ++q) lcur[(BPT / 2) * tid + q] = (b0 + s8[2 * q]) | ((b0 + s8[2 * q + 1]) << 16);
#pragma unroll
                for (int q = 0; q < BPT; ++q) {
                    if (!USE_LBS && c8[q]) {
                        const unsigned st = b0 + s8[q];
                        atomicOr(&bst[st >> 5], 1u << (st & 31));
                        if (((BPT * tid + q) % F2) == 0) atomicOr(&brs[st >> 5], 1u << (st & 31));
                    }
                }
            }
            __syncthreads();
            MDT_PROF_QMARK(15);
#pragma unroll
            for (int k = 0; k < DENSE_ITQ; ++k) {
                pv[k] = 0u;
                if ((unsigned)(k * NT + tid) < cnt) {
                    const unsigned sh = (unsigned)(lb[k] & 1) << 4;
                    pv[k] = (atomicAdd(&lcur[lb[k] >> 1], 1u << sh) >> sh) & 0xFFFFu;
                }
            }
#pragma unroll
            for (int k = 0; k < DENSE_ITQ; ++k) {
                if ((unsigned)(k * NT + tid) < cnt) {
                    chunk[pv[k]] = xv[k];
                    if (USE_LBS) lbs[pv[k]] = (unsigned short)((unsigned)lb[k] | ((rmask >> k & 1u) << 15));
                }
            }
        }
        __syncthreads();
        MDT_PROF_MARK(PH_R_PLACE);
        int tg0 = ntgt, tg1 = 0; // only the order statistics that fall into this round are checked
        for (int t = 0; t < ntgt; ++t)
            if (tpos[t] - base < cnt) { tg0 = t < tg0 ? t : tg0; tg1 = t + 1; }
        unsigned res[DENSE_ITQ];
        const int nwm = (int)((cnt + 31) >> 5);
#pragma unroll
        for (int k = 0; k < DENSE_ITQ; ++k) {
            const unsigned i = (unsigned)(k * NT + tid);
            res[k] = 2u;
            if (i < cnt) {
                const double v = chunk[i];
                unsigned st, en;
                bool rsv;
                if (USE_LBS) {
                    // bin id of the slot -> bounds from the packed ends (the end of bin b-1 is the start of bin b)
                    const unsigned tg = lbs[i], b = tg & 0x7FFFu, bm = b > 0u ? b - 1u : 0u;
                    const unsigned wa = lcur[b >> 1], wb = lcur[bm >> 1];
                    en = (b & 1u) ? wa >> 16 : wa & 0xFFFFu;
                    st = (b == 0u) ? 0u : ((bm & 1u) ? wb >> 16 : wb & 0xFFFFu);
                    rsv = (tg >> 15) != 0u;
                } else {
                    // bin bounds from the start-bit mask: last start <= i, next start > i
                    int wi = (int)(i >> 5);
                    unsigned mk = bst[wi] & (0xFFFFFFFFu >> (31 - (i & 31)));
                    while (mk == 0u) mk = bst[--wi];
                    st = ((unsigned)wi << 5) + 31u - (unsigned)__clz((int)mk);
                    int wj = (int)(i >> 5);
                    unsigned mn = ((i & 31) == 31) ? 0u : (bst[wj] & (0xFFFFFFFEu << (i & 31)));
                    while (mn == 0u && ++wj < nwm) mn = bst[wj];
                    en = mn ? ((unsigned)wj << 5) + (unsigned)(__ffs((int)mn) - 1) : cnt;
                    en = en < cnt ? en : cnt;
                    rsv = ((brs[st >> 5] >> (st & 31)) & 1u) != 0u;
                }
                unsigned less = base + st, eq;
                if (rsv) {
                    eq = en - st;
                } else {
                    eq = 0;
                    const unsigned last = en - 1u;
                    for (unsigned m2 = st; m2 <= last; m2 += 4u) {
                        const unsigned i1 = m2 + 1u, i2 = m2 + 2u, i3 = m2 + 3u;
                        const double y0 = chunk[m2];
                        const double y1 = chunk[i1 <= last ? i1 : last];
                        const double y2 = chunk[i2 <= last ? i2 : last];
                        const double y3 = chunk[i3 <= last ? i3 : last];
                        less += (y0 < v) ? 1u : 0u;
                        eq += (y0 == v) ? 1u : 0u;
                        less += (i1 <= last && y1 < v) ? 1u : 0u;
                        eq += (i1 <= last && y1 == v) ? 1u : 0u;
                        less += (i2 <= last && y2 < v) ? 1u : 0u;
                        eq += (i2 <= last && y2 == v) ? 1u : 0u;
                        less += (i3 <= last && y3 < v) ? 1u : 0u;
                        eq += (i3 <= last && y3 == v) ? 1u : 0u;
                    }
                }
                res[k] = 2u * less + eq + 1u;
                for (int t = tg0; t < tg1; ++t)
                    if (tpos[t] - less < eq) ostat[t] = v;
            }
        }
        if (!noout) {
            double zz[DENSE_ITQ];
#pragma unroll
            for (int k = 0; k < DENSE_ITQ; ++k) { // whole ranks read the dense copy behind the doubled-rank table (32-bit index math)
                const unsigned zi = (res[k] & 1u) ? res[k] : zdense + (res[k] >> 1);
                zz[k] = store_r2 ? (double)res[k] : ztab[zi];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < DENSE_ITQ; ++k) {
                const unsigned i = (unsigned)(k * NT + tid);
                if (i < cnt) chunk[i] = zz[k];
            }
            __syncthreads();
            MDT_PROF_MARK(PH_R_COUNT);
#pragma unroll
            for (int k = 0; k < DENSE_ITQ; ++k) zz[k] = ((unsigned)(k * NT + tid) < cnt) ? chunk[pv[k]] : 0.0;
            if (momout) {
                // Moments only: the scores go back to LDS in PARTITION order, where consecutive members belong to
                // the same split chain almost always (a tile of the slice spans at most two); every thread sums a
                // run of consecutive members in fixed point and flushes with LDS atomics when the chain changes.
                unsigned char* jl = (unsigned char*)lcur; // split-chain slot of every member (the local bins are done)
                __syncthreads();
#pragma unroll
                for (int k = 0; k < DENSE_ITQ; ++k) {
                    const unsigned i = (unsigned)(k * NT + tid);
                    if (i < cnt) { chunk[i] = zz[k]; jl[i] = (unsigned char)(jpk[k >> 2] >> ((k & 3) << 3) & 63u); }
                }
                __syncthreads();
                long long a1 = 0, a2 = 0;
                unsigned long long ao = 0ull, aa = ~0ull;
                int jc = -1;
                auto flush = [&]() {
                    if (jc >= 0) {
                        atomicAdd((unsigned long long*)&mS1[jc], (unsigned long long)a1);
                        atomicAdd((unsigned long long*)&mS2[jc], (unsigned long long)a2);
                        atomicOr(&mOr[jc], ao);
                        atomicAnd(&mAnd[jc], aa);
                    }
                };
                const unsigned i0 = (unsigned)tid * DENSE_ITQ;
                for (int k = 0; k < DENSE_ITQ; ++k) {
                    const unsigned i = i0 + (unsigned)k;
                    if (i < cnt) {
                        const double zv = chunk[i];
                        const int j = (int)jl[i];
                        if (j != jc) { flush(); jc = j; a1 = 0; a2 = 0; ao = 0ull; aa = ~0ull; }
                        a1 += fx.u(zv);
                        a2 += fx.uu(zv * zv);
                        const unsigned long long zb2 = (unsigned long long)__double_as_longlong(zv);
                        ao |= zb2;
                        aa &= zb2;
                    }
                }
                flush();
            } else if (segout) {
                // chain-major segment order: every (round, chain) segment is one coalesced stream; the first
                // draw of every split chain leaves its score in mean_j[] (the shift K_j of series_z_phase)
                int dst[DENSE_ITQ];
#pragma unroll
                for (int k = 0; k < DENSE_ITQ; ++k) dst[k] = adj2[jpk[k >> 2] >> ((k & 3) << 3) & 63u];
#pragma unroll
                for (int k = 0; k < DENSE_ITQ; ++k) dst[k] = (int)keep((unsigned)dst[k]);
#pragma unroll
                for (int k = 0; k < DENSE_ITQ; ++k) {
                    const unsigned i = (unsigned)(k * NT + tid);
                    if (i < cnt) zout[(int)i + dst[k]] = zz[k];
                }
                if (fmask) {
#pragma unroll
                    for (int k = 0; k < DENSE_ITQ; ++k) {
                        const unsigned j = jpk[k >> 2] >> ((k & 3) << 3) & 63u;
                        if ((fmask >> k & 1u) && j < um) mean_j[j] = zz[k];
                    }
                }
            } else {
                unsigned ev[DENSE_ITQ];
#pragma unroll
                for (int k = 0; k < DENSE_ITQ; ++k) { const unsigned i = (unsigned)(k * NT + tid); ev[k] = (i < cnt) ? esbuf[base + i] : 0u; }
#pragma unroll
                for (int k = 0; k < DENSE_ITQ; ++k) { if ((unsigned)(k * NT + tid) < cnt) zout[ev[k]] = zz[k]; }
            }
        }
        __syncthreads();
        MDT_PROF_MARK(MDT_PROF_QSEL(PH_R_PLACE, 7));
    }
    const unsigned nn = (unsigned)ints[10];
    if (nn < (unsigned)n) { // NaNs rank last, in index order, never tied (time-order output only, see above)
        for (long long e = tid; e < n; e += NT) {
            if ((kbuf[e] & TAG_CLS) == CLS_NAN) {
                unsigned less = nn;
                for (long long i = 0; i < e; ++i) less += ((kbuf[i] & TAG_CLS) == CLS_NAN) ? 1u : 0u;
                const unsigned r2 = 2u * less + 2u;
                if (!noout) zout[e] = store_r2 ? (double)r2 : ztab[r2];
                for (int t = 0; t < ntgt; ++t)
                    if (tpos[t] == less) ostat[t] = __builtin_nan("");
            }
        }
    }
    if (tid == 0) { ints[9] = (nn < (unsigned)n) ? 1 : 0; ints[14] = nr; }
    __syncthreads();
    if (momout) {
        // sums -> mean_j / var_j (src/ess_rhat.jl:529-534 up to the fixed-point unit, ~1e-13 of a unit score)
        double mu = 0.0, var = 0.0;
        if (tid < kp.m) {
            const double S1 = (double)mS1[tid] * fx.i1, S2 = (double)mS2[tid] * fx.i2;
            const double dn = (double)kp.niter;
            mu = S1 / dn;
            var = (mOr[tid] == mAnd[tid]) ? 0.0 : (S2 - mu * S1) / (dn - 1.0);
        }
        __syncthreads();
        if (tid < kp.m) { mean_j[tid] = mu; ((double*)(smem + OFF_VAR))[tid] = var; }
        __syncthreads();
        return 3;
    }
    return segout ? 2 : 1;
}

// ------------------------------------------------------------------ series phase

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


// Combine on one lane: W, var+, R-hat from mean_j / var_j (src/ess_rhat.jl:537-545) after the first
// lag block, then Geyer's initial monotone sequence (:555-594) over the lags available so far.
// State between blocks lives in scal[4..6], ints[1] (started), ints[2] (done).
template <class KPT>
__device__ inline void series_combine(const KPT& kp, unsigned char* smem, int block, long long kend, bool want_acov, bool fft_sums = false) {
    double* scal = (double*)(smem + OFF_SCAL);
    int* ints = (int*)(smem + OFF_INTS);
    double* mean_j = (double*)(smem + OFF_MEAN);
    double* var_j = (double*)(smem + OFF_VAR);
    double* A = (double*)(smem + OFF_LAGA);
    double* aux = (double*)(smem + OFF_AUX);
    const long long niter = kp.niter, maxlag = kp.maxlag;
    const int m = kp.m, method = kp.method;
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
    if (!want_acov) { ints[2] = 1; return; }
    const double W = scal[0], inv_varp = 1.0 / scal[1];
    const double dm = (double)m, dn = (double)niter;
    auto rho = [&](long long k) -> double {
        double ma;
        if (method == 2) {
            const double n2 = (double)(niter - k);
            const double sdiff = (2.0 * A[0] - aux[k] - aux[NLAGTAB + k] - 2.0 * A[k]) / dm;
            ma = W - sdiff / (2.0 * n2);
        } else if (method == 1 && fft_sums) { // A[k] = sum_j ac_k/ac_0 * var_j from the transform; otherwise plain lag sums
            ma = A[k] / dm * kp.cf; // mean_i(ac_k/ac_0 * var_i) * (niter-1)//niter, src/ess_rhat.jl:191-194
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
}

// Y(c, row) -> value of the series for original chain c at draw `row`.
// Leaves mean_j / var_j of every split chain in LDS; returns rhat (and ess if want_acov).
// have_moments: mean_j / var_j are already in LDS (the z-series moments sweep, series_z_phase) and are
// used as they are -- the means centre the lag products, nothing overwrites them.
template <class YF>
MDT_PHASE SeriesOut series_phase(const KP& kp_, YF Y, bool want_acov_, unsigned char* smem, bool have_moments_ = false) {
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    Y.make_uniform();
    const bool want_acov = uni(want_acov_), have_moments = uni(have_moments_);
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
            if (block == 0 && !have_moments) {
                // pass 1: mean of the split chain (Statistics.mean!, src/ess_rhat.jl:529)
                double sum = 0.0;
                for (int ti = 0; ti < ntile; ++ti) {
                    const long long tau = (long long)ti * TL;
                    const int tlen = (int)((niter - tau < TL) ? niter - tau : TL);
                    double loc = 0.0;
                    for (int i0 = tid; i0 < tlen; i0 += NT * 8) { // 8 loads in flight per thread
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + u * NT;
                            v[u] = (i < tlen) ? Y(c, r0 + tau + i) : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + u * NT;
                            if (i < tlen) {
                                if (ntile == 1) ser[i] = v[u];
                                loc += v[u];
                            }
                        }
                    }
                    sum += wg_sum(loc, red);
                }
                mu = sum / (double)niter;
                if (tid == 0) mean_j[j] = mu;
            } else {
                mu = mean_j[j];
            }
            // pass 2: centred lag products; the sum of squares for var_j is taken per thread over the
            // values it centres and reduced in a fixed order, so that it does not depend on the lag-block
            // layout: rhat(x) == ess_rhat(x).rhat bit for bit (test/ess_rhat.jl:66-98, 167-204)
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            double sq = 0.0;
            const int halo = want_acov ? (int)k0 + 8 * G : 8; // ser[t + k] is read for k < k0 + 8G
            for (int ti = 0; ti < ntile; ++ti) {
                const long long tau = (long long)ti * TL;
                const int tlen = (int)((niter - tau < TL) ? niter - tau : TL);
                if (ntile == 1 && block == 0 && !have_moments) {
                    for (int i = tid; i < tlen + halo; i += NT) {
                        const double cv = (i < tlen) ? ser[i] - mu : 0.0;
                        ser[i] = cv;
                        sq = fma(cv, cv, sq);
                    }
                } else {
                    __syncthreads();
                    for (int i0 = tid; i0 < tlen + halo; i0 += NT * 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + u * NT;
                            v[u] = (i < tlen + halo && tau + i < niter) ? Y(c, r0 + tau + i) - mu : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + u * NT;
                            if (i < tlen + halo) ser[i] = v[u];
                            if (i < tlen) sq = fma(v[u], v[u], sq);
                        }
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
            const double S2 = (block == 0) ? wg_sum(sq, red) : 0.0;
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
                    if (kq == 0) {
                        sm = S2;
                        if (!have_moments) var_j[j] = sm / (double)(niter - 1); // Statistics.var(...; mean, corrected=true)
                    }
                    A[kq] += sm;
                }
                __syncthreads();
            }
        }
        if (tid == 0) series_combine(kp, smem, block, kend, want_acov);
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


// ------------------------------------------------------------------ series phase for normal scores
// The hot variant of series_phase for the rank-normalised series (values bounded, |z| < 9).
// Input, SEG = false: z in time order (per-workgroup scratch).  SEG = true: z in chain-major SEGMENT order
//   as rank_phase_q leaves it (split chain j owns z[j * niter ..), members in ranking-round order, the draw
//   index of every slot beside it, 16 bits): a chain is streamed -- coalesced -- and scattered to time order
//   in LDS, so the ranking never issues a scattered global store.
// Moments: u = z - K_j with K_j = the score of the chain's first draw (exactly 0 deviation for a constant
//   chain); S1 = sum u, S2 = sum u^2 are accumulated in FIXED POINT (struct Fx): exact integer sums, so
//   they do not depend on the order of the terms.  This is the ONLY place the means and variances of a
//   z-series come from, so rhat(x) and ess_rhat(x).rhat are bit-identical, as in the reference
//   (test/ess_rhat.jl:167-204), whatever the geometry, the input order or the autocovariance path.
//   want_acov = false streams the chains once (any length), no LDS staging, no per-chain barrier.
// Autocovariance: one split chain per LDS buffer, the loads of chain j+1 in flight while chain j is
//   processed; lagged products of the shifted values are centred algebraically:
//     sum_t (u_t - mu)(u_{t+k} - mu) = P_k - mu (S1 - tail_k + S1 - head_k) + (n - k) mu^2,
//   head_k / tail_k = sums of the first / last k values.  With |mu| << sd this loses nothing
//   ("one-pass centring"); the reference's two-pass order (src/ess_rhat.jl:529-548) is
//   kept for raw draws in series_phase, where offsets can be large.
constexpr int SZ = (SMALL && MDT_NT == 1024) ? 6336 : 6400; // doubles per buffer: niter + lag halo must fit
constexpr int ZNBUF = SMALL ? 1 : 2;      // LDS buffers: double buffered where the LDS allows (one barrier per chain instead of two)
constexpr int SZ_EDGE = 256;              // first / last values kept for head_k / tail_k (maxlag <= 255)
constexpr int LD = (SZ + NT - 1) / NT;    // values per thread
constexpr int ZPW = NWAVE * 8 + 2 * NWAVE; // per-chain partials: 8 lags x segments, then S1 and S2 (int64) per wave
constexpr int OFF_ZS1 = OFF_LAGA;                              // moments-only view: int64 S1i[MAXM], S2i[MAXM], double Kt[MAXM]
static_assert(3 * MAXM * 8 <= 3 * NLAGTAB * 8, "moments-only view must fit the lag tables");
constexpr int OFF_ZSER = OFF_SER;                              // double zser[ZNBUF][SZ] (A / aux shared with series_phase)
constexpr int OFF_ZPART = OFF_ZSER + ZNBUF * SZ * 8;           // double zpart[3][ZPW]
constexpr int OFF_ZEDGE = OFF_ZPART + 3 * ZPW * 8;             // double zedge[3][2][SZ_EDGE]
constexpr int OFF_ZK = OFF_ZEDGE + 3 * 2 * SZ_EDGE * 8;        // double zk[4]
constexpr int LDS_ZSER_END = OFF_ZK + 4 * 8;
static_assert(LDS_ZSER_END <= LDS_BYTES, "z-series view exceeds the LDS carve");

// RAW = true: the series are raw draws of type XT in time order (kind = :basic, mcse): values are unbounded, so the
//   moments are plain double sums, taken by 1024 virtual threads (a 512-thread workgroup plays two each) and
//   reduced in a fixed order -- the same bits in every geometry.  They always come from the streaming branch
//   (want_acov = false); the autocovariance sweep then takes mean_j / var_j as given, like series_phase does.
constexpr int VT = 1024;                  // virtual threads of the raw-draw moment sums
constexpr int NV = VT / NT;               // virtual threads per thread
constexpr int VW = VT / 64;               // virtual waves = moment partials per chain
constexpr int ZMB = 16;                   // chains per batch of the raw-draw moment sweep
static_assert(ZMB * VW * 2 * 8 + ZMB * 8 <= 3 * NLAGTAB * 8, "raw-draw moment partials must fit the lag tables");
template <bool SEG, bool RAW = false, typename XT = double>
MDT_PHASE SeriesOut series_z_phase(const KP& kp_, const XT* z_, const unsigned short* loc_, bool want_acov_, unsigned char* smem) {
    static_assert(!(SEG && RAW), "raw draws are never in segment order");
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    const bool want_acov = uni(want_acov_);
    const MDT_GA XT* z = uniptr_g(z_);
    const MDT_GA unsigned short* locs = uniptr_g(loc_);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* scal = (double*)(smem + OFF_SCAL);
    int* ints = (int*)(smem + OFF_INTS);
    double* mean_j = (double*)(smem + OFF_MEAN);
    double* var_j = (double*)(smem + OFF_VAR);
    double* zser = (double*)(smem + OFF_ZSER);
    double* A = (double*)(smem + OFF_LAGA);
    double* aux = (double*)(smem + OFF_AUX);
    double* zpart = (double*)(smem + OFF_ZPART);
    double* zedge = (double*)(smem + OFF_ZEDGE);
    double* zk = (double*)(smem + OFF_ZK);
    const long long niter = kp.niter, D = kp.D;
    const int n = (int)niter;
    const int m = kp.m, s = kp.s, rem = kp.rem;
    const long long maxlag = kp.maxlag;
    const int method = kp.method;
    const int he = n < SZ_EDGE ? n : SZ_EDGE;
    SeriesOut out;

    const Fx fx = Fx::make(niter);
    auto chain_e0 = [&](int j) -> long long {
        const int c = j / s, kk = j - c * s;
        return (long long)c * D + split_row0(kk, niter, rem);
    };
    auto chain_K = [&](int j) -> double {
        if (SEG) return mean_j[j];
        // (a vector load on purpose: the scalar cache is not coherent with the stores of the rank phase)
        int zero = 0;
        asm volatile("" : "+v"(zero));
        return (double)z[chain_e0(j) + zero];
    };

    if (!want_acov && RAW) {
        // ---- moments only, raw draws: per virtual thread sums of u = x - K_j and u^2, fixed-order reduction
        double* zmom = (double*)(smem + OFF_ZS1);     // [ZMB][VW][2]
        double* Kt = zmom + ZMB * VW * 2;             // [ZMB]
        constexpr int SU = 8;                         // loads in flight per thread (a multiple of NV)
        static_assert(SU % NV == 0, "virtual-thread bookkeeping");
        for (int jb = 0; jb < m; jb += ZMB) {
            const int nb = (m - jb < ZMB) ? m - jb : ZMB;
            for (int jl = 0; jl < nb; ++jl) {
                const int j = jb + jl;
                const double Kj = chain_K(j);
                const MDT_GA XT* zc = z + chain_e0(j);
                double s1[NV], s2[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) { s1[q] = 0.0; s2[q] = 0.0; }
                for (int i0 = 0; i0 < n; i0 += NT * SU) {
                    double vv[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int i = i0 + u * NT + tid;
                        vv[u] = (i < n) ? (double)zc[i] : Kj;
                    }
#pragma unroll
                    for (int u = 0; u < SU; ++u) { // element i belongs to virtual thread i mod 1024 = (u mod NV) * NT + tid
                        const double uu = vv[u] - Kj; // exactly 0 for the padding lanes
                        s1[u % NV] += uu;
                        s2[u % NV] = fma(uu, uu, s2[u % NV]);
                    }
                }
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const double r1 = wave_sum(s1[q]), r2 = wave_sum(s2[q]);
                    if (lane == 0) { zmom[(jl * VW + q * NWAVE + w) * 2] = r1; zmom[(jl * VW + q * NWAVE + w) * 2 + 1] = r2; }
                }
                if (tid == 0) Kt[jl] = Kj;
            }
            __syncthreads();
            if (tid < nb) {
                double S1 = 0.0, S2 = 0.0;
                for (int q = 0; q < VW; ++q) { S1 += zmom[(tid * VW + q) * 2]; S2 += zmom[(tid * VW + q) * 2 + 1]; }
                const double mu = S1 / (double)n;
                mean_j[jb + tid] = Kt[tid] + mu;
                var_j[jb + tid] = (S2 - mu * S1) / (double)(n - 1);
            }
            __syncthreads();
        }
        if (tid == 0) series_combine(kp, smem, 0, 1, false);
        __syncthreads();
        out.rhat = scal[2];
        out.ess = __builtin_nan("");
        __syncthreads();
        return out;
    }

    if (!want_acov) {
        // ---- moments only: stream every chain once, integer sums, one LDS atomic per wave and chain
        long long* S1i = (long long*)(smem + OFF_ZS1);
        long long* S2i = S1i + MAXM;
        double* Kt = (double*)(S2i + MAXM);
        for (int k = tid; k < 2 * MAXM; k += NT) S1i[k] = 0;
        __syncthreads();
        constexpr int SU = 8; // loads in flight per thread
        for (int j = 0; j < m; ++j) {
            const double Kj = chain_K(j);
            const MDT_GA XT* zc = z + (SEG ? (long long)j * niter : chain_e0(j));
            long long a1 = 0, a2 = 0;
            for (int i0 = 0; i0 < n; i0 += NT * SU) {
                double vv[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = i0 + u * NT + tid;
                    vv[u] = (i < n) ? (double)zc[i] : Kj;
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const double uu = vv[u] - Kj; // exactly 0 for the padding lanes
                    a1 += fx.u(uu);
                    a2 += fx.uu(uu * uu);
                }
            }
            a1 = wave_sum_i64(a1);
            a2 = wave_sum_i64(a2);
            if (lane == 0) {
                atomicAdd((unsigned long long*)&S1i[j], (unsigned long long)a1);
                atomicAdd((unsigned long long*)&S2i[j], (unsigned long long)a2);
            }
            if (tid == 0) Kt[j] = Kj;
        }
        __syncthreads();
        for (int j = tid; j < m; j += NT) {
            const double S1 = (double)S1i[j] * fx.i1, S2 = (double)S2i[j] * fx.i2;
            const double mu = S1 / (double)n;
            var_j[j] = (S2 - mu * S1) / (double)(n - 1);
            mean_j[j] = Kt[j] + mu;
        }
        __syncthreads();
        if (tid == 0) series_combine(kp, smem, 0, 1, false);
        __syncthreads();
        out.rhat = scal[2];
        out.ess = __builtin_nan("");
        __syncthreads();
        return out;
    }

    double v[LD];
    int loc[LD];
    double Kj = 0.0;
    auto issue_loads = [&](int j) {
        if (SEG) {
            const MDT_GA XT* zc = z + (long long)j * niter;
            const MDT_GA unsigned short* lc = locs + (long long)j * niter;
#pragma unroll
            for (int u = 0; u < LD; ++u) {
                const int i = u * NT + tid;
                v[u] = (i < n) ? (double)zc[i] : 0.0;
                loc[u] = (i < n) ? (int)lc[i] : i;
            }
        } else {
            const MDT_GA XT* zc = z + chain_e0(j);
#pragma unroll
            for (int u = 0; u < LD; ++u) {
                const int i = u * NT + tid;
                v[u] = (i < n) ? (double)zc[i] : 0.0;
                loc[u] = i;
            }
        }
        Kj = chain_K(j);
    };

    for (int k = tid; k < NLAGTAB; k += NT) { A[k] = 0.0; aux[k] = 0.0; aux[NLAGTAB + k] = 0.0; }
    if (tid == 0) { ints[1] = 0; ints[2] = 0; }
    __syncthreads();

    long long k0 = 0;
    int block = 0;
    while (true) {
        const int G = 4;                 // 32 lags per block: lag-groups of 8 x NWAVE/4 time-segments
        long long kend = k0 + 32;
        if (kend > maxlag + 1) kend = maxlag + 1;
        const int nseg = NWAVE / G;
        const int halo = (int)k0 + 8 * G + 2;
        const int g = w % G, q = w / G;
        const int seglen = (((n + nseg - 1) / nseg) + 1) & ~1; // even, so that every lane's pair is 16-byte aligned
        const int t0 = q * seglen;
        const int t1 = (t0 + seglen < n) ? t0 + seglen : n;
        const int kb = (int)k0 + 8 * g;

        // Lagged products are linear in the chain, and only their SUM over chains is ever used
        // (src/ess_rhat.jl:170-178), so the accumulators run across all split chains and are reduced
        // once per sweep; per chain only the small centring terms are finalised (by the lanes that
        // own the lags) from S1, S2 and the first / last values.
        double corr = 0.0; // this lane's lag: sum over chains of the centring correction (or C_0 for lag 0)
        double bda_h = 0.0, bda_t = 0.0;
        auto finalize = [&](int j) {
            if (w != 0) return; // wave 0 owns the lags of a block (<= 32); everything below is DPP on full waves
            const int b3 = j % 3;
            const double* pp = zpart + b3 * ZPW;
            const long long* ppi = (const long long*)pp;
            const double* eh = zedge + b3 * 2 * SZ_EDGE;
            const double* et = eh + SZ_EDGE;
            const long long kq = k0 + lane;
            // S1, S2: exact integer sums of the wave partials (raw draws: from the given mean_j / var_j)
            double S1, S2, mu;
            if (RAW) {
                mu = mean_j[j] - zk[b3];
                S1 = (double)n * mu;
                S2 = var_j[j] * (double)(n - 1) + mu * S1;
            } else {
                S1 = (double)wave_sum_i64((lane < NWAVE) ? ppi[NWAVE * 8 + lane] : 0) * fx.i1;
                S2 = (double)wave_sum_i64((lane < NWAVE) ? ppi[NWAVE * 9 + lane] : 0) * fx.i2;
                mu = S1 / (double)n;
            }
            // head_k = sum_{t<k} u_t, tail_k = sum of the last k values (k clamped to the kept edge):
            // totals of the 64-wide chunks below k0, then an exclusive scan inside [k0, k0+64)
            double ch = 0.0, ct = 0.0, chq = 0.0, ctq = 0.0;
            const int k0c = (int)k0 < he ? (int)k0 : he;
            for (int base = 0; base < k0c; base += 64) {
                const int t = base + lane;
                const double a2 = (t < k0c) ? eh[t] : 0.0, b2 = (t < k0c) ? et[he - 1 - t] : 0.0;
                ch += wave_sum(a2);
                ct += wave_sum(b2);
                if (method == 2) { chq += wave_sum(a2 * a2); ctq += wave_sum(b2 * b2); }
            }
            const int t = (int)k0 + lane;
            const double a2 = (t < he) ? eh[t] : 0.0, b2 = (t < he) ? et[he - 1 - t] : 0.0;
            const double head = ch + (wave_sum_l63(a2) - a2), tail = ct + (wave_sum_l63(b2) - b2);
            double hsq = 0.0, tsq = 0.0;
            if (method == 2) {
                hsq = chq + (wave_sum_l63(a2 * a2) - a2 * a2);
                tsq = ctq + (wave_sum_l63(b2 * b2) - b2 * b2);
            }
            if (lane < 8 * G && kq < kend) {
                const int kk2 = (int)kq < he ? (int)kq : he;
                if (kq == 0) {
                    const double C = RAW ? var_j[j] * (double)(n - 1) : S2 - mu * S1; // same expression as the moments-only path: bit-identical R-hat
                    if (!RAW) {
                        var_j[j] = C / (double)(n - 1);
                        mean_j[j] = zk[b3] + mu;
                    }
                    corr += C;
                } else {
                    corr += (double)(n - kq) * mu * mu - mu * ((S1 - tail) + (S1 - head));
                }
                if (method == 2) {
                    // variogram pieces on centred values c = u - mu: sum_{t<k} c_t^2 = hsq - 2 mu head + k mu^2
                    bda_h += hsq - 2.0 * mu * head + (double)kk2 * mu * mu;
                    bda_t += tsq - 2.0 * mu * tail + (double)kk2 * mu * mu;
                }
            }
        };

        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        issue_loads(0);
        for (int j = 0; j < m; ++j) {
            double* buf = zser + (j % ZNBUF) * SZ;
            const int b3 = j % 3;
            double* eh = zedge + b3 * 2 * SZ_EDGE;
            double* et = eh + SZ_EDGE;
            double* pp = zpart + b3 * ZPW;
            // the chain, shifted by K_j, to time order (SEG: scattered here, in LDS, not in global memory),
            // its first / last values, and this thread's share of the integer moments
            long long a1 = 0, a2 = 0;
#pragma unroll
            for (int u = 0; u < LD; ++u) {
                const int i = u * NT + tid;
                if (i < n) {
                    const double uu = v[u] - Kj;
                    const int l = loc[u];
                    buf[l] = uu;
                    if (l < he) eh[l] = uu;
                    if (l >= n - he) et[l - (n - he)] = uu;
                    if (!RAW) { a1 += fx.u(uu); a2 += fx.uu(uu * uu); }
                } else if (i < n + halo) {
                    buf[i] = 0.0;
                }
            }
            if (!RAW) {
                a1 = wave_sum_i64(a1);
                a2 = wave_sum_i64(a2);
                if (lane == 0) { ((long long*)pp)[NWAVE * 8 + w] = a1; ((long long*)pp)[NWAVE * 9 + w] = a2; }
            }
            if (tid == 0) zk[b3] = Kj;
            MDT_PROF_ZMARK(13); // waiting for the loads + LDS stores + moments
            __syncthreads();
            MDT_PROF_ZMARK(14); // barrier
            if (j + 1 < m) issue_loads(j + 1);
            {
                // two consecutive t per lane, 16-byte LDS reads (full LDS rate): 6 reads feed 16 FMAs
                const double2* b2 = (const double2*)buf;
                for (int t = t0 + 2 * lane; t < t1; t += 128) {
                    const double2 a2 = b2[t >> 1];
                    const int pq = (t + kb) >> 1;
                    const double2 w0 = b2[pq], w1 = b2[pq + 1], w2 = b2[pq + 2], w3 = b2[pq + 3], w4 = b2[pq + 4];
                    acc[0] = fma(a2.x, w0.x, acc[0]); acc[0] = fma(a2.y, w0.y, acc[0]);
                    acc[1] = fma(a2.x, w0.y, acc[1]); acc[1] = fma(a2.y, w1.x, acc[1]);
                    acc[2] = fma(a2.x, w1.x, acc[2]); acc[2] = fma(a2.y, w1.y, acc[2]);
                    acc[3] = fma(a2.x, w1.y, acc[3]); acc[3] = fma(a2.y, w2.x, acc[3]);
                    acc[4] = fma(a2.x, w2.x, acc[4]); acc[4] = fma(a2.y, w2.y, acc[4]);
                    acc[5] = fma(a2.x, w2.y, acc[5]); acc[5] = fma(a2.y, w3.x, acc[5]);
                    acc[6] = fma(a2.x, w3.x, acc[6]); acc[6] = fma(a2.y, w3.y, acc[6]);
                    acc[7] = fma(a2.x, w3.y, acc[7]); acc[7] = fma(a2.y, w4.x, acc[7]);
                }
            }
            MDT_PROF_ZMARK(15); // lag products
            if (j > 0) finalize(j - 1); // its S1 / S2 / edges were complete before this iteration's barrier
            MDT_PROF_ZMARK(7);  // finalize
            if (ZNBUF == 1) __syncthreads(); // single buffer: the next chain overwrites what was just read
        }
        {
            double* pp = zpart; // lag partials of the whole sweep (the S1/S2 slots of buffer 0 are not touched)
            __syncthreads();
            finalize(m - 1);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double r = wave_sum(acc[i]);
                if (lane == 0) pp[(g * 8 + i) * nseg + q] = r;
            }
            __syncthreads();
            const long long kq = k0 + tid;
            if (tid < 8 * G && kq < kend) {
                double P = 0.0;
                for (int q2 = 0; q2 < nseg; ++q2) P += pp[tid * nseg + q2];
                A[kq] += (kq == 0) ? corr : P + corr;
                if (method == 2) { aux[kq] += bda_h; aux[NLAGTAB + kq] += bda_t; }
            }
        }
        __syncthreads();
        if (tid == 0) series_combine(kp, smem, block, kend, true);
        __syncthreads();
        if (ints[2]) break;
        k0 = kend;
        ++block;
    }
    out.rhat = scal[2];
    out.ess = scal[3];
    __syncthreads();
    return out;
}


// ------------------------------------------------------------------ series phase, FFTAutocovMethod
// src/ess_rhat.jl:103-118 (cache), :130-152 (update!), :181-195 (query): per split chain the centred
// series is zero-padded, transformed, |.|^2, transformed back, and the mean over chains of
// ac_k / ac_0 * var_j * (niter-1)/niter is the lag-k autocovariance (NaN when a chain is constant:
// 0/0, as in the reference).  Here: in-LDS radix-2 complex FFT, decimation in frequency forward
// (natural -> bit-reversed), |.|^2 in place (order-free), decimation in time inverse (bit-reversed ->
// natural), so no reordering pass is needed.  Pad length N = nextpow2(niter + maxlag + 1): only lags
// <= maxlag are ever read (:482,563,590), and for those any N >= niter + maxlag + 1 gives the same
// linear autocorrelation as the reference's nextprod([2,3], 2 niter - 1).
constexpr int FFT_MAXN = SMALL ? 4096 : 8192;               // complex points resident in LDS (64 / 128 KiB)
constexpr int OFF_FFT = OFF_AUX;                            // double2 fbuf[FFT_MAXN]
static_assert(OFF_FFT + FFT_MAXN * 16 <= LDS_BYTES, "FFT buffer exceeds the LDS carve");

template <class YF>
MDT_PHASE SeriesOut series_fft_phase(const KP& kp_, YF Y, unsigned char* smem) {
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    Y.make_uniform();
    const int tid = threadIdx.x;
    double* scal = (double*)(smem + OFF_SCAL);
    int* ints = (int*)(smem + OFF_INTS);
    double* mean_j = (double*)(smem + OFF_MEAN);
    double* var_j = (double*)(smem + OFF_VAR);
    double* A = (double*)(smem + OFF_LAGA);
    double2* fb = (double2*)(smem + OFF_FFT);
    const long long niter = kp.niter, maxlag = kp.maxlag;
    const int n = (int)niter, m = kp.m, s = kp.s, rem = kp.rem;
    int N = 1, logN = 0;
    while (N < n + (int)maxlag + 1) { N <<= 1; ++logN; }
    SeriesOut out;

    // mean_j / var_j were left in LDS by the moments sweep that ran just before (same expressions
    // as every other path, so R-hat is bit-identical across autocov methods, test/ess_rhat.jl:228)
    for (int k = tid; k < NLAGTAB; k += NT) A[k] = 0.0;
    if (tid == 0) { ints[1] = 0; ints[2] = 0; }
    __syncthreads();

    for (int j = 0; j < m; ++j) {
        const int c = j / s, kk = j - c * s;
        const long long r0 = split_row0(kk, niter, rem);
        const double mu = mean_j[j];
        for (int i = tid; i < N; i += NT) {
            double2 v;
            v.x = (i < n) ? Y(c, r0 + i) - mu : 0.0;
            v.y = 0.0;
            fb[i] = v;
        }
        __syncthreads();
        // forward, decimation in frequency
        for (int half = N >> 1; half >= 1; half >>= 1) {
            for (int b = tid; b < (N >> 1); b += NT) {
                const int pos = b & (half - 1);
                const int i = ((b - pos) << 1) + pos, i2 = i + half;
                const double2 u = fb[i], v = fb[i2];
                double sn, cs;
                sincospi(-(double)pos / (double)half, &sn, &cs); // exp(-2 pi i pos / (2 half))
                const double dx = u.x - v.x, dy = u.y - v.y;
                double2 o1, o2;
                o1.x = u.x + v.x; o1.y = u.y + v.y;
                o2.x = dx * cs - dy * sn; o2.y = dx * sn + dy * cs;
                fb[i] = o1; fb[i2] = o2;
            }
            __syncthreads();
        }
        for (int i = tid; i < N; i += NT) {
            double2 v = fb[i];
            v.x = fma(v.x, v.x, v.y * v.y);
            v.y = 0.0;
            fb[i] = v;
        }
        __syncthreads();
        // inverse (unnormalised: only ratios are used), decimation in time
        for (int half = 1; half <= (N >> 1); half <<= 1) {
            for (int b = tid; b < (N >> 1); b += NT) {
                const int pos = b & (half - 1);
                const int i = ((b - pos) << 1) + pos, i2 = i + half;
                const double2 u = fb[i], v = fb[i2];
                double sn, cs;
                sincospi((double)pos / (double)half, &sn, &cs);
                double2 t, o1, o2;
                t.x = v.x * cs - v.y * sn; t.y = v.x * sn + v.y * cs;
                o1.x = u.x + t.x; o1.y = u.y + t.y;
                o2.x = u.x - t.x; o2.y = u.y - t.y;
                fb[i] = o1; fb[i2] = o2;
            }
            __syncthreads();
        }
        if (tid <= maxlag) A[tid] += fb[tid].x / fb[0].x * var_j[j]; // :192, summed over chains in order
        __syncthreads();
    }
    if (tid == 0) series_combine(kp, smem, 1 /* moments already combined */, maxlag + 1, true, true);
    __syncthreads();
    out.rhat = scal[2];
    out.ess = scal[3];
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
MDT_PHASE double indicator_phase(const KP& kp_, const T* xp_, double qlo_, double qhi_, unsigned char* smem, int fold_ = 0, T med_ = (T)0) {
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    const double qlo = uni(qlo_), qhi = uni(qhi_);
    const int fold = uni(fold_); // indicators of |x - med| (mad proxy, src/ess_rhat.jl:643-646) instead of x
    const T med = uni(med_);
    const MDT_GA T* xp = uniptr_g(xp_);
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
                T v = (t < D) ? xp[(long long)c * D + t] : (T)0;
                if (fold) { const T d = (T)(v - med); v = d < (T)0 ? -d : d; }
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
                // wave -> (indicator, word segment); lanes own lags.  Only A_k = sum_t I_t I_{t+k} walks the chain per lag
                // (S is the same for every lag, T_k and H_k are S minus a few leading / trailing words).
                {
                    const int ind = w & 1, seg = w >> 1;
                    constexpr int NSEG = NWAVE / 2;
                    const unsigned long long* sb = sbits + ind * BITW;
                    const long long k = k0 + lane;
                    const int wq = (nwS + NSEG - 1) / NSEG, i0 = seg * wq, i1 = (i0 + wq < nwS) ? i0 + wq : nwS;
                    int sxp = 0, akp = 0;
                    if (k < kend) {
                        const int q = (int)(k >> 6), r = (int)(k & 63);
                        unsigned long long lo = (i0 + q < nwS) ? sb[i0 + q] : 0ull;
                        for (int i = i0; i < i1; ++i) {
                            const unsigned long long v = sb[i];
                            const unsigned long long hi = (i + q + 1 < nwS) ? sb[i + q + 1] : 0ull;
                            const unsigned long long vs = r ? ((lo >> r) | (hi << (64 - r))) : lo;
                            sxp += __popcll(v);
                            akp += __popcll(v & vs);
                            lo = hi;
                        }
                    }
                    int* ipart = (int*)(smem + OFF_IPART);
                    ipart[((ind * NSEG + seg) * 64 + lane) * 2] = sxp;
                    ipart[((ind * NSEG + seg) * 64 + lane) * 2 + 1] = akp;
                }
                __syncthreads();
                if (w < 2) {
                    const int ind = w;
                    constexpr int NSEG = NWAVE / 2;
                    const unsigned long long* sb = sbits + ind * BITW;
                    const int* ipart = (const int*)(smem + OFF_IPART);
                    const long long k = k0 + lane;
                    if (k < kend) {
                        long long Sx = 0, Ak = 0;
                        for (int sg = 0; sg < NSEG; ++sg) { Sx += ipart[((ind * NSEG + sg) * 64 + lane) * 2]; Ak += ipart[((ind * NSEG + sg) * 64 + lane) * 2 + 1]; }
                        const int q = (int)(k >> 6);
                        long long head = 0, tail = 0; // ones among the first k / the last k positions
                        for (int i = 0; i <= q && i < nwS; ++i) {
                            const long long nb2 = k - ((long long)i << 6);
                            const unsigned long long mk = nb2 >= 64 ? ~0ull : (nb2 <= 0 ? 0ull : ((1ull << nb2) - 1ull));
                            head += __popcll(sb[i] & mk);
                        }
                        {
                            const long long p0 = niter - k;
                            for (int i = (int)((p0 > 0 ? p0 : 0) >> 6); i < nwS; ++i) {
                                const long long sk = p0 - ((long long)i << 6);
                                const unsigned long long mk = sk <= 0 ? ~0ull : (sk >= 64 ? 0ull : ~((1ull << sk) - 1ull));
                                tail += __popcll(sb[i] & mk);
                            }
                        }
                        const long long Tk = Sx - head, Hk = Sx - tail;
                        // n^2 * sum_t (I_t - mu)(I_{t+k} - mu) with mu = S/n, exact in int64
                        const long long N = niter * niter * Ak - niter * Sx * (Hk + Tk) + (niter - k) * Sx * Sx;
                        double ck = (double)N / (dn * dn);
                        if (kp.method == 1 && k > 0 && (Sx == 0 || Sx == niter)) ck = __builtin_nan(""); // 0/0, :192
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

// ------------------------------------------------------------------ indicator phase, register-resident
// Pooled ballots straight from the registers (wave w, slot j holds elements j*NT + 64w .. +63, i.e.
// word j*16 + w of the pooled bit stream), all split chains and both indicators in parallel.
constexpr int IND_MAXW = SMALL ? 1024 : 1536; // words of the pooled stream per indicator (n <= 65536 / 98304)
constexpr int IND_CB = SMALL ? 8 : 16; // split chains per batch (2 indicators x 16 chains = 32 units)
constexpr int OFF_PBITS = OFF_BIG;                           // uint64 pbits[2][IND_MAXW]
constexpr int OFF_SB2 = OFF_PBITS + 2 * IND_MAXW * 8;        // uint64 sb2[SB2_WORDS]: units packed with stride nwS
constexpr int SB2_WORDS = SMALL ? 2048 : 4096;           // (>= 2 units of up to IND_MAXW + maxlag/64 + 2 words: checked by the host limits)
constexpr int OFF_ITAB = OFF_SB2 + SB2_WORDS * 8;            // double itab[2][2*IND_CB][64] (cov, dif)
constexpr int OFF_ICNT2 = OFF_ITAB + 2 * 2 * IND_CB * 64 * 8; // double icov[2][NLAGTAB], idif[2][NLAGTAB], imean[2][MAXM]
constexpr int LDS_IND2_END = OFF_ICNT2 + (4 * NLAGTAB + 2 * MAXM) * 8;
static_assert(LDS_IND2_END <= LDS_BYTES, "indicator (register) view exceeds the LDS carve");

template <typename T>
MDT_PHASE double indicator_phase_pooled(const KP& kp_, const T* xp_, T qlo_, T qhi_, unsigned char* smem, int fold_ = 0, T med_ = (T)0) {
    const MDT_CA KP& kp = MDT_UNIFORM_KP(kp_);
    const T qlo = uni(qlo_), qhi = uni(qhi_);
    const int fold = uni(fold_);
    const T med = uni(med_);
    const MDT_GA T* xp = uniptr_g(xp_);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* scal = (double*)(smem + OFF_SCAL);
    int* ints = (int*)(smem + OFF_INTS);
    unsigned long long* pbits = (unsigned long long*)(smem + OFF_PBITS);
    unsigned long long* sb2 = (unsigned long long*)(smem + OFF_SB2);
    double* itab = (double*)(smem + OFF_ITAB);
    double* icov = (double*)(smem + OFF_ICNT2);
    double* idif = icov + 2 * NLAGTAB;
    double* imean = idif + 2 * NLAGTAB;
    const long long niter = kp.niter, D = kp.D, n = kp.n, maxlag = kp.maxlag;
    const int s = kp.s, m = kp.m;
    const int nwP = (int)((n + 63) >> 6);
    const int nwS = (int)((niter + 63) >> 6);
    const double dn = (double)niter;

    // pooled ballots: wave w of pass j holds elements (j*NT + 64w .. +63) = word j*NWAVE + w
    for (long long j0 = 0; j0 * NT < n; j0 += UB) {
        T xb[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) { const long long e = (j0 + u) * NT + tid; xb[u] = (e < n) ? xp[e] : (T)0; }
        if (fold) {
#pragma unroll
            for (int u = 0; u < UB; ++u) { const T d = (T)(xb[u] - med); xb[u] = d < (T)0 ? -d : d; }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long long e = (j0 + u) * NT + tid;
            const unsigned long long b0 = __ballot(e < n && xb[u] <= qlo);
            const unsigned long long b1 = __ballot(e < n && xb[u] <= qhi);
            const long long wd = (j0 + u) * NWAVE + w;
            if (lane == 0 && wd < IND_MAXW) { pbits[wd] = b0; pbits[IND_MAXW + wd] = b1; }
        }
    }
    for (int k = tid; k < 4 * NLAGTAB; k += NT) icov[k] = 0.0;
    if (tid < 2) { ints[4 + tid] = 0; ints[6 + tid] = 0; }
    __syncthreads();

    long long k0 = 0;
    while (true) {
        // 16 lags per block: a wave covers one unit as 16 lags x 4 word-segments, so that a well-mixed chain (Geyer
        // stops after a few lags) costs a quarter of a 64-lag block
        long long kend = k0 + 16;
        if (kend > maxlag + 1) kend = maxlag + 1;
        // units (chain, indicator) are stored with a stride of nwU words: nwS words of bits (tail masked) and
        // zero words behind them, so that a shifted read never needs a bounds check
        const int nwU = nwS + (int)(maxlag >> 6) + 2;
        int cb = SB2_WORDS / (2 * nwU);
        cb = cb < 1 ? 1 : (cb > IND_CB ? IND_CB : cb);
        for (int jb = 0; jb < m; jb += cb) {
            const int nb = (m - jb < cb) ? m - jb : cb;
            // split-aligned words of nb chains x 2 indicators
            for (int i = tid; i < 2 * nb * nwU; i += NT) {
                const int u = i / nwU, wi = i - u * nwU;      // u = 2*jl + ind
                const int jl = u >> 1, ind = u & 1;
                const int j2 = jb + jl, c = j2 / s, kk = j2 - c * s;
                const long long pos = (long long)c * D + split_row0(kk, niter, kp.rem) + ((long long)wi << 6);
                unsigned long long v = (wi < nwS) ? bits_at(pbits + ind * IND_MAXW, pos, nwP) : 0ull;
                const long long left = niter - ((long long)wi << 6);
                if (left < 64) v &= (left <= 0) ? 0ull : ((1ull << left) - 1ull);
                sb2[u * nwU + wi] = v;
            }
            __syncthreads();
            for (int u = w; u < 2 * nb; u += NWAVE) {
                const unsigned long long* sb = sb2 + u * nwU;
                const int kl = lane & 15, seg = lane >> 4;
                const long long k = k0 + kl;
                const int wq = (nwS + 3) >> 2, i0 = seg * wq, i1 = (i0 + wq < nwS) ? i0 + wq : nwS;
                int sxp = 0, akp = 0; // this lane's word segment
                if (k < kend) {
                    // A_k = sum_t I_t I_{t+k}: the only sum that needs the whole chain per lag.  S is the same
                    // for every lag; T_k = ones in [k, n) = S - ones in [0, k); H_k = ones in [0, n-k) =
                    // S - ones in [n-k, n): a few words each.
                    const int q = (int)(k >> 6), r = (int)(k & 63);
                    unsigned long long lo = sb[i0 + q];
                    for (int i = i0; i < i1; ++i) {
                        const unsigned long long v = sb[i];
                        const unsigned long long hi = sb[i + q + 1];
                        const unsigned long long vs = r ? ((lo >> r) | (hi << (64 - r))) : lo;
                        sxp += __popcll(v);
                        akp += __popcll(v & vs);
                        lo = hi;
                    }
                }
                sxp += __shfl_xor(sxp, 16); sxp += __shfl_xor(sxp, 32); // the 4 segments of a lag (exact integers)
                akp += __shfl_xor(akp, 16); akp += __shfl_xor(akp, 32);
                long long Sx = sxp, Ak = akp, Hk = 0, Tk = 0;
                double ck = 0.0, dk = 0.0;
                if (k < kend && seg == 0) {
                    const int q = (int)(k >> 6);
                    long long head = 0, tail = 0; // ones among the first k / the last k positions
                    for (int i = 0; i <= q; ++i) {
                        const long long nb2 = k - ((long long)i << 6); // bits of word i that belong to [0, k)
                        const unsigned long long mk = nb2 >= 64 ? ~0ull : (nb2 <= 0 ? 0ull : ((1ull << nb2) - 1ull));
                        head += __popcll(sb[i] & mk);
                    }
                    {
                        const long long p0 = niter - k; // first position of the tail
                        for (int i = (int)((p0 > 0 ? p0 : 0) >> 6); i < nwS; ++i) {
                            const long long b0 = (long long)i << 6; // positions b0 .. b0+63
                            const long long sk = p0 - b0;           // bits below sk are not in the tail
                            const unsigned long long mk = sk <= 0 ? ~0ull : (sk >= 64 ? 0ull : ~((1ull << sk) - 1ull));
                            tail += __popcll(sb[i] & mk);
                        }
                    }
                    Tk = Sx - head;
                    Hk = Sx - tail;
                    const long long N = niter * niter * Ak - niter * Sx * (Hk + Tk) + (niter - k) * Sx * Sx;
                    ck = (double)N / (dn * dn);
                    // FFTAutocovMethod normalises by the lag-0 value of the same chain: 0/0 -> NaN for a
                    // constant split chain (src/ess_rhat.jl:192); Sx in {0, niter} <=> constant
                    if (kp.method == 1 && k > 0 && (Sx == 0 || Sx == niter)) ck = __builtin_nan("");
                    dk = (double)(Hk + Tk - 2 * Ak);
                    if (k == 0) imean[(u & 1) * MAXM + jb + (u >> 1)] = (double)Sx / dn;
                }
                if (seg == 0) {
                    itab[u * 64 + kl] = ck;
                    itab[(2 * IND_CB + u) * 64 + kl] = dk;
                }
            }
            __syncthreads();
            // fixed-order sum over the chains of this batch
            if (tid < 128) {
                const int ind = tid >> 6, l = tid & 63;
                const long long k = k0 + l;
                if (k < kend) {
                    double a = icov[ind * NLAGTAB + k], d = idif[ind * NLAGTAB + k];
                    for (int jl = 0; jl < nb; ++jl) {
                        a += itab[(2 * jl + ind) * 64 + l];
                        d += itab[(2 * IND_CB + 2 * jl + ind) * 64 + l];
                    }
                    icov[ind * NLAGTAB + k] = a;
                    idif[ind * NLAGTAB + k] = d;
                }
            }
            __syncthreads();
        }
        if (tid < 2) {
            const int ind = tid;
            const double* cv = icov + ind * NLAGTAB;
            const double* dv = idif + ind * NLAGTAB;
            const double dm = (double)m;
            const double W = cv[0] / dm / (double)(niter - 1);
            double mb = 0.0;
            for (int j = 0; j < m; ++j) mb += imean[ind * MAXM + j];
            mb /= dm;
            double vb = 0.0;
            for (int j = 0; j < m; ++j) { const double d = imean[ind * MAXM + j] - mb; vb = fma(d, d, vb); }
            vb /= (double)(m > 1 ? m - 1 : m);
            const double varp = kp.cf * W + vb;
            const double inv_varp = 1.0 / varp;
            const int bda = kp.method == 2;
            auto rho = [&](long long k) -> double {
                const double ma = bda ? W - dv[k] / dm / (2.0 * (double)(niter - k)) : cv[k] / dm / dn;
                return 1.0 - inv_varp * (W - ma);
            };
            double pt = scal[8 + 4 * ind], sum = scal[9 + 4 * ind];
            long long k = (long long)scal[10 + 4 * ind];
            const long long avail = kend - 1;
            if (!ints[6 + ind]) {
                if (!ints[4 + ind]) { pt = 1.0 + rho(1); sum = pt; k = 2; ints[4 + ind] = 1; }
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
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(MDT_WAVES_PER_EU, MDT_WAVES_PER_EU))) void mdt_kernel(const KP* __restrict__ kpp) {
    const KP& kp = *kpp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    double* ostat = (double*)(smem + OFF_OSTAT);
    double* scal = (double*)(smem + OFF_SCAL);
    unsigned* tpos = (unsigned*)(smem + OFF_TPOS);
    double* red = (double*)(smem + OFF_RED);
    double* mean_j = (double*)(smem + OFF_MEAN);
    double* var_j = (double*)(smem + OFF_VAR);
    const long long n = kp.n, D = kp.D;
    double* zb = kp.zbuf + (long long)blockIdx.x * kp.n_pad;
    unsigned* pbuf = kp.r2buf + (long long)blockIdx.x * kp.n_pad; // sorted-order slots (u16 or u32)
    unsigned short* bb = kp.binbuf + (long long)blockIdx.x * kp.n_pad;
    double* xs = kp.xsbuf + (long long)blockIdx.x * kp.n_pad;
    unsigned short* bs = kp.bsbuf + (long long)blockIdx.x * kp.n_pad;
    const bool pos16 = n < 65535;
    const double* __restrict__ ztab = kp.ztab;
    const double nan = __builtin_nan("");
    if (kp.prof && tid == 0) *(long long*)(smem + OFF_PROFT) = clock64();

    // parameters are handed out dynamically (one atomic ticket per parameter): slowly mixing
    // parameters need more lag blocks than well-mixed ones, and a static stride leaves CUs idle
    int* tick = (int*)(smem + OFF_INTS) + 11;
    while (true) {
        if (tid == 0) *tick = atomicAdd(kp.status + 1, 1);
        __syncthreads();
        const long long p = (long long)uni(*tick); // wave-uniform: every branch on p is a scalar branch
        __syncthreads();
        if (p >= kp.P) break;
        const T* __restrict__ xp = (const T*)kp.x + p * n;
        const SrcPlain<T> VX{(const MDT_GA T*)xp};
        const SerX<T> YX{(const MDT_GA T*)xp, D};
        const SerZ YZ{(const MDT_GA double*)zb, D};
        const bool zfast = (kp.niter + kp.maxlag + 40 <= SZ) && (kp.maxlag < SZ_EDGE);

        // nested R-hat combine from mean_j / var_j (src/rhat_nested.jl:152-184); lane 0 only
        auto nested_combine = [&]() -> double {
            const int K = kp.K, S = kp.S, s = kp.s;
            const int mk = K * s;
            double within = 0.0, smean_sum = 0.0;
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
            __syncthreads(); // (an exit from the parameter without any barrier since the ticket wedged the workgroup)
            continue;
        }

        // FFTAutocovMethod: by default the same linear autocorrelation from lagged products (zero padding makes the
        // transform's result exactly those sums: ac_k / ac_0 * var_j * (niter-1)/niter == sum_t c_t c_{t+k} / niter;
        // only the lags Geyer's scan visits are computed, 4x faster at config 2), plus the one place where the
        // formulas differ: a constant split chain is 0/0 = NaN in the reference (src/ess_rhat.jl:192).  With
        // MDT_FFT_LDS=1 the in-LDS transform runs whenever the padded chain fits.
        const bool fft_lds = kp.method == 1 && kp.fft_lds && kp.niter + kp.maxlag + 1 <= FFT_MAXN;
        auto fft_const_chain_nan = [&](double ess) -> double {
            if (kp.method != 1 || fft_lds) return ess;
            if (tid == 0) {
                int c0 = 0;
                for (int j = 0; j < kp.m; ++j) c0 |= (var_j[j] == 0.0) ? 1 : 0;
                ((int*)(smem + OFF_INTS))[16] = c0;
            }
            __syncthreads();
            const bool anyc = ((int*)(smem + OFF_INTS))[16] != 0;
            __syncthreads();
            return anyc ? nan : ess;
        };

        auto basic_ess_rhat = [&](auto Y, bool acov) -> SeriesOut { // _ess_rhat_basic! on the series Y
            SeriesOut so;
            if (acov && fft_lds) {
                (void)series_phase(kp, Y, false, smem); // moments -> mean_j / var_j / R-hat
                so = series_fft_phase(kp, Y, smem);
            } else {
                so = series_phase(kp, Y, acov, smem);
                if (acov) so.ess = fft_const_chain_nan(so.ess);
            }
            return so;
        };

        if (kp.est == 5) { // ess(kind = std) / mcse(kind = std): basic ESS of (x - mean)^2 (src/ess_rhat.jl:640-642)
            double loc = 0.0;
            for (long long e = tid; e < n; e += NT) loc += (double)xp[e];
            const T muT = (T)(wg_sum(loc, red) / (double)n);
            const SerSq<T> YS{(const MDT_GA T*)xp, D, muT};
            const SeriesOut so = basic_ess_rhat(YS, kp.ess_ok != 0);
            const double ess = kp.ess_ok ? so.ess : nan;
            if (kp.mode == 4) {
                // src/mcse.jl:52-65: sqrt((E[mu4] / E[var] - E[var]) / S) / 2, E[var] = mean(proxy), E[mu4] = mean(proxy^2)
                double l1 = 0.0, l2 = 0.0;
                for (long long e = tid; e < n; e += NT) {
                    const T d = (T)(xp[e] - muT);
                    const double y = (double)(T)(d * d);
                    l1 += y;
                    l2 = fma(y, y, l2);
                }
                const double mv = wg_sum(l1, red) / (double)n, m4 = wg_sum(l2, red) / (double)n;
                if (tid == 0) store_out<T>(kp.out_eb, p, sqrt((m4 / mv - mv) / ess) / 2.0);
            } else if (tid == 0) {
                store_out<T>(kp.out_eb, p, ess);
            }
            __syncthreads();
            MDT_PROF_MARK(PH_SERB);
            continue;
        }

        if (kp.basic && kp.mode != 3) {
            // raw draws: moments (and R-hat) from the streaming sweep of series_z_phase<RAW>, whatever follows --
            // rhat(x) and ess_rhat(x).rhat are the same bits; then the autocovariance: the in-LDS FFT if asked
            // for, the one-chain-per-LDS-buffer sweep when a split chain fits, the tiled series_phase otherwise
            const bool acov = kp.need_eb && kp.ess_ok;
            SeriesOut so = series_z_phase<false, true, T>(kp, xp, nullptr, false, smem);
            if (acov) {
                SeriesOut sa;
                if (fft_lds) sa = series_fft_phase(kp, YX, smem);
                else if (zfast) sa = series_z_phase<false, true, T>(kp, xp, nullptr, true, smem);
                else sa = series_phase(kp, YX, true, smem, true);
                so.ess = fft_lds ? sa.ess : fft_const_chain_nan(sa.ess);
            }
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
            MDT_PROF_MARK(PH_SERB);
            continue;
        }


        // ---- ranking of x; capture median / quantile neighbours on the way
        const bool est_ind = kp.est == 4 || kp.est == 6 || kp.est == 7; // estimators whose proxy is an indicator
        const bool need_med = kp.need_rt || (kp.mode == 3 && kp.fold) || ((kp.est == 4 || kp.est == 6) && kp.ess_ok);
        const bool need_q = (kp.need_et || kp.est == 7) && kp.ess_ok;
        if (kp.mode == 5 && kp.tpos_arr[2 * p] == 0xFFFFFFFFu) { // mcse quantile of a NaN ESS (src/mcse.jl:95-97)
            if (tid == 0) store_out<T>(kp.out_eb, p, nan);
            __syncthreads();
            continue;
        }
        int ntgt = 0;
        if (tid == 0) {
            int t = 0;
            if (kp.mode == 5) { tpos[t++] = kp.tpos_arr[2 * p]; tpos[t++] = kp.tpos_arr[2 * p + 1]; }
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
        const bool need_rank_x = kp.need_eb || kp.need_rb || need_med || need_q || (kp.mode == 3 && !kp.fold) || kp.mode == 5;
        // the z-series phase can take the scores in segment order (no scattered store in the ranking at all)
        const bool seg_req = zfast && !fft_lds && kp.mode != 3;
        int rk = 0; // what the last ranking left in zb: 2 = segment order, otherwise time order
        if (need_rank_x) {
            if (kp.rank_variant == 2) {
                rk = rank_phase_q(kp, VX, n, smem, zb, pbuf, bb, xs, bs, ntgt, kp.mode == 3,
                                  // order statistics only (no score is consumed): estimators, mcse quantile, tail-only requests
                                  (est_ind || kp.mode == 5 || (!(kp.need_eb || kp.need_rb) && kp.mode != 3)) ? 4
                                      // nested R-hat only needs the moments (and has no ess_rhat twin whose bits it must match)
                                      : ((kp.mode == 1 && !kp.need_eb) ? 2 : (seg_req ? 1 : 0)));
                if (!rk) rank_phase_p(kp, VX, n, smem, zb, pbuf, bb, xs, bs, ntgt, kp.mode == 3);
            } else if (kp.rank_variant == 1) rank_phase_p(kp, VX, n, smem, zb, pbuf, bb, xs, bs, ntgt, kp.mode == 3);
            else if (pos16) rank_phase_b<unsigned short>(kp, VX, n, smem, zb, (unsigned short*)pbuf, bb, ntgt, kp.mode == 3);
            else rank_phase_b<unsigned>(kp, VX, n, smem, zb, pbuf, bb, ntgt, kp.mode == 3);
        }
        MDT_PROF_MARK(PH_RANKX);
        if (kp.mode == 5) { // _mcse_quantile (src/mcse.jl:109-117): half the distance of two order statistics, in T
            if (tid == 0) store_out<T>(kp.out_eb, p, (double)(T)(((T)ostat[1] - (T)ostat[0]) / (T)2));
            __syncthreads();
            continue;
        }

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
        const SrcFold<T> VF{(const MDT_GA T*)xp, medT};
        auto rank_folded = [&](int nt, int out_req) {
            rk = 0;
            if (kp.rank_variant == 2) {
                rk = rank_phase_q(kp, VF, n, smem, zb, pbuf, bb, xs, bs, nt, kp.mode == 3, out_req);
                if (!rk) rank_phase_p(kp, VF, n, smem, zb, pbuf, bb, xs, bs, nt, kp.mode == 3);
            } else if (kp.rank_variant == 1) rank_phase_p(kp, VF, n, smem, zb, pbuf, bb, xs, bs, nt, kp.mode == 3);
            else if (pos16) rank_phase_b<unsigned short>(kp, VF, n, smem, zb, (unsigned short*)pbuf, bb, nt, kp.mode == 3);
            else rank_phase_b<unsigned>(kp, VF, n, smem, zb, pbuf, bb, nt, kp.mode == 3);
        };

        if (est_ind) {
            // ess(kind = median | mad | quantile(p)): basic ESS of an indicator (src/ess_rhat.jl:630-659); both
            // thresholds of the indicator phase are the same here
            double ee = nan;
            if (kp.ess_ok) {
                T thr = (kp.est == 7) ? qloT : medT;
                int fold = 0;
                if (kp.est == 6 && !has_nan) { // median of |x - median|: the folded ranking, order statistics only
                    if (tid == 0) { tpos[0] = (unsigned)((n - 1) / 2); tpos[1] = (unsigned)(n / 2); }
                    __syncthreads();
                    rank_folded(2, 4);
                    const T a = (T)ostat[0], b = (T)ostat[1];
                    thr = (n & 1) ? a : (T)(a / (T)2 + b / (T)2);
                    fold = 1;
                    __syncthreads();
                }
                if (kp.est == 7 && thr != thr) { if (tid == 0) atomicOr(kp.status, 2); } // Statistics.quantile throws on NaN
                else if (n <= (long long)IND_MAXW * 64 && 2 * (((kp.niter + 63) >> 6) + (kp.maxlag >> 6) + 2) <= SB2_WORDS)
                    ee = indicator_phase_pooled<T>(kp, xp, thr, thr, smem, fold, medT);
                else ee = indicator_phase<T>(kp, xp, (double)thr, (double)thr, smem, fold, medT);
            }
            if (tid == 0) store_out<T>(kp.out_eb, p, ee);
            __syncthreads();
            MDT_PROF_MARK(PH_IND);
            continue;
        }

        if (kp.mode == 3) { // stage output: _rank_normalize of x or of the folded x
            if (kp.fold) rank_folded(0, 0);
            for (long long e = tid; e < n; e += NT) {
                const double r = zb[e]; // doubled rank (store_r2 mode)
                if (kp.rank_out) kp.rank_out[p * n + e] = r * 0.5;
                if (kp.z_out) ((T*)kp.z_out)[p * n + e] = (T)ztab[(unsigned)r];
            }
            __syncthreads();
            continue;
        }

        double eb = nan, rb = nan, et = nan, rt = nan;
        if (kp.need_eb || kp.need_rb) {
            const bool acov = kp.need_eb && kp.ess_ok;
            SeriesOut so;
            // the moments of a z-series always come from series_z_phase (fixed point): R-hat is the same
            // bits whichever autocovariance path follows
            if (rk == 3) { // the ranking delivered mean_j / var_j itself (nested R-hat)
                so.rhat = nan; so.ess = nan;
            } else if (acov && fft_lds) {
                (void)series_z_phase<false>(kp, zb, nullptr, false, smem);
                so = series_fft_phase(kp, YZ, smem);
            } else if (zfast || !acov) {
                so = (rk == 2) ? series_z_phase<true>(kp, zb, (const unsigned short*)pbuf, acov, smem) : series_z_phase<false>(kp, zb, nullptr, acov, smem);
                if (acov) so.ess = fft_const_chain_nan(so.ess);
            } else {
                (void)series_z_phase<false>(kp, zb, nullptr, false, smem);
                so = series_phase(kp, YZ, true, smem, true);
                so.ess = fft_const_chain_nan(so.ess);
            }
            rb = (kp.mode == 1) ? 0.0 : so.rhat;
            if (acov) eb = so.ess;
            if (kp.mode == 1) {
                if (tid == 0) scal[20] = nested_combine();
                __syncthreads();
                rb = scal[20];
                __syncthreads();
            }
        }
        MDT_PROF_MARK(PH_SERB);
        if (kp.need_rt) {
            rank_folded(0, kp.mode != 3 ? 2 : 0);
            MDT_PROF_MARK(PH_RANKF);
            // R-hat of the folded scores needs the split-chain moments only: the two-level ranking delivers them
            // itself (rk == 3, no score was ever written); otherwise one streaming sweep over the scores
            if (rk == 3) {
                if (tid == 0) series_combine(kp, smem, 0, 1, false);
                __syncthreads();
                rt = scal[2];
                __syncthreads();
            } else {
                rt = series_z_phase<false>(kp, zb, nullptr, false, smem).rhat;
            }
            if (kp.mode == 1) {
                if (tid == 0) scal[20] = nested_combine();
                __syncthreads();
                rt = scal[20];
                __syncthreads();
            }
            MDT_PROF_MARK(PH_SERT);
        }
        if (kp.need_et && kp.ess_ok) {
            if (qloT != qloT || qhiT != qhiT) { et = nan; if (tid == 0) atomicOr(kp.status, 2); }
            else if (n <= (long long)IND_MAXW * 64 && 2 * (((kp.niter + 63) >> 6) + (kp.maxlag >> 6) + 2) <= SB2_WORDS)
                et = indicator_phase_pooled<T>(kp, xp, qloT, qhiT, smem);
            else et = indicator_phase<T>(kp, xp, (double)qloT, (double)qhiT, smem);
            MDT_PROF_MARK(PH_IND);
        }
        if (tid == 0) {
            store_out<T>(kp.out_eb, p, eb);
            store_out<T>(kp.out_rb, p, rb);
            store_out<T>(kp.out_et, p, et);
            store_out<T>(kp.out_rt, p, rt);
            store_out<T>(kp.out_rrank, p, jl_max(rt, rb)); // map(max, Rtail, Rbulk), src/ess_rhat.jl:419,622
        }
        __syncthreads();
        MDT_PROF_MARK(PH_OUT);
    }
}

} // namespace MDT_NS
