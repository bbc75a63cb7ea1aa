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
//                 autocovariance is popcount(bits & bits>>k), exact integers.  In the fused request the bits come from
//                 the z-series while a chain sits in LDS (scores are strictly increasing in the rank).
//   fold merge    rhat(kind = :tail) ranks |x - median|: the ranking of x leaves the slice sorted, the folded values are
//                 two monotone runs mirrored at the median, their ranks are those of a two-way merge (mdt_fold.inc).
//   long chains   chains that do not fit the LDS: scores stay in partition order with a (tile, round) run table and halo
//                 strips; series_win_phase gathers time windows from them (mdt_series.inc).
//   many chains   chains * split_chains > 256: moments, R-hat and ESS with one thread per split chain (many_phase); the
//                 nested combination sums in sorted order (permutation-invariant bit for bit) (mdt_nested.inc).
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
#include <type_traits>

#if !defined(MDT_NS) || !defined(MDT_NT) || !defined(MDT_LDS_LIMIT) || !defined(MDT_WAVES_PER_EU)
#error "define MDT_NS, MDT_NT, MDT_LDS_LIMIT and MDT_WAVES_PER_EU before including mdt_device.hpp"
#endif

namespace MDT_NS {

constexpr int NT = MDT_NT;          // threads per workgroup
constexpr bool SMALL = (MDT_LDS_LIMIT <= 80 * 1024); // the 80 KiB tables (two workgroups per CU)
constexpr bool TINY = (MDT_LDS_LIMIT <= 40 * 1024);  // the 40 KiB tables (four 256-thread workgroups per CU): short slices
constexpr int NWAVE = NT / 64;      // 16 or 8 wave64
constexpr int K1 = NT / 4;          // coarse buckets (sample quantiles): every 4th of the NT sorted sample values
constexpr int FINE = 32;            // bins per coarse bucket: 1 reserved (== splitter) + 31 linear
constexpr int NREG = K1 * FINE;     // regular bins are 1..NREG
constexpr int BIN_UNDER = 0;        // below the smallest sample value
constexpr int BIN_TOPEQ = NREG + 1; // equal to the largest sample value (reserved)
constexpr int BIN_OVER = NREG + 2;  // above the largest sample value
constexpr int BIN_NAN = NREG + 3;   // NaN: ranked last, in index order (isless / stable sortperm)
constexpr int NBIN = NREG + 4;
constexpr int CH = TINY ? 2816 : SMALL ? (MDT_NT == 512 ? 6400 : 4096) : 13760; // elements of bin-ordered data staged in LDS per round
constexpr int GRID = TINY ? 256 : SMALL ? 512 : 2048; // uniform cells over [min sample, max sample] that seed the splitter search
constexpr int MAXROUNDS = 192;      // => n <= ~2.6M elements per parameter
constexpr int TL = TINY ? 2048 : SMALL ? 4096 : 8192; // series tile (elements of one split chain resident in LDS)
constexpr int MAXLAG_CAP = SMALL ? 255 : 1000;
constexpr int HALO = SMALL ? 352 : 1136; // >= MAXLAG_CAP + 16 + 64 + 16: the last lag tile's shifted operand behind a window's last 64-block
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
constexpr int OFF_ORK = OFF_WTOT + 16 * 4;                // uint32 orank[8]: doubled tied rank of every captured order statistic
constexpr int OFF_RLO = OFF_ORK + 8 * 4;                  // int rlo[MAXROUNDS]
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
constexpr int BITW = TINY ? 1024 : SMALL ? 2048 : 4096;   // => draws / split_chains <= 65536 / 131072 / 262144
constexpr int OFF_SBITS = OFF_BIG;                        // uint64 sbits[2][BITW + IND_PAD] words of the split chains of a batch, each aligned to its first draw
constexpr int IND_PAD = 24;                               // zero words behind a chain's bits: a read shifted by a lag <= 1000 needs no bound
constexpr int OFF_ICNT = OFF_SBITS + 2 * (BITW + IND_PAD) * 8; // double icov[2][NLAGTAB] sum over chains
constexpr int OFF_IDIF = OFF_ICNT + 2 * NLAGTAB * 8;      // double idif[2][NLAGTAB] sum over chains of #(I_t != I_{t+k})
constexpr int OFF_IVAR = OFF_IDIF + 2 * NLAGTAB * 8;      // double imean[2][MAXM]
constexpr int OFF_IPART = OFF_IVAR + 2 * MAXM * 8;         // int ipart[IND_CBX][2][NWAVE/2][64][2]: (S, A_k) of a word segment, per chain of the batch
constexpr int IND_CBX = TINY ? 1 : SMALL ? 4 : 5;         // split chains per batch of indicator_phase (round 6): as many as fit the bit buffer, at most these
constexpr int LDS_IND_END = OFF_IPART + IND_CBX * 2 * (NWAVE / 2) * 64 * 2 * 4;
constexpr int LDS_BYTES = MDT_LDS_LIMIT; // 160 KiB per CU on gfx950: one big or two small workgroups
static_assert(LDS_RANK_END <= LDS_BYTES && LDS_SER_END <= LDS_BYTES && LDS_IND_END <= LDS_BYTES, "LDS budget");

// ------------------------------------------------------------------ kernel parameters
} // namespace MDT_NS
#include "mdt_kp.hpp"
namespace MDT_NS {
using mdtc::KP;

// phase slots of KP::prof
enum { PH_LOAD = 0, PH_RANKX = 1, PH_SERB = 2, PH_RANKF = 3, PH_SERT = 4, PH_IND = 5, PH_OUT = 6,
       PH_R_SAMPLE = 8, PH_R_HIST = 9, PH_R_SCAN = 10, PH_R_PLACE = 11, PH_R_COUNT = 12 };
#if defined(MDT_PROF_F) /* diagnostics build: the z-series sub-slots are lent to fold_merge_phase */
#define MDT_PROF_QMARK(slot)
#define MDT_PROF_ZMARK(slot)
#define MDT_PROF_FMARK(slot) MDT_PROF_MARK(slot)
#define MDT_PROF_QSEL(a, b) (a)
#elif defined(MDT_PROF_Q) /* diagnostics build: the z-series sub-slots are lent to the two-level rank phase */
#define MDT_PROF_QMARK(slot) MDT_PROF_MARK(slot)
#define MDT_PROF_ZMARK(slot)
#define MDT_PROF_FMARK(slot)
#define MDT_PROF_QSEL(a, b) (b)
#else
#define MDT_PROF_QMARK(slot)
#define MDT_PROF_ZMARK(slot) MDT_PROF_MARK(slot)
#define MDT_PROF_FMARK(slot)
#define MDT_PROF_QSEL(a, b) (a)
#endif
#if defined(MDT_STOP_AFTER) /* diagnostics build (tools/pmc_phases.sh): the ticket ends after phase k -- counters per phase by difference */
#define MDT_STOP(k) if (MDT_STOP_AFTER == (k)) { __syncthreads(); continue; }
#else
#define MDT_STOP(k)
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
// Streams -- arrays a parameter passes through once per pass, far larger than the 4 MiB L2 of an XCD -- are marked
// non-temporal so that they do not push the lines that ARE reused (the score table, the callee-saved register frames the
// phase functions store and reload at every call: ~100 KB per workgroup and call) out of the L2.  Measured in round 3
// on the streams of the tile-major path (x, the class-sorted chain image and its tags, the scores, the sorted copy):
// loads alone -0.2 %, stores alone -2.1 %, both -2.3 ... -3.5 % on the headline by box, -2 % on config 5; the streams
// of the histogram / partition / windowed passes (bit 2) re-read what they wrote soon enough to lose by it (config 3
// +1.3 %).  MDT_STREAM_NT: build-time switch, bit 0 the loads, bit 1 the stores, bits 2 / 3 the loads / stores of the second set (A/B timing).
#ifndef MDT_STREAM_NT
#define MDT_STREAM_NT 3
#endif
#if MDT_STREAM_NT & 1
#define MDT_LD_STREAM(p) __builtin_nontemporal_load(p)
#else
#define MDT_LD_STREAM(p) (*(p))
#endif
#if MDT_STREAM_NT & 2
#define MDT_ST_STREAM(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MDT_ST_STREAM(p, v) (*(p) = (v))
#endif
#if MDT_STREAM_NT & 4 /* the streams of the passes the tile-major path does not take (histogram, partition, windowed layout): loads */
#define MDT_LD_STREAM2(p) __builtin_nontemporal_load(p)
#else
#define MDT_LD_STREAM2(p) (*(p))
#endif
#if MDT_STREAM_NT & 8 /* ... and their stores */
#define MDT_ST_STREAM2(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MDT_ST_STREAM2(p, v) (*(p) = (v))
#endif
// Keys in the per-workgroup scratch (the partitioned copy, the sorted copy of the fold merge): stored in the type of the
// data -- Float32 slices as 4-byte keys (exact: every key IS a Float32), everything else as doubles (round 4: the scratch
// traffic of the config-5 shard, 10 000 x 16 Float32, was that of a Float64 array)
template <typename T>
using xs_of = typename std::conditional<std::is_same<T, float>::value, float, double>::type;
// The normal-score table in the same spirit: the scores of Float32 data are Float32 values (src/utils.jl:182 evaluates
// norminvcdf in the eltype), so their table holds 4-byte entries -- half the table traffic, and 1.9 MB instead of 3.8 at
// config 5's slice size, against 4 MB of L2 per XCD.
template <typename T>
struct ZTab {
    const MDT_GA xs_of<T>* p;
    __device__ __forceinline__ double operator[](unsigned long long i) const { return (double)p[i]; }
};
template <typename T, class KPT>
__device__ __forceinline__ ZTab<T> ztab_of(const KPT& kp) { return ZTab<T>{(const MDT_GA xs_of<T>*)uni64((unsigned long long)kp.ztab)}; }
template <typename T>
struct SrcPlain {      // x itself
    typedef xs_of<T> xs_t;
    typedef T val_t;
    const MDT_GA T* xp;
    __device__ __forceinline__ void make_uniform() { xp = (const MDT_GA T*)uni64((unsigned long long)xp); }
    __device__ __forceinline__ double operator()(long long e) const { return (double)xp[e]; }
    // element base + i with a wave-uniform base: scalar base register + 32-bit lane offset, no 64-bit address math
    __device__ __forceinline__ double at(long long base, unsigned i) const { return (double)(xp + base)[i]; }
    // the one read of a pass that takes the slice once (tile-major pass)
    __device__ __forceinline__ double at_stream(long long base, unsigned i) const { return (double)MDT_LD_STREAM((xp + base) + i); }
};
template <typename T>
struct SrcFold {       // |x - median|, evaluated in T like the reference (src/utils.jl:155)
    typedef xs_of<T> xs_t;
    typedef T val_t;
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
    }    __device__ __forceinline__ double at_stream(long long base, unsigned i) const {
        const T d = (T)(MDT_LD_STREAM((xp + base) + i) - med);
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
template <typename T>
struct SerInd {        // series = 1[x <= thr] or 1[|x - med| <= thr], compared in T: the expectand proxy of median / mad / quantile
    const MDT_GA T* xp; // (src/ess_rhat.jl:630-659); only where the indicator phases' tables do not reach (many_phase)
    long long D;
    T thr, med;
    int fold;
    __device__ __forceinline__ void make_uniform() { xp = (const MDT_GA T*)uni64((unsigned long long)xp); D = uni(D); thr = uni(thr); med = uni(med); fold = uni(fold); }
    __device__ __forceinline__ double operator()(int c, long long row) const {
        T v = xp[(long long)c * D + row];
        if (fold) { const T d = (T)(v - med); v = d < (T)0 ? -d : d; }
        return (v <= thr) ? 1.0 : 0.0;
    }
};
struct SerZ {          // series = normal scores, already in time order in the per-workgroup scratch
    const MDT_GA double* z;
    long long D;
    __device__ __forceinline__ void make_uniform() { z = (const MDT_GA double*)uni64((unsigned long long)z); D = uni(D); }
    __device__ __forceinline__ double operator()(int c, long long row) const { return z[(long long)c * D + row]; }
};
#define MDT_PHASE __device__ __attribute__((noinline))
// The phases of the common path (two-level ranking, z-series, fold merge, pooled indicators) are separate functions: each
// call saves and restores the 48 callee-saved VGPRs it touches (~100 KB of scratch stores and as many loads per workgroup and
// call), and still wins -- built into the kernel body they were measured in round 3 at +27 % (all four), +5 % / +2 % / +18 %
// (ranking / z-series / fold merge + indicators alone): the register allocation of one large function loses more than the
// frames cost.
#define MDT_PHASE_HOT MDT_PHASE
// The dynamic LDS block reaches the phase functions as a plain pointer, but the compiler knows which symbol it is and
// re-derives its address wherever it is used -- in a function that is not a kernel that is a LOAD from a table in
// global memory (llvm.amdgcn.dynlds.offset.table), rematerialised inside hot loops (17 exposed L2 round trips in
// series_z_phase).  Pinning the LDS address once per phase in a scalar register makes every later access base + offset.
#define MDT_LA __attribute__((address_space(3)))
__device__ __forceinline__ unsigned char* lds_pin(unsigned char* p) {
    unsigned a = (unsigned)(unsigned long long)(MDT_LA unsigned char*)p;
    a = (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    asm volatile("" : "+s"(a));
    return (unsigned char*)(MDT_LA unsigned char*)(unsigned long long)a;
}
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
    if (f32) ((float*)tab)[i0] = (float)z; // (Float32 data: a table of Float32 scores, see ZTab)
    else tab[i0] = z;
}

#include "mdt_rank.inc"

#include "mdt_fold.inc"

#include "mdt_series.inc"

#include "mdt_indicator.inc"

#include "mdt_nested.inc"

#include "mdt_short.inc"

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
    const ZTab<T> ztab = ztab_of<T>(kp);
    const double nan = __builtin_nan("");
    if (kp.prof && tid == 0) *(long long*)(smem + OFF_PROFT) = clock64();

    // parameters are handed out dynamically (one atomic ticket per parameter): slowly mixing
    // parameters need more lag blocks than well-mixed ones, and a static stride leaves CUs idle
    int* tick = (int*)(smem + OFF_INTS) + 11;
    // The persistent launch ends with a tail in which the workgroups run out of tickets one after the other; it is
    // shortest when the costly parameters went first.  Every launch therefore leaves the duration of each parameter
    // (kp.cost), an epilogue kernel sorts them (order_kernel), and the next launch of the same shape hands its tickets out
    // in that order (kp.order): requests repeated on the same chains -- monitoring a run, the bench -- start with the
    // slowly mixing parameters.  The order never changes a result (every parameter is computed independently).
    long long t_prev = 0, p_prev = -1;
    while (true) {
        if (tid == 0) {
            if (kp.cost) {
                const long long now = clock64();
                if (p_prev >= 0) kp.cost[p_prev] = (unsigned)((unsigned long long)(now - t_prev) >> 6);
                t_prev = now;
            }
            *tick = atomicAdd(kp.status + 1, 1);
        }
        __syncthreads();
        const long long tk = (long long)uni(*tick); // wave-uniform: every branch on it is a scalar branch
        __syncthreads();
        if (tk >= kp.P) break;
        long long p = tk;
        if (kp.order) { // (a vector load made uniform: the scalar cache may hold a stale line of the table)
            int zero = 0;
            asm volatile("" : "+v"(zero));
            p = (long long)(unsigned)uni((int)kp.order[tk + zero]);
        }
        p_prev = p;
        const T* xp = (const T*)kp.x + p * (kp.strided ? kp.pstride : n);
        if (kp.strided) {
            // x[d + cstride * c + pstride * p] (e.g. MCMCChains' (iterations, parameters, chains) storage, read in place):
            // the chains of this parameter are gathered -- each a coalesced stream -- into the workgroup's own dense
            // scratch slice, which stays cache-resident for the passes that follow
            T* xc = (T*)kp.xcbuf + (long long)blockIdx.x * kp.n_pad;
            constexpr int GU = 10; // loads in flight per thread
            if (D >= NT) {
                for (long long c = 0; c < kp.C; ++c) {
                    const MDT_GA T* src = (const MDT_GA T*)xp + c * kp.cstride;
                    MDT_GA T* dst = (MDT_GA T*)xc + c * D;
                    for (long long d0 = 0; d0 < D; d0 += (long long)NT * GU) {
                        T v[GU];
#pragma unroll
                        for (int u = 0; u < GU; ++u) { const long long d = d0 + u * NT + tid; v[u] = src[d < D ? d : D - 1]; }
#pragma unroll
                        for (int u = 0; u < GU; ++u) { const long long d = d0 + u * NT + tid; if (d < D) dst[d] = v[u]; }
                    }
                }
            } else { // many short chains: one pass over the pooled index
                const unsigned uD = (unsigned)D;
                for (unsigned e = (unsigned)tid; e < (unsigned)n; e += NT) {
                    const unsigned c = e / uD, d = e - c * uD;
                    xc[e] = xp[(long long)c * kp.cstride + d];
                }
            }
            __syncthreads();
            xp = xc;
        }
        if (tid == 0) ((int*)(smem + OFF_INTS))[19] = 0; // set when the ranking tables overflow: the parameter's outputs become NaN
        const bool many = kp.m > MAXM; // more split chains than the LDS moment tables hold (many_phase)
        const SrcPlain<T> VX{(const MDT_GA T*)xp};
        const SerX<T> YX{(const MDT_GA T*)xp, D};
        const SerZ YZ{(const MDT_GA double*)zb, D};
        // (a chain is processed in blocks of 64 values -- the last one zero-padded -- and the shifted operand of a lag tile
        // reaches maxlag + 16 past it: 63 + 16 + 1 values of room behind niter + maxlag)
        const bool zfast = (kp.niter + kp.maxlag + Z_MARGIN <= SZ) && (kp.maxlag < SZ_EDGE);

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
            // (round 6: a split chain that fits the LDS takes the one-chain-per-buffer sweep of the raw draws with the proxy applied on
            // the fly -- moments and lag products from one staged pass -- instead of the tiled series_phase)
            SeriesOut so;
            if (many) so = many_phase(kp, YS, xs, kp.ess_ok != 0, smem);
            else if (zfast && !fft_lds && kp.ess_ok && kp.sq_fast) {
                so = series_z_phase<false, true, T, true>(kp, xp, nullptr, true, smem, nullptr, 0.0, 0.0, true, (double)muT);
                so.ess = fft_const_chain_nan(so.ess);
            } else so = basic_ess_rhat(YS, kp.ess_ok != 0);
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

        // raw draws (kind = basic, mcse(mean)): rmode = 0 ess_rhat, 1 nested R-hat, 2 mcse(mean) into out_m
        auto raw_request = [&](int rmode, bool r_need_eb, void* out_m) {
            // raw draws: moments (and R-hat) from the streaming sweep of series_z_phase<RAW>, whatever follows --
            // rhat(x) and ess_rhat(x).rhat are the same bits; then the autocovariance: the in-LDS FFT if asked
            // for, the one-chain-per-LDS-buffer sweep when a split chain fits, the tiled series_phase otherwise
            const bool acov = r_need_eb && kp.ess_ok;
            if (many) { // more split chains than the moment tables hold: one thread per split chain (many_phase)
                const SeriesOut sm = many_phase(kp, YX, xs, acov, smem, rmode == 1 ? 1 : 0);
                const double ess = acov ? sm.ess : nan;
                if (rmode == 2) { // pooled corrected std over all draws and chains (src/mcse.jl:50), two passes
                    double loc = 0.0;
                    for (long long e = tid; e < n; e += NT) loc += (double)xp[e];
                    const double mu = wg_sum(loc, red) / (double)n;
                    loc = 0.0;
                    for (long long e = tid; e < n; e += NT) { const double d = (double)xp[e] - mu; loc = fma(d, d, loc); }
                    const double var = wg_sum(loc, red) / (double)(n - 1);
                    if (tid == 0) store_out<T>(out_m, p, sqrt(var) / sqrt(ess));
                } else if (tid == 0) {
                    if (rmode != 1) store_out<T>(kp.out_eb, p, ess);
                    store_out<T>(kp.out_rb, p, sm.rhat);
                }
                __syncthreads();
                return;
            }
            // (a split chain that fits the LDS: the autocovariance sweep takes the moments itself, from the chain it holds --
            // same virtual threads, order and expressions as the streaming sweep, so the same bits -- one pass over x less)
            const bool fuse = acov && zfast && !fft_lds;
            SeriesOut so;
            if (!fuse) so = series_z_phase<false, true, T>(kp, xp, nullptr, false, smem);
            if (acov) {
                SeriesOut sa;
                if (fft_lds) sa = series_fft_phase(kp, YX, smem);
                else if (zfast) sa = series_z_phase<false, true, T>(kp, xp, nullptr, true, smem, nullptr, 0.0, 0.0, true);
                else sa = series_phase(kp, YX, true, smem, true);
                if (fuse) so.rhat = sa.rhat;
                so.ess = fft_lds ? sa.ess : fft_const_chain_nan(sa.ess);
            }
            double ess = acov ? so.ess : nan;
            if (rmode == 1) {
                const double r = nested_combine_wg(kp, mean_j, var_j, smem);
                if (tid == 0) store_out<T>(kp.out_rb, p, r);
                __syncthreads();
            } else if (rmode == 2 && kp.rem == 0) {
                // pooled corrected std over all draws and chains (src/mcse.jl:50).  Every draw belongs to a split chain
                // here, so the pooled moments follow from the split-chain ones already in LDS (Chan's combination:
                // sum of squares about the pooled mean = sum_j [(niter - 1) var_j + niter (mean_j - mu)^2]) -- no further
                // pass over x
                if (tid == 0) {
                    double mu = 0.0;
                    for (int j = 0; j < kp.m; ++j) mu += mean_j[j];
                    mu /= (double)kp.m;
                    double ss = 0.0;
                    for (int j = 0; j < kp.m; ++j) { const double d = mean_j[j] - mu; ss += (double)(kp.niter - 1) * var_j[j] + (double)kp.niter * d * d; }
                    store_out<T>(out_m, p, sqrt(ss / (double)(n - 1)) / sqrt(ess));
                }
                __syncthreads();
            } else if (rmode == 2) {
                // (rows dropped by the split, D % split_chains != 0: two passes over the pooled slice)
                double loc = 0.0;
                for (long long e = tid; e < n; e += NT) loc += (double)xp[e];
                const double mu = wg_sum(loc, red) / (double)n;
                loc = 0.0;
                for (long long e = tid; e < n; e += NT) { const double d = (double)xp[e] - mu; loc = fma(d, d, loc); }
                const double var = wg_sum(loc, red) / (double)(n - 1);
                if (tid == 0) store_out<T>(out_m, p, sqrt(var) / sqrt(ess));
            } else if (tid == 0) {
                store_out<T>(kp.out_eb, p, ess);
                store_out<T>(kp.out_rb, p, so.rhat);
            }
            MDT_PROF_MARK(PH_SERB);
        };
        // fused request (mdt_mcse_nested, BASELINE config 5): mcse(kind = mean) of the raw draws (src/mcse.jl:45-51), then
        // rhat_nested(kind) below, from the same visit to the parameter: its slice is read from HBM once, the later passes
        // find it in the caches
        const bool fused_mn = kp.mode == 6;
        if (fused_mn) {
            raw_request(2, true, kp.out_mcse);
            __syncthreads();
            if (kp.basic) { raw_request(1, false, nullptr); continue; } // rhat_nested(kind = basic): raw draws again
        } else if (kp.basic && kp.mode != 3) {
            raw_request(kp.mode, kp.need_eb != 0, kp.out_eb);
            continue;
        }
        const bool nested = kp.mode == 1 || fused_mn;


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
        // chains that cannot fit the LDS whatever maxlag is -- the zfast test above with the smallest maxlag, 1: ONE margin for both
        // predicates, so that no chain length falls between them (until round 4 the margins were 80 and 40: split chains of SZ - 80
        // ... SZ - 41 draws were neither, and took the slow double pass) -- (so that rhat(x), which runs with maxlag = 1, and
        // ess_rhat(x) decide alike and their R-hat stays the same bits): the ranking delivers the moments itself and,
        // when an autocovariance follows, the windowed score layout of series_win_phase instead of a time-order scatter
        const bool long_chain = kp.winout && kp.niter + 1 + Z_MARGIN > SZ && !fft_lds && kp.mode != 3 && kp.m + 1 <= 64 &&
                                kp.maxlag + 40 + MAXROUNDS <= PT_CAP / 4;
        int rk = 0; // what the last ranking left in zb: 2 = segment order, otherwise time order; 5: the scores are in LDS (one-chunk path)
        // short slices: the whole parameter in one LDS chunk (mdt_short.inc); a NaN in the slice sends it down the general path
        bool oc = false;
        if constexpr (OC_ON) {
            oc = kp.onechunk && need_rank_x && kp.rank_variant == 2 && kp.mode == 0 && kp.est == 0 && n <= OC_MAX &&
                 kp.rem == 0 && kp.m <= 8 && kp.niter >= 64 && zfast && !fft_lds && (kp.need_eb || kp.need_rb);
            if (oc) {
                oc = uni(oc_rank_phase<T>(kp, xp, n, smem, ntgt)) != 0;
                if (oc) rk = 5;
            }
        }
        if (need_rank_x && !oc) {
            if (kp.rank_variant == 2) {
                rk = rank_phase_q(kp, VX, n, smem, zb, pbuf, bb, xs, bs, ntgt, kp.mode == 3,
                                  // order statistics only (no score is consumed): estimators, mcse quantile, ess(kind = tail);
                                  // a tail-only request that needs R-hat as well ranks every round for the SORTED COPY (5), so that
                                  // its folded ranking is the merge of fold_merge_phase and not a second full ranking (round 6:
                                  // ess_rhat(kind = tail) 9.0 -> 6.6 ms at config 2's shape; same bits either way)
                                  (est_ind || kp.mode == 5) ? 4
                                      : (!(kp.need_eb || kp.need_rb) && kp.mode != 3) ? ((kp.need_rt && kp.fold_merge && !many && kp.tail_srt) ? 5 : 4)
                                      // nested R-hat only needs the moments (and has no ess_rhat twin whose bits it must match)
                                      : ((nested && !kp.need_eb) ? 2 : (seg_req ? 1 : (long_chain ? ((kp.need_eb && kp.ess_ok) ? 3 : 2) : 0))),
                                  // the folded ranking that follows is a merge of the sorted slice (fold_merge_phase)
                                  kp.need_rt && kp.mode != 3 && kp.fold_merge && !many); // (many chains: the sorted copy's scratch holds their moments)
                if (!rk) rank_phase_p(kp, VX, n, smem, zb, pbuf, bb, xs, bs, ntgt, kp.mode == 3);
            } else if (kp.rank_variant == 1) rank_phase_p(kp, VX, n, smem, zb, pbuf, bb, xs, bs, ntgt, kp.mode == 3);
            else if (pos16) rank_phase_b<unsigned short>(kp, VX, n, smem, zb, (unsigned short*)pbuf, bb, ntgt, kp.mode == 3);
            else rank_phase_b<unsigned>(kp, VX, n, smem, zb, pbuf, bb, ntgt, kp.mode == 3);
        }
        // the ranking left the slice sorted, tags in sorted order, no NaN (rank_phase_q, srt)
        const bool sorted_ok = need_rank_x && kp.rank_variant == 2 && rk != 0 && uni(((int*)(smem + OFF_INTS))[18]) != 0;
        MDT_PROF_MARK(PH_RANKX);
        MDT_STOP(1)
        MDT_STOP(10)
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
        // tail-ESS indicators from the z-series (series_z_phase, gbits): x <= q  <=>  x <= (largest order statistic <= q)
        // <=> z <= its score.  q lies between the two captured neighbours a <= b: that statistic is a (q < b) or b (q == b);
        // anything else (rounding past b, NaN) leaves the decision to the comparisons on x themselves
        double zq[2] = {0.0, 0.0};
        bool zq_ok = need_q && !has_nan && kp.need_et && kp.est == 0;
        if (zq_ok) {
            const int t0 = need_med ? 2 : 0;
            const unsigned* orank = (const unsigned*)(smem + OFF_ORK);
            const T qq2[2] = {qloT, qhiT};
            for (int i = 0; i < 2; ++i) {
                const T a = (T)ostat[t0 + 2 * i], b = (T)ostat[t0 + 2 * i + 1];
                const int pick = (qq2[i] >= a && qq2[i] < b) ? 0 : ((qq2[i] == b) ? 1 : -1);
                if (pick < 0) { zq_ok = false; break; }
                const unsigned r2 = orank[t0 + 2 * i + pick];
                int zero = 0;
                asm volatile("" : "+v"(zero));
                zq[i] = ztab[((r2 & 1u) ? r2 : (unsigned)(2 * n + 1) + (r2 >> 1)) + zero];
            }
            zq_ok = uni(zq_ok);
        }
        unsigned long long* gb = kp.ibits + (long long)blockIdx.x * kp.ibits_stride;
        bool bits_done = false;
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
                else if (many) ee = many_phase(kp, SerInd<T>{(const MDT_GA T*)xp, D, thr, medT, fold}, xs, true, smem).ess;
                else if (kp.niter > (long long)BITW * 64) // (a split chain beyond the bit vectors of the indicator phases: the proxy, tiled)
                    ee = basic_ess_rhat(SerInd<T>{(const MDT_GA T*)xp, D, thr, medT, fold}, true).ess;
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
        bool rt_done = false;
        if constexpr (OC_ON) {
            // one-chunk path: the folded ranking comes FIRST, as a merge of the sorted copy that still sits in LDS (the bulk
            // series then takes its place); when it cannot serve the slice (ties in bulk) the general folded ranking runs
            // after the bulk series, as everywhere else
            if (rk == 5 && kp.need_rt && (medT - medT == (T)0)) {
                if (uni(oc_fold_phase<T>(kp, medT, n, smem, bb)) == 3) {
                    if (tid == 0) series_combine(kp, smem, 0, 1, false);
                    __syncthreads();
                    rt = scal[2]; // (the next write to it lies behind the barriers of the series phase)
                    rt_done = true;
                }
                MDT_PROF_MARK(PH_RANKF);
            }
        }
        if (kp.need_eb || kp.need_rb) {
            const bool acov = kp.need_eb && kp.ess_ok;
            SeriesOut so;
            // the moments of a z-series always come from series_z_phase (fixed point): R-hat is the same
            // bits whichever autocovariance path follows
            if (many) { // time-order scores, one thread per split chain
                so = many_phase(kp, YZ, xs, acov, smem, nested ? 1 : 0);
            } else if (rk == 5) { // one-chunk path: every split chain is resident in LDS, one sweep over the pooled series
                if constexpr (OC_ON) {
                    bits_done = zq_ok && acov && 2LL * kp.m * ((kp.niter + 63) >> 6) <= kp.ibits_stride;
                    so = oc_series_phase(kp, acov, smem, bits_done ? gb : nullptr, zq[0], zq[1]);
                    if (acov) so.ess = fft_const_chain_nan(so.ess);
                }
            } else if (rk == 4) { // long chains: moments from the ranking, lag products from its windowed score layout
                bits_done = zq_ok && 2LL * kp.m * ((kp.niter + 63) >> 6) <= kp.ibits_stride;
                so = series_win_phase(kp, zb, pbuf, bb, smem, bits_done ? gb : nullptr, zq[0], zq[1]);
                so.ess = fft_const_chain_nan(so.ess);
            } else if (rk == 3) { // the ranking delivered mean_j / var_j itself (nested R-hat, long chains without ESS)
                so.ess = nan;
                if (tid == 0) series_combine(kp, smem, 0, 1, false);
                __syncthreads();
                so.rhat = scal[2];
                __syncthreads();
            } else if (acov && fft_lds) {
                (void)series_z_phase<false>(kp, zb, nullptr, false, smem);
                so = series_fft_phase(kp, YZ, smem);
            } else if (zfast || !acov) {
                // (with the autocovariance every chain passes through LDS in time order: the indicator bits are taken there)
                bits_done = zq_ok && acov && kp.ess_ok && 2LL * kp.m * ((kp.niter + 63) >> 6) <= kp.ibits_stride;
                unsigned long long* gq = bits_done ? gb : nullptr;
                so = (rk == 2) ? series_z_phase<true>(kp, zb, (const unsigned short*)pbuf, acov, smem, gq, zq[0], zq[1])
                               : series_z_phase<false>(kp, zb, nullptr, acov, smem, gq, zq[0], zq[1]);
                if (acov) so.ess = fft_const_chain_nan(so.ess);
            } else {
                (void)series_z_phase<false>(kp, zb, nullptr, false, smem);
                so = series_phase(kp, YZ, true, smem, true);
                so.ess = fft_const_chain_nan(so.ess);
            }
            rb = so.rhat;
            if (acov) eb = so.ess;
            if (nested && !many) rb = nested_combine_wg(kp, mean_j, var_j, smem);
        }
        MDT_PROF_MARK(PH_SERB);
        MDT_STOP(2)
        if (kp.need_rt && !rt_done) {
            rk = 0;
            if (sorted_ok && (medT - medT == (T)0)) rk = fold_merge_phase<T>(kp, medT, n, smem, xs, bs, bb);
            if (rk != 3) rank_folded(0, kp.mode != 3 ? 2 : 0);
            MDT_PROF_MARK(PH_RANKF);
            // R-hat of the folded scores needs the split-chain moments only: the two-level ranking delivers them
            // itself (rk == 3, no score was ever written); otherwise one streaming sweep over the scores
            if (many) {
                rt = many_phase(kp, YZ, xs, false, smem, nested ? 1 : 0).rhat;
            } else if (rk == 3) {
                if (tid == 0) series_combine(kp, smem, 0, 1, false);
                __syncthreads();
                rt = scal[2];
                __syncthreads();
            } else {
                rt = series_z_phase<false>(kp, zb, nullptr, false, smem).rhat;
            }
            if (nested && !many) rt = nested_combine_wg(kp, mean_j, var_j, smem);
            MDT_PROF_MARK(PH_SERT);
        }
        MDT_STOP(3)
        if (kp.need_et && kp.ess_ok) {
            if (qloT != qloT || qhiT != qhiT) { et = nan; if (tid == 0) atomicOr(kp.status, 2); }
            else if (many) { // min over the two tails of the basic ESS of the indicator (src/ess_rhat.jl:301-311)
                const double e0 = many_phase(kp, SerInd<T>{(const MDT_GA T*)xp, D, qloT, (T)0, 0}, xs, true, smem).ess;
                const double e1 = many_phase(kp, SerInd<T>{(const MDT_GA T*)xp, D, qhiT, (T)0, 0}, xs, true, smem).ess;
                et = jl_min(e0, e1);
            } else if (kp.niter > (long long)BITW * 64) { // (a split chain beyond the bit vectors of the indicator phases: the proxies, tiled)
                const double e0 = basic_ess_rhat(SerInd<T>{(const MDT_GA T*)xp, D, qloT, (T)0, 0}, true).ess;
                const double e1 = basic_ess_rhat(SerInd<T>{(const MDT_GA T*)xp, D, qhiT, (T)0, 0}, true).ess;
                et = jl_min(e0, e1);
            } else if (n <= (long long)IND_MAXW * 64 && 2 * (((kp.niter + 63) >> 6) + (kp.maxlag >> 6) + 2) <= SB2_WORDS)
                et = indicator_phase_pooled<T>(kp, xp, qloT, qhiT, smem, 0, (T)0, bits_done ? gb : nullptr);
            else et = indicator_phase<T>(kp, xp, (double)qloT, (double)qhiT, smem, 0, (T)0, bits_done ? gb : nullptr);
            MDT_PROF_MARK(PH_IND);
        }
        if (((int*)(smem + OFF_INTS))[19]) { eb = nan; rb = nan; et = nan; rt = nan; } // ranking tables overflowed (status bit 0)
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
