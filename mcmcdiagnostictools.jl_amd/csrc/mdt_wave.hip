// mdt_wave.hip -- short slices: ONE PARAMETER PER WAVE, no workgroup barrier after the start-up table load.
//
// What a sampler emits -- 1000 draws x 4 chains, BASELINE config 1's shape -- is a slice of a few thousand values.  On the
// workgroup-per-parameter kernels (mdt_device.hpp) such a parameter is a chain of about forty barrier-separated steps, each a
// few dependent LDS or L2 round trips: 2.6 % of the HBM roof at 1000 x 4, 1.5 % at 500 x 4.  Here a
// wave64 owns a parameter from its ticket to its four outputs and synchronises with nobody: waves of a CU drift apart, and
// whenever one waits for memory or the LDS the others issue.  What makes that possible is that nothing of the size of the
// slice lives in LDS: a wave holds 10.9 KB (one ROUND of the ranking, one fold tile, one split chain; 13.1 KB for split chains of
// more than 640 draws), the slice itself
// passes through a per-wave scratch in global memory that stays in the L2 / Infinity Cache, and the normal-score table of
// the shape (src/utils.jl:182, 189-193) is shared by the waves of a workgroup, read-only, in LDS (up to 4096 values; above, the series
// phase reads it from its global copy and the workgroup keeps its twelve waves).
//
// Per parameter (reference: src/ess_rhat.jl:517-594, src/utils.jl:148-193):
//   sample     128 values spread over the slice, sorted by the wave (bitonic, shuffles): 127 splitters, 128 classes
//   pass A     class of every value (branch-free binary search in LDS), class histogram, min / max, non-finite check
//   rounds     maximal runs of classes with at most WR members; exact class starts by prefix
//   pass B     every value to its class's segment of the scratch (LDS cursor per class): the slice is class-sorted
//   round      the members of a round are ONE contiguous piece of the scratch: piecewise-linear fine bins inside every class
//              (sub-bin 0 = equal to the class's lower splitter: point masses are constant bins), LDS histogram + prefix,
//              placement, all-pairs count inside the bin (~1 member): pooled tied rank (StatsBase.tiedrank), order statistics
//              (median, type-7 quantile neighbours), the sorted copy (in place, over the round's piece of the scratch) and the
//              doubled rank of every member appended to the rank list of its split chain (draw | rank: semi-coalesced stores)
//   fold       rhat(kind = tail): |x - median| of the sorted copy is two monotone runs mirrored at the median; their tied
//              ranks are those of a two-way merge (merge path per lane inside a 512-output tile staged in LDS); scores by
//              merged position (coalesced), exact fixed-point moments per split chain by LDS atomics
//   series     per split chain: rank list -> time order (LDS) -> scores (shared table) -> stage in LDS, exact fixed-point moments
//              (the same integers and expressions as every other path: R-hat is the same BITS), centred, lag products as
//              blocked Hankel tiles on v_mfma_f64_16x16x4_f64 accumulated over the chains; Geyer's scan between lag blocks
//   indicators tail ESS: ballots of x <= q per split chain, popcount autocovariance, exact integer centring, two scans
// Parameters this path cannot serve -- a NaN or an infinity, a point mass larger than a round, more than WTCAP ties among
// the folded values -- are appended to a list; the host queues the general kernel on that list behind this launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "mdt_wave.h"

namespace mdtw {
using mdtc::KP;

#define WV_GA __attribute__((address_space(1)))
#define WV_CA __attribute__((address_space(4)))

constexpr int NS = 128;            // sample values; NS - 1 splitters
constexpr int NCLS = 128;          // classes: #{splitters <= x}
constexpr int WR = 512;            // members of a round (the LDS image).  (448 = seven rows lets a twelfth wave per CU in at 4096 values, and the extra
                                   // rounds cost what the wave brings: 10.0 against 9.7 - 9.9 ms at 1000 x 4 x 100 000 -- measured in round 5, not kept)
constexpr int RROWS = WR / 64;
constexpr int NBIN = 512;          // fine bins of a round
constexpr int MAXROUNDS = 64;
constexpr int FT = 512;            // merged outputs of a fold tile (8 per lane)
constexpr int WTCAP = 64;          // folded members with equals kept for the exact pass
constexpr int ZT = 3;              // Hankel tiles per sweep: 32 lags
constexpr int STG = 64 * ((WV_NITER + 63) / 64) + 272; // doubles of the time-order stage: chain + zero halo for the last tile's shifted operand
constexpr int NTHREADS = 768;      // at most 12 waves per workgroup (3 per SIMD: 168 VGPRs)

// ---- per-wave LDS (bytes from the wave's base)
constexpr int OFF_SP = 0;                              // double sp[130]: sp[0] = min, sp[1 + i] = splitter i (i < 127), sp[128] = +inf
constexpr int OFF_IW = OFF_SP + 130 * 8;               // float iw[128]: 1 / width of every class (only the fine bins of a round are cut with it: any monotone cut ranks the same)
constexpr int OFF_LAGA = OFF_SP;                       // (series phase, over the dead splitters) double lagA[256]
constexpr int OFF_CCUR = OFF_IW + 128 * 4;             // unsigned ccur[64]: class histogram, then cursors, as packed 16-bit pairs (<= 8192 members); after pass B, read as 16-bit entries, the END of every class
constexpr int OFF_RTAB = OFF_CCUR + 64 * 4;            // unsigned short rtab[MAXROUNDS + 2]: first class of every round
constexpr int OFF_OST = ((OFF_RTAB + (MAXROUNDS + 2) * 2 + 7) / 8) * 8; // double ostat[8]
constexpr int OFF_ORK = OFF_OST + 8 * 8;               // unsigned orank[8]: doubled tied rank of every captured order statistic
constexpr int OFF_LCUR = OFF_ORK + 8 * 4;              // unsigned lcur[16]: fill of every split chain's rank list
constexpr int OFF_JBASE = OFF_LCUR + WV_MAXM * 4;      // unsigned short jbase[16]: pooled index of every split chain's first draw
constexpr int OFF_MEAN = OFF_JBASE + WV_MAXM * 2;      // double mean_j[16], var_j[16], mu_j[16]
constexpr int OFF_WS = OFF_MEAN + 3 * WV_MAXM * 8;     // struct WS: what the phases of a parameter hand to each other (wave-uniform)
constexpr int WS_BYTES = 192;
constexpr int OFF_PROF = OFF_WS + WS_BYTES;            // (diagnostics build WV_PROF) unsigned long long prof[16]: cycles per phase
#ifdef WV_PROF
constexpr int OFF_VIEW = ((OFF_PROF + 25 * 8 + 15) / 16) * 16;
#else
constexpr int OFF_VIEW = ((OFF_PROF + 15) / 16) * 16;
#endif
// (iw as floats and the cursors packed: 768 bytes of the 704 that a twelfth wave per CU needs beside the 32 KB table of a 4000-value slice --
// 11 -> 12 waves is worth 4.3 %, measured on 724 x 4 where both fit: 10.45 -> 10.0 ms)
static_assert(OFF_LAGA + 256 * 8 <= OFF_MEAN, "lag table lives over the tables that are dead once the rounds are done (splitters, widths, cursors, round table, order statistics, list fills)");
// rank view
constexpr int OFF_VAL = OFF_VIEW;                      // double val[WR]; during passes A / B: uint8 cls[WV_MAXN]
constexpr int OFF_TAG = OFF_VAL + WR * 8;              // unsigned tag[WR]
constexpr int OFF_BIN = OFF_TAG + WR * 4;              // unsigned short binid[WR]: bin of every slot | reserved-bin flag << 15
constexpr int OFF_CNT = OFF_BIN + WR * 2;              // unsigned cnt[NBIN / 2]: packed 16-bit counters, then cursors (read as 16-bit entries: the end of every bin)
constexpr int END_RANK = OFF_CNT + NBIN * 2;
static_assert(WV_MAXN <= END_RANK - OFF_VAL, "the time-order class bytes of passes A / B live in the bytes of the round tables");
// fold view
constexpr int FE = FT + 4;                             // staged entries of a tile: for each piece the member in front of it, its members, the one behind
constexpr int OFF_FA = OFF_VIEW;                       // double fv[FE]
constexpr int OFF_SA = OFF_FA + FE * 8;                // uint8 fs[FE]: their split-chain slots
constexpr int OFF_FACC = OFF_SA + ((FE + 7) / 8) * 8;  // int64 facc[2][16][8]
constexpr int OFF_TOR = OFF_FACC + 2 * WV_MAXM * 8 * 8; // uint64 tor[16], tand[16]
constexpr int OFF_TCNT = OFF_TOR + 2 * WV_MAXM * 8;    // unsigned tcnt[16], tln, pad
constexpr int OFF_TB = OFF_TCNT + (WV_MAXM + 2) * 4;   // unsigned tb[WV_MAXN / FT + 2]: members of A in front of every tile
constexpr int OFF_TVAL = ((OFF_TB + (WV_MAXN / FT + 2) * 4 + 7) / 8) * 8; // double tval[WTCAP]
constexpr int OFF_TSL = OFF_TVAL + WTCAP * 8;          // uint8 tsl[WTCAP]
constexpr int END_FOLD = OFF_TSL + WTCAP;
// series view: the stage, and behind it (when both fit) the indicator words taken from the ranks while a chain is staged
constexpr int OFF_STG = OFF_VIEW;                      // double stage[64 nblk + 272] (and the 16 x 17 tile staging of a sweep's end)
constexpr int END_SER = OFF_STG + STG * 8;
// indicator view
constexpr int NWU_MAX = (WV_NITER + 63) / 64 + (WV_MAXLAG >> 6) + 2;
constexpr int OFF_ICOV = OFF_VIEW;                     // double icov[2][256]
constexpr int OFF_IMEAN = OFF_ICOV + 2 * 256 * 8;      // double imean[2][16]
constexpr int OFF_SB2 = OFF_IMEAN + 2 * WV_MAXM * 8;   // uint64 sb2[2 m nwU] (the unfused form; the fused one sits behind the stage)
constexpr int SB2_WORDS = 2 * (WV_MAXN / 64 + 6 * WV_MAXM);  // 2 m nwU with nwU <= niter / 64 + 6 and m niter <= WV_MAXN
constexpr int END_IND = OFF_SB2 + SB2_WORDS * 8;
constexpr int cmax2(int a, int b) { return a > b ? a : b; }
constexpr int PERWAVE = ((cmax2(cmax2(END_RANK, END_FOLD), cmax2(END_SER, END_IND)) + 15) / 16) * 16;   // split chains of up to WV_NITER draws
constexpr int END_SER_SHORT = OFF_STG + (64 * ((WV_NITER_SHORT + 63) / 64) + 272) * 8;
constexpr int PERWAVE_SHORT = ((cmax2(cmax2(END_RANK, END_FOLD), cmax2(END_SER_SHORT, END_IND)) + 15) / 16) * 16; // ... of up to WV_NITER_SHORT: what a sampler's 1000 x 4 needs, 12 waves beside its table
constexpr int LDS_TOTAL = 160 * 1024;

// one member of the slice in the per-wave scratch, 12 bytes (one store / one load per member): its value and a tag -- in the
// class-sorted copy element | split-chain slot << 16 | class << 24, in the sorted copy the split-chain slot
struct __attribute__((packed, aligned(4))) Rec { double v; unsigned t; };
static_assert(sizeof(Rec) == 12, "12-byte scratch records");
typedef unsigned u3v __attribute__((ext_vector_type(3)));
// one 12-byte store per record (global_store_dwordx3: the compiler would split the packed struct into an 8- and a 4-byte
// store, i.e. two requests per lane of a scattered store); ordered against later loads by mem_fence() like every store here
__device__ __forceinline__ void rec_store(__attribute__((address_space(1))) Rec* p, double v, unsigned t) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    u3v w;
    w.x = (unsigned)b; w.y = (unsigned)(b >> 32); w.z = t;
    asm volatile("global_store_dwordx3 %0, %1, off" : : "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ Rec rec_load(const __attribute__((address_space(1))) Rec* p) {
    Rec r;
    r.v = p->v;
    r.t = p->t;
    return r;
}
struct WvArgs {
    Rec* rec;              // [waves][npad]: the slice class-sorted, then (in place, round by round) sorted
    unsigned* ra;          // [waves][npad]: per split chain the list of (draw | doubled tied rank << 16) of its members, in the order the rounds rank them
    unsigned* fb_list;     // parameters left to the general kernel
    int* fb_count;
    long long npad;
    int tab_bytes;
    int perwave;           // bytes of LDS per wave (PERWAVE_SHORT or PERWAVE)
    unsigned* dbg;         // diagnostics (MDT_WAVE_DBG = address of a host-pinned array): progress word of every wave, or null
    int stop;              // diagnostics (MDT_WAVE_STOP behind the test-hooks gate): the parameter ends after phase `stop` with NaN outputs; 0 = never
    int osonly;            // 1: ess(kind = tail) alone ranks only the rounds that hold its quantiles (MDT_WAVE_OSONLY=0: the full ranking; tests, A/B timing)
};

// What the phases of one parameter hand to each other: wave-uniform values in the wave's LDS block.  The phases are separate,
// non-inlined functions (own register allocation: one function of this size spilled and turned every uniform value into a
// lane of a spill register); their only argument is the LDS address of the wave's block.
struct WS {
    unsigned long long kp;            // const KP* (read through the constant address space)
    unsigned long long rec, ra;       // the wave's scratch
    unsigned long long xp;            // the parameter's slice
    unsigned long long dbg;           // progress words, or 0
    long long p;                      // the parameter
    double xmin, xmax;
    double med, qlo, qhi;             // median and the two tail quantiles (values of T)
    double fmed;                      // ess(kind = mad): the median of x while the second ranking runs on |x - fmed| (foldon)
    double out[4];                    // ESS bulk, R-hat bulk, ESS tail, R-hat tail
    unsigned nrounds, anyodd;         // rounds of the ranking; some rank is a half-integer
    unsigned r2q[2];                  // x <= q  <=>  doubled rank <= r2q (when zq_ok)
    int fallback, zq_ok, bits_done, stop;
    unsigned lz, gw;                  // LDS address of the score table; global wave index
    unsigned tab_lds;                 // the score table of whole ranks sits in LDS (else: read from the global copy)
    unsigned view_bytes;              // bytes of the wave's block behind OFF_VIEW
    unsigned osonly;                  // ess(kind = tail) alone: nothing but the two quantile pairs is wanted of the ranking (round 6) -- pass B stores
                                      // only the classes of the rounds that hold them, only those rounds are ranked, the indicators come from x
    unsigned foldon;                  // the passes over x (classify, partition, indicators) read |T(x - fmed)| instead of x
};
static_assert(sizeof(WS) <= WS_BYTES, "wave state block");
#define WV_LA __attribute__((address_space(3)))

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned rflu(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ double rfld(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = rflu((unsigned)b), hi = rflu((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// same-wave exchange through LDS / global memory: the hardware keeps a wave's LDS (and vector memory) operations in order;
// what has to be stopped is the compiler moving a read over the write of another lane
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void mem_fence() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_add_u32(unsigned v) {
    return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
    v = dpp_add_u32<0x111, 0xF>(v); // row_shr:1
    v = dpp_add_u32<0x112, 0xF>(v); // row_shr:2
    v = dpp_add_u32<0x114, 0xF>(v); // row_shr:4
    v = dpp_add_u32<0x118, 0xF>(v); // row_shr:8
    v = dpp_add_u32<0x142, 0xA>(v); // row_bcast:15
    v = dpp_add_u32<0x143, 0xC>(v); // row_bcast:31
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_add_i64(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xF, true);
    return v + (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) { // total in every lane
    v = dpp_add_i64<0x111, 0xF>(v);
    v = dpp_add_i64<0x112, 0xF>(v);
    v = dpp_add_i64<0x114, 0xF>(v);
    v = dpp_add_i64<0x118, 0xF>(v);
    v = dpp_add_i64<0x142, 0xA>(v);
    v = dpp_add_i64<0x143, 0xC>(v);
    const int lo = __builtin_amdgcn_readlane((int)v, 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// The wave-wide double sum of mdt_device.hpp (wave_sum there), instruction for instruction: the same association order is what makes
// the raw-draw moments below the same bits as the workgroup-per-parameter kernels'.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, true);
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) { // total in every lane
    v = dpp_add_d<0x111, 0xF>(v); // row_shr:1
    v = dpp_add_d<0x112, 0xF>(v); // row_shr:2
    v = dpp_add_d<0x114, 0xF>(v); // row_shr:4
    v = dpp_add_d<0x118, 0xF>(v); // row_shr:8
    v = dpp_add_d<0x142, 0xA>(v); // row_bcast:15
    v = dpp_add_d<0x143, 0xC>(v); // row_bcast:31
    return readlane_d(v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ double jl_min(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }
__device__ __forceinline__ double jl_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }
// The fixed-point images of mdt_device.hpp (struct Fx there): a shifted normal score u (|u| < 32) plus 1.5 * 2^6 leaves u in
// units of 2^-46 in the mantissa (u^2 < 256: 1.5 * 2^9, units of 2^-43); int64 sums of them are exact and order-free.  The
// SAME constants and expressions as there -- that is what makes R-hat the same bits on every path.
struct Fx {
    double m1, m2, i1, i2;
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
template <typename T>
using xs_of = typename std::conditional<std::is_same<T, float>::value, float, double>::type;
typedef double d4 __attribute__((ext_vector_type(4)));

// LDS atomics on address-space-3 pointers (ds_add_rtn_u32 / ds_add_u64 ...; HIP's atomicAdd takes generic pointers only)
template <class P, class V> __device__ __forceinline__ V lds_add(P p, V v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class P, class V> __device__ __forceinline__ V lds_or(P p, V v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class P, class V> __device__ __forceinline__ V lds_and(P p, V v) { return __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename T>
__device__ __forceinline__ void store_out(void* p, long long i, double v) {
    if (p) ((T*)p)[i] = (T)v;
}

// W, var+, R-hat from the split-chain means and variances (src/ess_rhat.jl:537-545): the expressions of series_combine
// (mdt_series.inc), every lane the same value
struct WB { double W, varp, rhat; };
template <class P>
__device__ __forceinline__ WB combine_moments(P mean_j, P var_j, int m, double cf) {
    double W = 0.0, mb = 0.0;
    for (int j = 0; j < m; ++j) { W += var_j[j]; mb += mean_j[j]; }
    W /= (double)m;
    mb /= (double)m;
    double vb = 0.0;
    for (int j = 0; j < m; ++j) { const double d = mean_j[j] - mb; vb = fma(d, d, vb); }
    vb /= (double)(m > 1 ? m - 1 : m);
    WB o;
    o.W = W;
    o.varp = cf * W + vb;
    o.rhat = sqrt(o.varp / W);
    return o;
}


// ------------------------------------------------------------------ phase plumbing
__device__ __forceinline__ unsigned long long rfl64(unsigned long long v) {
    const unsigned lo = rflu((unsigned)v), hi = rflu((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
#define WV_PHASE template <typename T> __device__ __attribute__((noinline))
// every phase starts with this: the wave's LDS block as an address-space-3 pointer held in a scalar register, its state
// block, the parameter block through the constant address space (scalar loads)
#define WV_ENTER(wbase_arg)                                                                          \
    const unsigned wbase = rflu(wbase_arg);                                                          \
    WV_LA unsigned char* wb = (WV_LA unsigned char*)(unsigned long long)wbase;                       \
    WV_LA WS* ws = (WV_LA WS*)(wb + OFF_WS);                                                         \
    const WV_CA KP& kp = *(const WV_CA KP*)rfl64(ws->kp);                                            \
    const int lane = (int)(threadIdx.x & 63u);                                                       \
    (void)lane; (void)kp
#ifdef WV_PROF
#define WPROF(slot) do { if (lane == 0) { WV_LA unsigned long long* pr_ = (WV_LA unsigned long long*)(wb + OFF_PROF); const unsigned long long t_ = (unsigned long long)__builtin_readcyclecounter(); pr_[slot] += t_ - pr_[24]; pr_[24] = t_; } __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define WPROF(slot) do { } while (0)
#endif

// row r of the slice (rows of 64 draws; the last row of a chain is partly masked): pooled index of the chain's first draw,
// this lane's draw, and the (clamped) element offset from the slice
struct RowAt { unsigned eb, d; long long ad; };
__device__ __forceinline__ RowAt row_at(unsigned r, unsigned rpc, unsigned D, long long cs, int lane) {
    const unsigned c = r / rpc, i = r - c * rpc;
    RowAt o;
    o.eb = c * D;
    o.d = 64u * i + (unsigned)lane;
    o.ad = (long long)c * cs + (o.d < D ? o.d : D - 1u);
    return o;
}
constexpr int UB = 8; // rows per batch of the streaming passes

// StatsBase.mad's proxy (src/ess_rhat.jl:643-646): |x - median(x)| in T, the expression of SerInd in mdt_device.hpp
template <typename T> __device__ __forceinline__ T fold_T(T v, T med, bool on) {
    if (on) { const T d = (T)(v - med); v = d < (T)0 ? -d : d; }
    return v;
}

// 0-based sorted places of the order statistics a request wants captured: the median pair first (rhat(kind = tail)'s fold;
// ess(kind = median), round 6), then the quantile pairs (ess(kind = tail): both; ess(kind = quantile(p)): the same pair twice)
__device__ __forceinline__ int wv_targets(const WV_CA KP& kp, unsigned n, long long p, unsigned (&tp)[6], bool& need_med, bool& need_q) {
    need_med = kp.need_rt != 0 || kp.est == 4 || kp.est == 6;
    need_q = (kp.need_et || kp.est == 7) && kp.ess_ok;
#pragma unroll
    for (int t = 0; t < 6; ++t) tp[t] = 0u;
    if (kp.mode == 5) { // second pass of mcse(kind = median | quantile(p)), src/mcse.jl:109-117: the parameter's own two places
        need_med = false; need_q = false;
        tp[0] = kp.tpos_arr[2 * p]; tp[1] = kp.tpos_arr[2 * p + 1];
        return 2;
    }
    const unsigned ql0 = (unsigned)(kp.q_j[0] - 1), ql1 = (unsigned)kp.q_j[0], qh0 = (unsigned)(kp.q_j[1] - 1), qh1 = (unsigned)kp.q_j[1];
    if (need_med && need_q) { tp[0] = (n - 1u) / 2u; tp[1] = n / 2u; tp[2] = ql0; tp[3] = ql1; tp[4] = qh0; tp[5] = qh1; return 6; }
    if (need_med) { tp[0] = (n - 1u) / 2u; tp[1] = n / 2u; return 2; }
    if (need_q) { tp[0] = ql0; tp[1] = ql1; tp[2] = qh0; tp[3] = qh1; return 4; }
    return 0;
}

// ------------------------------------------------------------------ sample -> splitters; pass A: classes, histogram, range; rounds
// (FOLD: the slice is read as |T(x - fmed)| -- the second ranking of ess(kind = mad); its own instantiation: as a run-time select in
// the loads it cost every request of the kernel 3 - 7 %)
template <typename T, bool FOLD = false> __device__ __attribute__((noinline)) void ph_classify(unsigned wbase_arg) {
    WV_ENTER(wbase_arg);
    WV_LA double* sp = (WV_LA double*)(wb + OFF_SP);
    WV_LA float* iw = (WV_LA float*)(wb + OFF_IW);
    WV_LA unsigned* ccur = (WV_LA unsigned*)(wb + OFF_CCUR);
    WV_LA unsigned short* rtab = (WV_LA unsigned short*)(wb + OFF_RTAB);
    WV_LA unsigned char* clsb = (WV_LA unsigned char*)(wb + OFF_VAL);
    const WV_GA T* xp = (const WV_GA T*)rfl64(ws->xp);
    const unsigned n = (unsigned)kp.n, D = (unsigned)kp.D, C = (unsigned)kp.C;
    const long long cs = kp.cstride;
    const double inf = __builtin_inf();
    constexpr bool fo = FOLD;
    const T fm = FOLD ? (T)rfld(ws->fmed) : (T)0;
    // ---- 128 values spread over the slice, sorted by the wave
    {
        double a, b;
        {
            const unsigned i0 = (unsigned)lane, i1 = 64u + (unsigned)lane;
            const unsigned q0 = ((2u * i0 + 1u) * n) >> 8, q1 = ((2u * i1 + 1u) * n) >> 8;
            const unsigned c0 = q0 / D, c1 = q1 / D;
            a = (double)fold_T<T>(xp[(long long)c0 * cs + (q0 - c0 * D)], fm, fo);
            b = (double)fold_T<T>(xp[(long long)c1 * cs + (q1 - c1 * D)], fm, fo);
            a = (a - a == 0.0) ? a : inf;
            b = (b - b == 0.0) ? b : inf;
        }
        for (int k = 2; k <= 64; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool lower = (lane & j) == 0;
                const bool upa = (k == 64) ? true : ((lane & k) == 0);
                const bool upb = (k == 64) ? false : ((lane & k) == 0);
                const double oa = __shfl_xor(a, j), ob = __shfl_xor(b, j);
                a = (lower == upa) ? (oa < a ? oa : a) : (oa > a ? oa : a);
                b = (lower == upb) ? (ob < b ? ob : b) : (ob > b ? ob : b);
            }
        }
        { const double lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; }
        for (int j = 32; j > 0; j >>= 1) {
            const bool lower = (lane & j) == 0;
            const double oa = __shfl_xor(a, j), ob = __shfl_xor(b, j);
            a = lower ? (oa < a ? oa : a) : (oa > a ? oa : a);
            b = lower ? (ob < b ? ob : b) : (ob > b ? ob : b);
        }
        sp[1 + lane] = a;
        sp[65 + lane] = (lane == 63) ? inf : b; // 127 splitters; the sentinel ends the search
        ccur[lane] = 0u; // (two classes per word)
        lds_fence();
#pragma unroll
        for (int q = 0; q < 2; ++q) { // inverse widths of the inner classes c = 1 .. 126: [sp[c], sp[c + 1])
            const int c = lane + 64 * q;
            if (c >= 1 && c <= 126) {
                const double wd = sp[c + 1] - sp[c];
                const double v = (wd > 0.0 && wd < inf) ? 1.0 / wd : 0.0;
                const float vf = (float)v;
                iw[c] = (vf < __builtin_inff()) ? vf : 0.0f; // (a class too narrow for a float's range: one fine bin)
            }
        }
    }
    WPROF(1);
    // ---- pass A.  A wave has few partners on its SIMD (about three), so every pass keeps its own loads in flight: batch
    // k + 1 is issued before batch k is worked on.
    const unsigned rpc = (D + 63u) >> 6, NRW = C * rpc;
    auto load_batch = [&](unsigned r0, double (&vv)[UB]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned r = r0 + (unsigned)u;
            vv[u] = (double)fold_T<T>(xp[row_at(r < NRW ? r : NRW - 1u, rpc, D, cs, lane).ad], fm, fo);
        }
    };
    double xmin = inf, xmax = -inf;
    unsigned bad = 0u;
    {
        double vn[UB];
        load_batch(0u, vn);
        for (unsigned r0 = 0; r0 < NRW; r0 += UB) {
            double v[UB];
            unsigned pos[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) { v[u] = vn[u]; pos[u] = 0u; }
            if (r0 + UB < NRW) load_batch(r0 + UB, vn);
#pragma unroll
            for (unsigned step = 64u; step > 0u; step >>= 1) {
                double t[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) t[u] = sp[pos[u] + step];
#pragma unroll
                for (int u = 0; u < UB; ++u) pos[u] += (t[u] <= v[u]) ? step : 0u;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const unsigned r = r0 + (unsigned)u;
                const RowAt ra_ = row_at(r < NRW ? r : NRW - 1u, rpc, D, cs, lane);
                if (r < NRW && ra_.d < D) {
                    bad |= (v[u] - v[u] == 0.0) ? 0u : 1u;
                    xmin = v[u] < xmin ? v[u] : xmin;
                    xmax = v[u] > xmax ? v[u] : xmax;
                    lds_add(&ccur[pos[u] >> 1], 1u << ((pos[u] & 1u) << 4));
                    clsb[ra_.eb + ra_.d] = (unsigned char)pos[u];
                }
            }
        }
    }
    WPROF(2);
    bool fallback = __ballot(bad != 0u) != 0ull;
    unsigned nrounds = 0u;
    if (!fallback) {
        xmin = wave_min_d(xmin);
        xmax = wave_max_d(xmax);
        lds_fence();
        const unsigned hw = ccur[lane], h0 = hw & 0xFFFFu, h1 = hw >> 16;
        const unsigned tot = h0 + h1, incl = wave_incl_scan_u32(tot);
        const unsigned b0 = incl - tot, b1 = b0 + h0;
        const unsigned cmax = rflu(wave_max_u32(h0 > h1 ? h0 : h1));
        if (cmax >= (unsigned)WR) fallback = true; // a point mass (or a crowd) larger than a round
        else {
            // rounds = runs of classes whose starts fall into one stretch of Q places: at most Q - 1 + cmax < WR members
            const unsigned Q = (unsigned)WR - cmax;
            const unsigned r0 = b0 / Q, r1 = b1 / Q;
            const unsigned rprev = (unsigned)__shfl_up((int)r1, 1);
            const unsigned f0 = (lane == 0 || r0 != rprev) ? 1u : 0u, f1 = (r1 != r0) ? 1u : 0u;
            const unsigned nb = f0 + f1, kin = wave_incl_scan_u32(nb), kx = kin - nb;
            nrounds = (unsigned)__builtin_amdgcn_readlane((int)kin, 63);
            if (nrounds > (unsigned)MAXROUNDS) fallback = true;
            else {
                lds_fence();
                ccur[lane] = b0 | (b1 << 16); // (b1 <= n <= 8192: no carry between the halves while pass B counts them up)
                if (f0) rtab[kx] = (unsigned short)(2 * lane);
                if (f1) rtab[kx + f0] = (unsigned short)(2 * lane + 1);
                if (lane == 0) {
                    rtab[nrounds] = (unsigned short)NCLS;
                    sp[0] = xmin;
                    const double w0 = sp[1] - xmin, w1 = xmax - sp[127];
                    const double v0 = (w0 > 0.0 && w0 < inf) ? 1.0 / w0 : 0.0, v1 = (w1 > 0.0 && w1 < inf) ? 1.0 / w1 : 0.0;
                    const float f0 = (float)v0, f1 = (float)v1;
                    iw[0] = (f0 < __builtin_inff()) ? f0 : 0.0f;
                    iw[127] = (f1 < __builtin_inff()) ? f1 : 0.0f;
                }
            }
        }
    }
    if (lane == 0) { ws->fallback = fallback ? 1 : 0; ws->nrounds = nrounds; ws->xmin = xmin; ws->xmax = xmax; }
    lds_fence();
    WPROF(3);
}

// ------------------------------------------------------------------ pass B: the slice, class-sorted, into the scratch
template <typename T, bool FOLD = false> __device__ __attribute__((noinline)) void ph_partition(unsigned wbase_arg) {
    WV_ENTER(wbase_arg);
    WV_LA unsigned* ccur = (WV_LA unsigned*)(wb + OFF_CCUR);
    WV_LA unsigned char* clsb = (WV_LA unsigned char*)(wb + OFF_VAL);
    const WV_GA T* xp = (const WV_GA T*)rfl64(ws->xp);
    WV_GA Rec* rec = (WV_GA Rec*)rfl64(ws->rec);
    const unsigned D = (unsigned)kp.D, C = (unsigned)kp.C;
    const unsigned niter = (unsigned)kp.niter, s = (unsigned)kp.s, rem = (unsigned)kp.rem;
    const long long cs = kp.cstride;
    const unsigned rpc = (D + 63u) >> 6, NRW = C * rpc;
    constexpr bool fo = FOLD;
    const T fm = FOLD ? (T)rfld(ws->fmed) : (T)0;
    auto row0 = [&](unsigned k) -> unsigned { return k * niter + (k < rem ? k : rem); }; // src/utils.jl:29-36
    auto load_batch = [&](unsigned r0, double (&vv)[UB]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned r = r0 + (unsigned)u;
            vv[u] = (double)fold_T<T>(xp[row_at(r < NRW ? r : NRW - 1u, rpc, D, cs, lane).ad], fm, fo);
        }
    };
    // Order statistics only (ess(kind = tail) alone, round 6): the classes of the rounds that hold one of the four wanted places are
    // marked (one byte per class, in the bytes of the moment tables, unused on this route); every member still draws its place -- the
    // class cursors must end where the rounds expect them -- but only the marked classes are stored.
    const bool osonly = rflu(ws->osonly) != 0u;
    WV_LA unsigned char* wcl = (WV_LA unsigned char*)(wb + OFF_MEAN);
    if (osonly) {
        const WV_LA unsigned short* cst = (const WV_LA unsigned short*)(wb + OFF_CCUR); // (before pass B: the START of every class)
        const WV_LA unsigned short* rtab = (const WV_LA unsigned short*)(wb + OFF_RTAB);
        const unsigned nrounds = rflu(ws->nrounds), n_ = (unsigned)kp.n;
        wcl[lane] = 0; wcl[64 + lane] = 0;
        lds_fence();
        if ((unsigned)lane < nrounds) {
            const unsigned cA = rtab[lane], cB = rtab[lane + 1];
            const unsigned g0 = cst[cA], g1 = cB < (unsigned)NCLS ? (unsigned)cst[cB] : n_;
            unsigned t6[6];
            bool nm_, nq_;
            const int nt_ = wv_targets(kp, n_, ws->p, t6, nm_, nq_);
            bool tin = false;
#pragma unroll
            for (int t = 0; t < 6; ++t) tin = tin || (t < nt_ && t6[t] >= g0 && t6[t] < g1);
            if (tin) for (unsigned c = cA; c < cB; ++c) wcl[c] = 1;
        }
        lds_fence();
    }
    double vn[UB];
    load_batch(0u, vn);
    for (unsigned r0 = 0; r0 < NRW; r0 += UB) {
        double v[UB];
        unsigned cl[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned r = r0 + (unsigned)u;
            const RowAt ra_ = row_at(r < NRW ? r : NRW - 1u, rpc, D, cs, lane);
            v[u] = vn[u];
            cl[u] = clsb[ra_.eb + (ra_.d < D ? ra_.d : D - 1u)];
        }
        if (r0 + UB < NRW) load_batch(r0 + UB, vn);
        if (osonly) {
#pragma unroll
            for (int u = 0; u < UB; ++u) cl[u] |= (unsigned)wcl[cl[u]] << 8; // (bit 8: the class is wanted)
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned r = r0 + (unsigned)u;
            const RowAt ra_ = row_at(r < NRW ? r : NRW - 1u, rpc, D, cs, lane);
            if (r < NRW && ra_.d < D) {
                const bool keepit = !osonly || (cl[u] >> 8) != 0u;
                cl[u] &= 0xFFu;
                const unsigned shc = (cl[u] & 1u) << 4;
                const unsigned pos = (lds_add(&ccur[cl[u] >> 1], 1u << shc) >> shc) & 0xFFFFu;
                if (!keepit) continue;
                unsigned k = 0u; // (s - 1 steps, usually one: kept from the loop vectoriser, which interleaved it by two behind a remainder loop)
#pragma clang loop vectorize(disable) interleave(disable)
                for (unsigned kk = 1u; kk < s; ++kk) k += (ra_.d >= row0(kk)) ? 1u : 0u;
                const unsigned c = (r < NRW ? r : NRW - 1u) / rpc;
                const unsigned slot = (ra_.d < row0(k) + niter) ? c * s + k : 255u; // (255: a row the split drops)
                rec_store(rec + pos, v[u], (ra_.eb + ra_.d) | (slot << 16) | (cl[u] << 24));
            }
        }
    }
    mem_fence();
    WPROF(4);
}

// ------------------------------------------------------------------ the rounds: tied ranks, order statistics, sorted copy
WV_PHASE void ph_rounds(unsigned wbase_arg) {
    WV_ENTER(wbase_arg);
    WV_LA double* sp = (WV_LA double*)(wb + OFF_SP);
    WV_LA float* iw = (WV_LA float*)(wb + OFF_IW);
    const WV_LA unsigned short* cend_cls = (const WV_LA unsigned short*)(wb + OFF_CCUR); // (pass B has counted every class's cursor up to its end)
    WV_LA unsigned short* rtab = (WV_LA unsigned short*)(wb + OFF_RTAB);
    WV_LA double* ostat = (WV_LA double*)(wb + OFF_OST);
    WV_LA unsigned* orank = (WV_LA unsigned*)(wb + OFF_ORK);
    WV_LA double* val = (WV_LA double*)(wb + OFF_VAL);
    WV_LA unsigned* tagl = (WV_LA unsigned*)(wb + OFF_TAG);
    WV_LA unsigned short* binid = (WV_LA unsigned short*)(wb + OFF_BIN);
    WV_LA unsigned* cnt = (WV_LA unsigned*)(wb + OFF_CNT);
    WV_GA Rec* rec = (WV_GA Rec*)rfl64(ws->rec);
    WV_GA unsigned* rl = (WV_GA unsigned*)rfl64(ws->ra);
    WV_LA unsigned* lcur = (WV_LA unsigned*)(wb + OFF_LCUR);
    WV_LA unsigned short* jbase = (WV_LA unsigned short*)(wb + OFF_JBASE);
    const unsigned n = (unsigned)kp.n, niter = (unsigned)kp.niter;
    const unsigned nrounds = rflu(ws->nrounds);
    bool need_med, need_q;
    unsigned tp[6]; // 0-based places of the order statistics: median pair first, then the quantile pairs
    const int ntgt = wv_targets(kp, n, ws->p, tp, need_med, need_q);
    const bool osonly = rflu(ws->osonly) != 0u; // (only the rounds that hold a wanted place are ranked; no rank list, no sorted copy)
    const bool need_ra = (kp.need_eb || kp.need_rb || need_q) && !osonly; // ranks in time order: the z-series, and the tail indicators
    if (lane < WV_MAXM) { // per split chain: fill of its rank list; pooled index of its first draw (src/utils.jl:29-36)
        const unsigned s_ = (unsigned)kp.s, rem_ = (unsigned)kp.rem, c = (unsigned)lane / s_, k = (unsigned)lane - c * s_;
        lcur[lane] = 0u;
        jbase[lane] = (unsigned short)(c * (unsigned)kp.D + k * niter + (k < rem_ ? k : rem_));
    }
    lds_fence();
    unsigned anyodd = 0u;
    // members of round r: one contiguous piece [g0, g1) of the class-sorted scratch; the next round's loads are issued as soon
    // as this round's members sit in the LDS image (their registers are free then)
    double v[RROWS];
    unsigned tg[RROWS];
    auto round_span = [&](unsigned r, unsigned& cA, unsigned& cB, unsigned& g0, unsigned& g1) {
        cA = rflu((unsigned)rtab[r]); cB = rflu((unsigned)rtab[r + 1]);
        g0 = cA ? rflu((unsigned)cend_cls[cA - 1u]) : 0u;
        g1 = rflu((unsigned)cend_cls[cB - 1u]);
    };
    auto issue = [&](unsigned r) {
        unsigned cA, cB, g0, g1;
        round_span(r, cA, cB, g0, g1);
        const unsigned M = g1 - g0;
#pragma unroll
        for (int i = 0; i < RROWS; ++i) {
            const unsigned idx = 64u * i + (unsigned)lane;
            const unsigned src = g0 + (idx < M ? idx : 0u);
            const Rec rc = rec_load(rec + (src < n ? src : n - 1u));
            v[i] = rc.v;
            tg[i] = rc.t;
        }
    };
    auto wanted_from = [&](unsigned r) -> unsigned { // the first round >= r that is ranked at all
        if (!osonly) return r;
        for (; r < nrounds; ++r) {
            unsigned cA, cB, g0, g1;
            round_span(r, cA, cB, g0, g1);
            bool tin = false;
#pragma unroll
            for (int t = 0; t < 6; ++t) tin = tin || (t < ntgt && tp[t] >= g0 && tp[t] < g1);
            if (tin) break;
        }
        return r;
    };
    const unsigned rfirst = wanted_from(0u);
    if (rfirst < nrounds) issue(rfirst);
    for (unsigned r = rfirst; r < nrounds;) {
        const unsigned rnext = wanted_from(r + 1u);
        unsigned cA, cB, g0, g1;
        round_span(r, cA, cB, g0, g1);
        const unsigned M = g1 - g0;
        const unsigned ncls = cB - cA;
        const unsigned sh = 31u - (unsigned)__clz((int)((unsigned)NBIN / ncls));
        const double fsub = (double)((1u << sh) - 1u), fcap = (double)((1u << sh) - 2u);
#pragma unroll
        for (int q = 0; q < NBIN / 2 / 64; ++q) cnt[lane + 64 * q] = 0u;
        unsigned bn[RROWS];
        {
            double lo[RROWS], wd[RROWS];
#pragma unroll
            for (int i = 0; i < RROWS; ++i) { const unsigned cls = tg[i] >> 24; lo[i] = sp[cls]; wd[i] = (double)iw[cls]; }
#pragma unroll
            for (int i = 0; i < RROWS; ++i) {
                // piecewise-linear fine bin inside the class; sub-bin 0 = equal to the class's lower bound (a constant bin)
                // (skipping the rows a round does not fill -- scalar branches -- was measured neutral, round 5)
                const unsigned cls = tg[i] >> 24;
                double f = (v[i] - lo[i]) * wd[i] * fsub;
                f = f < fcap ? f : fcap;
                const unsigned sub = (v[i] == lo[i]) ? 0u : 1u + (unsigned)(int)f;
                bn[i] = (((cls - cA) << sh) + sub) | ((sub == 0u ? 1u : 0u) << 15);
            }
        }
#pragma unroll
        for (int i = 0; i < RROWS; ++i) {
            const unsigned b = bn[i] & 0x7FFFu;
            if (64u * i + (unsigned)lane < M) lds_add(&cnt[b >> 1], 1u << ((b & 1u) << 4));
        }
        lds_fence();
        WPROF(10);
        { // exclusive prefix over the packed counters: 8 bins per lane
            unsigned wv[4], s8[8], tot = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = cnt[4 * lane + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const unsigned c8 = (wv[q >> 1] >> ((q & 1) << 4)) & 0xFFFFu; s8[q] = tot; tot += c8; }
            const unsigned incl = wave_incl_scan_u32(tot), bb = incl - tot;
#pragma unroll
            for (int q = 0; q < 4; ++q) cnt[4 * lane + q] = (bb + s8[2 * q]) | ((bb + s8[2 * q + 1]) << 16);
        }
        lds_fence();
        WPROF(11);
#pragma unroll
        for (int i = 0; i < RROWS; ++i) {
            if (64u * i + (unsigned)lane < M) {
                const unsigned b = bn[i] & 0x7FFFu, shb = (b & 1u) << 4;
                const unsigned pos = (lds_add(&cnt[b >> 1], 1u << shb) >> shb) & 0xFFFFu;
                val[pos] = v[i];
                tagl[pos] = tg[i];
                binid[pos] = (unsigned short)bn[i];
            }
        }
        if (rnext < nrounds) issue(rnext);
        lds_fence();
        WPROF(12);
        bool tin = false; // any target order statistic in this round?
#pragma unroll
        for (int t = 0; t < 6; ++t) tin = tin || (t < ntgt && tp[t] >= g0 && tp[t] < g1);
        // dense count in slot order: tied rank, place in sorted order.  The image is sorted by bin; after the placement the packed
        // cursors, read as 16-bit entries, are the END of every bin (and the start of the next one): a slot takes the bounds of
        // its bin from there and compares against the bin's members -- about one -- four at a time; a reserved bin (all members
        // equal: a point mass) needs no compare.  (A version that derived the bounds from start-bit masks with no second round
        // trip to LDS was tried: 64-bit bit scans per lane made it 25 % slower -- this loop is bound by instruction issue; so was
        // all rows of a round step by step with every step's LDS reads in flight together: 10.2 against 9.65 ms, the unfilled
        // rows of a round do their work too and the slots' state spills.)
        const WV_LA unsigned short* cend = (const WV_LA unsigned short*)cnt;
        for (unsigned i = 0; 64u * i < M; ++i) {
            const unsigned sl = 64u * i + (unsigned)lane;
            if (sl < M) {
                const double x = val[sl];
                const unsigned tgv = tagl[sl];
                const unsigned bb = binid[sl], b = bb & 0x7FFFu;
                const unsigned en = cend[b], st_ = b ? (unsigned)cend[b - 1u] : 0u;
                unsigned less, eq, eql = 0u; // members of the round below x; equal to x (x included); equal to x in front of its slot
                if (bb >> 15) {
                    less = st_;
                    eq = en - st_;
                    eql = sl - st_;
                } else {
                    unsigned lt = 0u, eqa = 0u;
                    for (unsigned k0 = st_; k0 < en; k0 += 4u) {
                        double y[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const unsigned k = k0 + (unsigned)j; y[j] = val[k < en ? k : en - 1u]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned k = k0 + (unsigned)j;
                            const bool in = k < en;
                            lt += (in && y[j] < x) ? 1u : 0u;
                            eqa += (in && y[j] == x) ? 1u : 0u;
                            eql += (in && k < sl && y[j] == x) ? 1u : 0u;
                        }
                    }
                    less = st_ + lt;
                    eq = eqa;
                }
                const unsigned gl = g0 + less;
                const unsigned r2 = 2u * gl + eq + 1u; // doubled tied rank: first + last place of the run of equals, 1-based
                anyodd |= r2 & 1u;                      // (a tie group of even size has a half-integer rank: odd doubled value)
                if (tin) {
#pragma unroll
                    for (int t = 0; t < 6; ++t)
                        if (t < ntgt && tp[t] - gl < eq) { ostat[t] = x; orank[t] = r2; }
                }
                const unsigned pos = gl + eql; // equals take the places of their run in slot order
                if (need_ra) {
                    // the rank goes to the list of its split chain (draw inside the chain | doubled rank << 16): lanes of one chain
                    // draw neighbouring places, so the store fills cache lines -- a 2-byte scatter to the time position cost one
                    // request per lane, 13 % of the kernel; the series phase puts the list in time order in LDS
                    const unsigned slot = (tgv >> 16) & 0xFFu;
                    if (slot != 255u) {
                        const unsigned ps = lds_add(&lcur[slot], 1u);
                        rl[slot * niter + ps] = (r2 << 16) | ((tgv & 0xFFFFu) - (unsigned)jbase[slot]);
                    }
                }
                if (!osonly) rec_store(rec + pos, x, (tgv >> 16) & 0xFFu);
            }
        }
        lds_fence();
        WPROF(13);
        r = rnext;
    }
    anyodd = (__ballot(anyodd != 0u) != 0ull) ? 1u : 0u;
    mem_fence();
    // ---- order statistics -> median / quantiles, in T arithmetic like the reference (Statistics.median: a/2 + b/2; type-7
    // quantile: a + g (b - a)); x <= q as a comparison of doubled ranks: q lies between the two captured neighbours a <= b, the
    // largest order statistic <= q is a (q < b) or b (q == b); anything else leaves the decision to the comparisons on x
    T medT = (T)0, qq[2] = {(T)0, (T)0};
    int zq_ok = 0;
    unsigned r2q[2] = {0u, 0u};
    {
        int t = 0;
        if (need_med) {
            const T a = (T)ostat[0], b = (T)ostat[1];
            medT = (n & 1u) ? a : (T)(a / (T)2 + b / (T)2);
            t = 2;
        }
        if (need_q) {
            zq_ok = 1;
            for (int i = 0; i < 2; ++i) {
                const T a = (T)ostat[t + 2 * i], b = (T)ostat[t + 2 * i + 1];
                const T g = (T)kp.q_g[i];
                const bool fin = (a - a == (T)0) && (b - b == (T)0);
                qq[i] = fin ? (T)(a + (T)(g * (T)(b - a))) : (T)((T)(((T)1 - g) * a) + (T)(g * b));
                const int pick = (qq[i] >= a && qq[i] < b) ? 0 : ((qq[i] == b) ? 1 : -1);
                if (pick < 0) zq_ok = 0;
                else r2q[i] = orank[t + 2 * i + pick];
            }
            if (osonly) zq_ok = 0; // (no rank was listed: the indicator phase compares x with the quantiles themselves)
        }
    }
    if (kp.est == 4 || kp.est == 6) { qq[0] = medT; qq[1] = medT; zq_ok = 0; } // ess(kind = median): the indicator of x <= median (src/ess_rhat.jl:630-633), both thresholds the same;
                                                                             // kind = mad: the same of |x - median(x)| once the second ranking has run
    if (kp.mode == 5 && lane == 0) ws->out[2] = (double)(T)(((T)ostat[1] - (T)ostat[0]) / (T)2); // _mcse_quantile: (xu - xl) / 2 in T
    if (lane == 0) {
        ws->anyodd = anyodd; ws->med = (double)medT; ws->qlo = (double)qq[0]; ws->qhi = (double)qq[1];
        ws->zq_ok = zq_ok; ws->r2q[0] = r2q[0]; ws->r2q[1] = r2q[1];
    }
    lds_fence();
    WPROF(5);
}

// ------------------------------------------------------------------ fold: rhat(kind = tail) as a merge of the sorted copy
// The folded values |T(x - med)| of the sorted slice S are two monotone runs mirrored at the median, A[a] = fold(S[imed-1-a])
// and B[b] = fold(S[imed+b]) (rounding is monotone); their pooled tied ranks (src/utils.jl:148-158, 169-193) are those of a
// two-way merge.  Tile boundaries by merge-path searches (one lane per boundary, all at once), then per tile of FT outputs:
// both pieces staged in LDS (the next tile's loads already in flight), a diagonal search and four merge steps per lane, scores
// by merged position (consecutive table entries), exact fixed-point sums per split chain by LDS atomics.  Members with equals
// are set aside and counted exactly afterwards.  Sets ws->fallback when it cannot serve the slice.
WV_PHASE void ph_fold(unsigned wbase_arg) {
    WV_ENTER(wbase_arg);
    WV_LA double* fv = (WV_LA double*)(wb + OFF_FA);            // the tile's two pieces: A piece at 0, B piece at na + 2
    WV_LA unsigned char* fs = (WV_LA unsigned char*)(wb + OFF_SA); // their split-chain slots
    WV_LA long long* facc = (WV_LA long long*)(wb + OFF_FACC);
    WV_LA unsigned long long* tor = (WV_LA unsigned long long*)(wb + OFF_TOR);
    WV_LA unsigned long long* tand = tor + WV_MAXM;
    WV_LA unsigned* tcnt = (WV_LA unsigned*)(wb + OFF_TCNT);
    WV_LA unsigned* tln = tcnt + WV_MAXM;
    WV_LA unsigned* tb = (WV_LA unsigned*)(wb + OFF_TB);
    WV_LA double* tval = (WV_LA double*)(wb + OFF_TVAL);
    WV_LA unsigned char* tsl = (WV_LA unsigned char*)(wb + OFF_TSL);
    WV_LA double* mean_j = (WV_LA double*)(wb + OFF_MEAN);
    WV_LA double* var_j = mean_j + WV_MAXM;
    const WV_GA Rec* rec = (const WV_GA Rec*)rfl64(ws->rec);
    typedef xs_of<T> ZT_t;
    const WV_GA ZT_t* gz = (const WV_GA ZT_t*)kp.ztab;
    const unsigned n = (unsigned)kp.n, niter = (unsigned)kp.niter;
    const int m = kp.m;
    const WV_GA ZT_t* gzd = gz + (2u * n + 1u);
    const double inf = __builtin_inf(), dn = (double)niter;
    const Fx fx = Fx::make(kp.niter);
    const T medT = (T)rfld(ws->med);
    auto fold = [&](double x) -> double {
        const T d = (T)((T)x - medT);
        return (double)(d < (T)0 ? -d : d);
    };
    auto S = [&](unsigned i) -> double { return rec[i].v; }; // the sorted slice
    const unsigned h = (n - 1u) / 2u;
    unsigned zero = 0u;
    asm volatile("" : "+v"(zero));
    const unsigned imed = rflu(((T)S(h + zero) == medT) ? h : h + 1u);
    bool ok;
    {
        const double below = S((imed > 0u ? imed - 1u : 0u) + zero), at = S((imed < n ? imed : n - 1u) + zero);
        ok = (imed == 0u || (T)below <= medT) && (imed == n || (T)at >= medT); // (equals of the median may sit on either side)
        const double f0 = fold(S(zero)), f1 = fold(S(n - 1u + zero));
        ok = ok && f0 < inf && f1 < inf; // (the +inf sentinels behind the runs must beat every member)
        ok = rfl(ok ? 1 : 0) != 0;
    }
    const unsigned nA = imed, nB = n - imed;
    const unsigned ntile = (n + (unsigned)FT - 1u) / (unsigned)FT;
#pragma unroll
    for (int q = 0; q < 2 * WV_MAXM * 8 / 64; ++q) facc[lane + 64 * q] = 0;
    if (lane < WV_MAXM) { tor[lane] = 0ull; tand[lane] = ~0ull; tcnt[lane] = 0u; }
    if (lane == 0) *tln = 0u;
    if (ok) {
        // tb[k] = members of A among the first k FT merged outputs (A first on ties): A[mid] is among the first d outputs iff
        // A[mid] <= B[d - mid - 1], monotone in mid.  One lane per boundary, all boundaries at once.
        if ((unsigned)lane <= ntile) {
            const unsigned d = ((unsigned)lane * (unsigned)FT < n) ? (unsigned)lane * (unsigned)FT : n;
            unsigned lo = d > nB ? d - nB : 0u, hi = d < nA ? d : nA;
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                if (fold(S(imed - 1u - mid)) <= fold(S(imed + (d - mid - 1u)))) lo = mid + 1u; else hi = mid;
            }
            tb[lane] = lo;
        }
        lds_fence();
        WPROF(17);
        const unsigned cp = (unsigned)lane & 7u;
        // A tile's image: entry e < na + 2 = member a0 - 1 + e of A (the member in front of the piece, its na members, the one
        // behind it), then the B piece likewise; -inf in front of a run, +inf behind it.  One record per entry; the next tile's
        // records are in flight while this one is merged.
        constexpr int NR = (FT + 4 + 63) / 64;
        Rec rr[NR];
        auto issue = [&](unsigned k) {
            const unsigned k0 = k * (unsigned)FT, cntk = (n - k0 < (unsigned)FT) ? n - k0 : (unsigned)FT;
            const unsigned a0 = rflu(tb[k]), a1 = rflu(tb[k + 1u]), b0 = k0 - a0, qb = a1 - a0 + 2u;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const unsigned e = 64u * (unsigned)i + (unsigned)lane;
                const bool isA = e < qb;
                const long long mem = isA ? (long long)a0 - 1 + e : (long long)b0 - 1 + (e - qb); // member of its run
                const bool val_ = e < cntk + 4u && mem >= 0 && mem < (long long)(isA ? nA : nB);
                const unsigned src = val_ ? (isA ? (unsigned)((long long)imed - 1 - mem) : (unsigned)((long long)imed + mem)) : 0u;
                rr[i] = rec_load(rec + src);
            }
        };
        issue(0u);
        for (unsigned k = 0; k < ntile; ++k) {
            const unsigned k0 = k * (unsigned)FT;
            const unsigned cntk = (n - k0 < (unsigned)FT) ? n - k0 : (unsigned)FT;
            const unsigned a0 = rflu(tb[k]), a1 = rflu(tb[k + 1u]), b0 = k0 - a0;
            const unsigned na = a1 - a0, nb = cntk - na, qb = na + 2u;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const unsigned e = 64u * (unsigned)i + (unsigned)lane;
                if (e < cntk + 4u) {
                    const bool isA = e < qb;
                    const long long mem = isA ? (long long)a0 - 1 + e : (long long)b0 - 1 + (e - qb);
                    const bool val_ = mem >= 0 && mem < (long long)(isA ? nA : nB);
                    fv[e] = val_ ? fold(rr[i].v) : (mem < 0 ? -inf : inf);
                    fs[e] = (unsigned char)rr[i].t;
                }
            }
            // the scores of this lane's merged positions: position p has rank p + 1 unless it has equals
            constexpr int EPL = FT / 64;
            const unsigned d = (unsigned)EPL * (unsigned)lane;
            double zq[EPL];
#pragma unroll
            for (int i = 0; i < EPL; ++i) { const unsigned pp = k0 + d + (unsigned)i; zq[i] = (double)gzd[(pp < n ? pp : n - 1u) + 1u]; }
            if (k + 1u < ntile) issue(k + 1u);
            lds_fence();
            WPROF(18);
            if (d < cntk) {
                // diagonal: members of the A piece among the first d outputs of the tile (A entry x at fv[x], B entry x at fv[qb + x])
                unsigned lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
                while (lo < hi) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (fv[1u + mid] <= fv[qb + d - mid]) lo = mid + 1u; else hi = mid;
                }
                WPROF(19);
                unsigned a = lo, b = qb + (d - lo); // indices of the entries in front of the two heads
                const double pa = fv[a], pb = fv[b];
                double prev = pa > pb ? pa : pb;
                double ca = fv[1u + a], cb = fv[1u + b];
#pragma unroll
                for (int i = 0; i < EPL; ++i) {
                    const bool ta = ca <= cb;
                    const double cur = ta ? ca : cb;
                    const unsigned is = ta ? 1u + a : 1u + b;
                    const unsigned slot = (unsigned)fs[is];
                    a += ta ? 1u : 0u;
                    b += ta ? 0u : 1u;
                    const double nx = fv[is + 1u];
                    ca = ta ? nx : ca;
                    cb = ta ? cb : nx;
                    const double nxt = ca < cb ? ca : cb;
                    if (d + (unsigned)i < cntk) {
                        if (cur == prev || cur == nxt) { // equals exist: exact count after the tiles
                            const unsigned ti = lds_add(tln, 1u);
                            if (ti < (unsigned)WTCAP) { tval[ti] = cur; tsl[ti] = (unsigned char)slot; }
                        } else if (slot < (unsigned)m) {
                            const double zv = zq[i];
                            lds_add((WV_LA unsigned long long*)&facc[slot * 8u + cp], (unsigned long long)fx.u(zv));
                            lds_add((WV_LA unsigned long long*)&facc[WV_MAXM * 8 + slot * 8u + cp], (unsigned long long)fx.uu(zv * zv));
                        }
                    }
                    prev = cur;
                }
            }
            lds_fence(); // (the next tile's staging stores come after every read of this tile's image)
            WPROF(20);
        }
    }
    unsigned nt = 0u;
    if (ok) { nt = rflu(*tln); if (nt > (unsigned)WTCAP) ok = false; }
    if (!ok) nt = 0u;
    // members with equals: exact tied rank from both runs (binary searches in the sorted copy)
    if ((unsigned)lane < nt) {
        auto count_run = [&](double f, bool arun, bool le) -> unsigned {
            unsigned lo = 0u, hi = arun ? nA : nB;
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                const double x = fold(arun ? S(imed - 1u - mid) : S(imed + mid));
                if (le ? (x <= f) : (x < f)) lo = mid + 1u; else hi = mid;
            }
            return lo;
        };
        const unsigned cp = (unsigned)lane & 7u;
        const double cur = tval[lane];
        const unsigned slot = tsl[lane];
        const unsigned lA = count_run(cur, true, false), lB = count_run(cur, false, false);
        const unsigned eA = count_run(cur, true, true), eB = count_run(cur, false, true);
        const unsigned less = lA + lB, eq = (eA - lA) + (eB - lB);
        const double zv = (double)gz[2u * less + eq + 1u];
        if (slot < (unsigned)m) {
            lds_add((WV_LA unsigned long long*)&facc[slot * 8u + cp], (unsigned long long)fx.u(zv));
            lds_add((WV_LA unsigned long long*)&facc[WV_MAXM * 8 + slot * 8u + cp], (unsigned long long)fx.uu(zv * zv));
            const unsigned long long zb = (unsigned long long)__double_as_longlong(zv);
            lds_or(&tor[slot], zb);
            lds_and(&tand[slot], zb);
            lds_add(&tcnt[slot], 1u);
        }
    }
    lds_fence();
    if (lane < m) { // sums -> moments: the expressions of fold_merge_phase (mdt_fold.inc)
        long long s1 = 0, s2 = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) { s1 += facc[lane * 8 + c]; s2 += facc[WV_MAXM * 8 + lane * 8 + c]; }
        const double S1 = (double)s1 * fx.i1, S2 = (double)s2 * fx.i2;
        const double mu = S1 / dn;
        const bool constant = niter == 1u || (tcnt[lane] == niter && tor[lane] == tand[lane]);
        mean_j[lane] = mu;
        var_j[lane] = constant ? 0.0 : (S2 - mu * S1) / (dn - 1.0);
    }
    lds_fence();
    const double rt = combine_moments(mean_j, var_j, m, kp.cf).rhat;
    if (lane == 0) { ws->out[3] = rt; if (!ok) ws->fallback = 1; }
    lds_fence();
    WPROF(6);
}

// ------------------------------------------------------------------ the std proxy: pooled mean, pooled moments of (x - mean)^2
// ess(kind = std) / mcse(kind = std) (src/ess_rhat.jl:640-642, src/mcse.jl:52-65; round 6): the expectand proxy (x - mean(x))^2, evaluated
// in T like SerSq of mdt_device.hpp, needs the mean over ALL draws of the slice first (rows the split drops included), mcse(std) also
// the pooled mean of the proxy and of its square.  Two streaming sweeps over the slice (the second one finds it in the cache).
WV_PHASE void ph_pooled_sq(unsigned wbase_arg) {
    WV_ENTER(wbase_arg);
    const WV_GA T* xp = (const WV_GA T*)rfl64(ws->xp);
    const unsigned D = (unsigned)kp.D, C = (unsigned)kp.C;
    const long long cs = kp.cstride;
    const unsigned rpc = (D + 63u) >> 6, NRW = C * rpc;
    const double dn = (double)kp.n;
    double acc = 0.0;
    for (unsigned r0 = 0; r0 < NRW; r0 += UB) {
        double vv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned r = r0 + (unsigned)u;
            const RowAt ra_ = row_at(r < NRW ? r : NRW - 1u, rpc, D, cs, lane);
            vv[u] = (r < NRW && ra_.d < D) ? (double)xp[ra_.ad] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) acc += vv[u];
    }
    const T muT = (T)(wave_sum_d(acc) / dn);
    double l1 = 0.0, l2 = 0.0;
    if (kp.mode == 4) {
        for (unsigned r0 = 0; r0 < NRW; r0 += UB) {
            double yy[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const unsigned r = r0 + (unsigned)u;
                const RowAt ra_ = row_at(r < NRW ? r : NRW - 1u, rpc, D, cs, lane);
                const T d = (T)(xp[ra_.ad] - muT);
                yy[u] = (r < NRW && ra_.d < D) ? (double)(T)(d * d) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) { l1 += yy[u]; l2 = fma(yy[u], yy[u], l2); }
        }
        l1 = wave_sum_d(l1) / dn;
        l2 = wave_sum_d(l2) / dn;
    }
    if (lane == 0) { ws->med = (double)muT; ws->qlo = l1; ws->qhi = l2; } // (the order-statistic fields: unused by raw requests)
    lds_fence();
}

// ------------------------------------------------------------------ bulk series: moments, lag products, Geyer's scan
// Per split chain: doubled ranks (time order) -> scores (the workgroup's LDS table) -> stage in LDS, shifted by the score of
// the chain's first draw; exact fixed-point moments (the integers and expressions of series_z_phase, mdt_series.inc: R-hat is
// the same bits); centred; lag products (src/ess_rhat.jl:170-178) as blocked Hankel tiles on v_mfma_f64_16x16x4_f64,
// accumulated over the chains; Geyer's initial monotone sequence (:555-594) between blocks of 32 lags.  While a chain is
// staged its tail-ESS indicator words are taken from the ranks (x <= q <=> doubled rank <= r2q).
// (NB: rows of 64 draws the code is unrolled for -- 4, 8, 10 or 16: the kernel picks the smallest that holds a split chain; 16 rows need the longer per-wave block)
// RAW (round 6): the series are the raw draws themselves (kind = basic, mcse(kind = mean): src/ess_rhat.jl:438-603, src/mcse.jl:45-51) --
// no ranking comes first, a split chain is read from x in time order; its moments are plain double sums taken in the order of the
// workgroup-per-parameter kernels (series_z_phase<RAW>: 1024 virtual threads, element i to thread i mod 1024, wave sums per virtual
// wave, the virtual waves added in index order -- with at most 1024 draws per split chain every virtual thread holds ONE draw, so a
// wave reproduces that order with one wave sum per row of 64 draws): R-hat is the same bits here and there.  A non-finite draw
// hands the parameter to the general kernel.
template <typename T, int NB, bool RAW = false> __device__ __attribute__((noinline)) void ph_series(unsigned wbase_arg) {
    WV_ENTER(wbase_arg);
    WV_LA double* lagA = (WV_LA double*)(wb + OFF_LAGA);
    WV_LA double* mean_j = (WV_LA double*)(wb + OFF_MEAN);
    WV_LA double* var_j = mean_j + WV_MAXM;
    WV_LA double* mu_j = var_j + WV_MAXM;
    WV_LA double* stage = (WV_LA double*)(wb + OFF_STG);
    typedef xs_of<T> ZT_t;
    const WV_LA ZT_t* lz = (const WV_LA ZT_t*)(unsigned long long)rflu(ws->lz);
    const bool tab_lds = rflu(ws->tab_lds) != 0u;
    const WV_GA ZT_t* gzd_s = (const WV_GA ZT_t*)kp.ztab + (2u * (unsigned)kp.n + 1u);
    const WV_GA ZT_t* gz = (const WV_GA ZT_t*)kp.ztab;
    const WV_GA unsigned* rl = (const WV_GA unsigned*)rfl64(ws->ra);
    const unsigned niter = (unsigned)kp.niter;
    const int m = kp.m;
    const long long maxlag = kp.maxlag;
    const double nan = __builtin_nan(""), dn = (double)niter, dm = (double)m;
    const Fx fx = Fx::make(kp.niter);
    const bool acov = (kp.need_eb || (RAW && kp.est == 5)) && kp.ess_ok;
    const bool sq = RAW && kp.est == 5;              // the std proxy (x - mean)^2 in T, mean from ph_pooled_sq
    const T muT = (T)rfld(ws->med);
    const bool anyodd = rflu(ws->anyodd) != 0u;
    const unsigned nblk = (niter + 63u) >> 6;
    const int c_hi = 16 * ((int)maxlag / 16 + 1);
    // indicator words behind the stage, when both fit the view
    const int nwS = (int)nblk, nwU = nwS + (int)(maxlag >> 6) + 2;
    const unsigned stg_doubles = 64u * nblk + 272u;
    const bool bits = !RAW && kp.need_et && kp.ess_ok && rfl(ws->zq_ok) != 0 && (int)(stg_doubles * 8u) + 2 * m * nwU * 8 <= (int)rflu(ws->view_bytes);
    WV_LA unsigned long long* sb2 = (WV_LA unsigned long long*)(stage + stg_doubles);
    const unsigned r2q0 = rflu(ws->r2q[0]), r2q1 = rflu(ws->r2q[1]);
    if (bits) { for (int i = lane; i < 2 * m * nwU; i += 64) sb2[i] = 0ull; }
    for (int k = lane; k < 256; k += 64) lagA[k] = 0.0;
    WB wbm = {0.0, 0.0, 0.0};
    double eb_ = nan;
    double pt = 0.0, gsum = 0.0; // Geyer's state, every lane the same
    long long gk = 0;
    bool started = false, done = false;
    long long k0 = 0;
    int block = 0;
    unsigned rn[RAW ? 1 : NB];
    T xn[RAW ? NB : 1];
    const WV_GA T* xpr = (const WV_GA T*)rfl64(ws->xp);
    const unsigned us_ = (unsigned)kp.s, urem_ = (unsigned)kp.rem;
    const long long cs_ = kp.cstride;
    unsigned bad = 0u;
    auto issue = [&](int j) { // the rank list of split chain j (RAW: its draws): all its rows in flight
        if constexpr (RAW) {
            const unsigned c = (unsigned)j / us_, kk = (unsigned)j - c * us_;
            const WV_GA T* xj = xpr + ((long long)c * cs_ + (long long)(kk * niter + (kk < urem_ ? kk : urem_))); // src/utils.jl:29-36
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const unsigned t = 64u * (unsigned)b + (unsigned)lane;
                xn[b] = xj[t < niter ? t : niter - 1u];
            }
        } else {
            const WV_GA unsigned* rj = rl + (unsigned)j * niter;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const unsigned t = 64u * (unsigned)b + (unsigned)lane;
                rn[b] = rj[t < niter ? t : niter - 1u];
            }
        }
    };
    WV_LA unsigned short* rank_t = (WV_LA unsigned short*)stage; // the list in time order (in the bytes of the stage, before the chain is staged)
    while (true) {
        long long kend = k0 + 16 * (ZT - 1);
        if (kend > maxlag + 1) kend = maxlag + 1;
        int cb[ZT];
#pragma unroll
        for (int i = 0; i < ZT; ++i) { const int c = (int)k0 + 16 * i; cb[i] = c < c_hi ? c : c_hi; }
        for (unsigned t = 64u * nblk + (unsigned)lane; t < stg_doubles; t += 64u) stage[t] = 0.0; // zeros behind every chain
        d4 acc[ZT];
#pragma unroll
        for (int i = 0; i < ZT; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
        issue(0);
        for (int j = 0; j < m; ++j) {
            unsigned r2[RAW ? 1 : NB];
            double z[NB];
            double K = 0.0;
            long long a1 = 0, a2 = 0;
            const double mu0 = (block > 0) ? mu_j[j] : 0.0;
            if constexpr (RAW) {
#pragma unroll
                for (int b = 0; b < NB; ++b) z[b] = (double)xn[b];
                if (sq) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) { const T d = (T)(xn[b] - muT); z[b] = (double)(T)(d * d); }
                }
                if (j + 1 < m) issue(j + 1);
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) { // list -> time order: (draw, rank) pairs scattered in LDS, read back by position
                    const unsigned t = 64u * (unsigned)b + (unsigned)lane;
                    if ((unsigned)b < nblk && t < niter) rank_t[rn[b] & 0xFFFFu] = (unsigned short)(rn[b] >> 16);
                }
                if (j + 1 < m) issue(j + 1);
                lds_fence();
#pragma unroll
                for (int b = 0; b < NB; ++b) { const unsigned t = 64u * (unsigned)b + (unsigned)lane; r2[b] = rank_t[t < niter ? t : niter - 1u]; }
                lds_fence();
#pragma unroll
                for (int b = 0; b < NB; ++b) z[b] = (double)lz[r2[b] >> 1];
                if (!tab_lds) { // (the shape's table is not in LDS: the same entries from the dense global copy, L2)
#pragma unroll
                    for (int b = 0; b < NB; ++b) z[b] = (double)gzd_s[r2[b] >> 1];
                }
                if (anyodd) { // a half-integer rank: the table by doubled rank
#pragma unroll
                    for (int b = 0; b < NB; ++b) if (r2[b] & 1u) z[b] = (double)gz[r2[b]];
                }
            }
            K = readlane_d(z[0], 0);
            double mu = mu0;
            if (RAW && block == 0) {
                // the sums of series_z_phase<RAW> (mdt_series.inc): virtual wave b holds the draws 64 b ... 64 b + 63 of the chain, one per
                // virtual thread (s1 = 0 + u, s2 = fma(u, u, 0)); its wave sums; the virtual waves added in index order
                double S1 = 0.0, S2 = 0.0;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const unsigned t = 64u * (unsigned)b + (unsigned)lane;
                    const bool okt = (unsigned)b < nblk && t < niter;
                    const double u = okt ? z[b] - K : 0.0;
                    bad |= (okt && !(z[b] - z[b] == 0.0)) ? 1u : 0u;
                    double s1 = 0.0, s2 = 0.0;
                    s1 += u;
                    s2 = fma(u, u, s2);
                    const double r1 = wave_sum_d(s1), r2s = wave_sum_d(s2);
                    S1 += r1;
                    S2 += r2s;
                }
                mu = S1 / dn;
                if (lane == 0) { var_j[j] = (S2 - mu * S1) / (double)(niter - 1u); mean_j[j] = K + mu; mu_j[j] = mu; }
            }
            if (!RAW && block == 0) {
                // exact integer sums of the shifted scores; the expressions of series_z_phase: bit-identical means and variances
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const unsigned t = 64u * (unsigned)b + (unsigned)lane;
                    const bool okt = (unsigned)b < nblk && t < niter;
                    const double u = z[b] - K;
                    a1 += okt ? fx.u(u) : 0;
                    a2 += okt ? fx.uu(u * u) : 0;
                }
                const double S1 = (double)wave_sum_i64(a1) * fx.i1, S2 = (double)wave_sum_i64(a2) * fx.i2;
                mu = S1 / dn;
                const double Cc = S2 - mu * S1;
                if (lane == 0) { var_j[j] = Cc / (double)(niter - 1u); mean_j[j] = K + mu; mu_j[j] = mu; }
                if (bits) {
#pragma unroll
                    for (int b = 0; b < (RAW ? 0 : NB); ++b) {
                        if ((unsigned)b < nblk) {
                            const bool okt = 64u * (unsigned)b + (unsigned)lane < niter;
                            const unsigned long long q0 = __ballot(okt && r2[b] <= r2q0), q1 = __ballot(okt && r2[b] <= r2q1);
                            if (lane == 0) { sb2[(2 * j) * nwU + b] = q0; sb2[(2 * j + 1) * nwU + b] = q1; }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }
            if (acov) { // the centred chain, zeros behind it
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const unsigned t = 64u * (unsigned)b + (unsigned)lane;
                    if ((unsigned)b < nblk) stage[t] = (t < niter) ? (z[b] - K) - mu : 0.0;
                }
            }
            WPROF(14);
            if (acov) {
                lds_fence();
                // A = u[64 b + lane], B = the same 64 values shifted by the tile's base: lag c + j - i sits at (i, j) of tile c
                for (unsigned b = 0; b < nblk; ++b) {
                    const WV_LA double* pb = stage + 64u * b + (unsigned)lane;
                    const double a = pb[0];
#pragma unroll
                    for (int i = 0; i < ZT; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[cb[i]], acc[i], 0, 0, 0);
                }
            }
            lds_fence();
            WPROF(15);
        }
        if (RAW && block == 0 && __ballot(bad != 0u) != 0ull) { // a NaN or an infinity among the draws: the general kernel serves the parameter
            if (lane == 0) ws->fallback = 1;
            lds_fence();
            return;
        }
        if (block == 0) wbm = combine_moments(mean_j, var_j, m, kp.cf);
        if (!acov) break;
        { // the diagonals of the tiles: lane e holds strip lag e - 15 (relative to k0)
            double sdiag = 0.0;
            WV_LA double* stg = stage;
#pragma unroll
            for (int i = 0; i < ZT; ++i) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) stg[((lane >> 4) + 4 * rg) * 17 + (lane & 15)] = acc[i][rg];
                lds_fence();
                const int dd = lane - 16 * i;
                if (dd >= 0 && dd <= 30) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int jj = r + dd - 15;
                        if (jj >= 0 && jj < 16) sdiag += stg[r * 17 + jj];
                    }
                }
                lds_fence();
            }
            const long long kq = k0 + lane - 15;
            if (lane >= 15 && kq < kend) lagA[kq] = sdiag;
            lds_fence();
        }
        { // Geyer's initial monotone sequence over the lags available so far (the logic of series_combine)
            const double W = wbm.W, inv_varp = 1.0 / wbm.varp;
            auto rho = [&](long long k) -> double {
                const double ma = lagA[k] / dm / dn;
                return 1.0 - inv_varp * (W - ma);
            };
            const long long avail = kend - 1;
            if (!started) { pt = 1.0 + rho(1); gsum = pt; gk = 2; started = true; }
            int stop = 0;
            while (gk < maxlag - 1) {
                if (gk + 1 > avail) break;
                const double delta = rho(gk) + rho(gk + 1);
                if (!(delta > 0.0)) { stop = 1; break; }
                pt = jl_min(delta, pt);
                gsum += pt;
                gk += 2;
            }
            if (!(gk < maxlag - 1)) stop = 1;
            if (stop && (gk <= avail || !(maxlag > 1))) {
                const double rho_even = (maxlag > 1) ? rho(gk) : 0.0;
                const double tau = jl_max(0.0, 2.0 * gsum + jl_max(0.0, rho_even) - 1.0);
                double e = jl_min(1.0 / tau, kp.rel_ess_max);
                if (!kp.relative) e *= kp.ntotal;
                eb_ = e;
                done = true;
            }
        }
        if (rfl(done ? 1 : 0)) break;
        k0 = kend;
        ++block;
    }
    if (acov && kp.method == 1) { // FFTAutocovMethod normalises by the chain's own lag 0: 0/0 = NaN for a constant split chain (src/ess_rhat.jl:192)
        bool anyc = false;
        for (int j = 0; j < m; ++j) anyc = anyc || (var_j[j] == 0.0);
        if (anyc) eb_ = nan;
    }
    if (lane == 0) { ws->out[0] = acov ? eb_ : nan; ws->out[1] = wbm.rhat; ws->bits_done = bits ? 1 : 0; }
    lds_fence();
    WPROF(7);
}

// ------------------------------------------------------------------ tail ESS: indicators of x <= q (src/ess_rhat.jl:301-311, 647-659)
// The indicator words of every split chain (from the ranks while the chain was staged, else built here from the ranks or from
// x itself), per chain and lag sum I_t I_{t+k} = popcount(bits & bits >> k), centred exactly in int64, lanes = 16 lags x 4
// word segments; lanes 0 / 1 run Geyer's scan of the two indicators between blocks of 16 lags.
template <typename T, int NB> __device__ __attribute__((noinline)) void ph_indicators(unsigned wbase_arg) {
    WV_ENTER(wbase_arg);
    WV_LA double* icov = (WV_LA double*)(wb + OFF_ICOV);
    WV_LA double* imean = (WV_LA double*)(wb + OFF_IMEAN);
    const WV_GA T* xp = (const WV_GA T*)rfl64(ws->xp);
    const WV_GA unsigned* rl = (const WV_GA unsigned*)rfl64(ws->ra);
    const unsigned niter = (unsigned)kp.niter, s = (unsigned)kp.s, rem = (unsigned)kp.rem;
    const int m = kp.m;
    const long long cs = kp.cstride, maxlag = kp.maxlag;
    const double nan = __builtin_nan(""), dn = (double)niter, dm = (double)m;
    auto row0 = [&](unsigned k) -> unsigned { return k * niter + (k < rem ? k : rem); };
    const int nwS = (int)((niter + 63u) >> 6);
    const int nwU = nwS + (int)(maxlag >> 6) + 2;
    const bool have = rfl(ws->bits_done) != 0;
    const unsigned stg_doubles = 64u * (unsigned)nwS + 272u;
    WV_LA unsigned long long* sb2 = (WV_LA unsigned long long*)(wb + OFF_SB2);
    if (have) { // the words the series phase took sit behind its stage, where this phase's tables go: move them (through registers)
        const WV_LA unsigned long long* src = (const WV_LA unsigned long long*)((WV_LA double*)(wb + OFF_STG) + stg_doubles);
        constexpr int NWL = (SB2_WORDS + 63) / 64;
        unsigned long long tmp[NWL];
#pragma unroll
        for (int q = 0; q < NWL; ++q) { const int i = lane + 64 * q; tmp[q] = (i < 2 * m * nwU) ? src[i] : 0ull; }
        lds_fence();
#pragma unroll
        for (int q = 0; q < NWL; ++q) { const int i = lane + 64 * q; if (i < 2 * m * nwU) sb2[i] = tmp[q]; }
        lds_fence();
    } else {
        const bool byrank = rfl(ws->zq_ok) != 0;
        const unsigned r2q0 = rflu(ws->r2q[0]), r2q1 = rflu(ws->r2q[1]);
        const T qloT = (T)rfld(ws->qlo), qhiT = (T)rfld(ws->qhi);
        const bool fo = rflu(ws->foldon) != 0u;
        const T fm = (T)rfld(ws->fmed);
        for (int i = lane; i < 2 * m * nwU; i += 64) sb2[i] = 0ull;
        lds_fence();
        for (int j = 0; j < m; ++j) {
            const unsigned c = (unsigned)j / s, k = (unsigned)j - c * s;
            unsigned long long q0[NB], q1[NB];
            if (byrank) {
                const WV_GA unsigned* rj = rl + (unsigned)j * niter;
                WV_LA unsigned short* rank_t = (WV_LA unsigned short*)icov; // (the list in time order, in the bytes of the lag table that is zeroed below)
                unsigned pr[NB], r2[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) { const unsigned t = 64u * (unsigned)b + (unsigned)lane; pr[b] = rj[t < niter ? t : niter - 1u]; }
#pragma unroll
                for (int b = 0; b < NB; ++b) { const unsigned t = 64u * (unsigned)b + (unsigned)lane; if (t < niter) rank_t[pr[b] & 0xFFFFu] = (unsigned short)(pr[b] >> 16); }
                lds_fence();
#pragma unroll
                for (int b = 0; b < NB; ++b) { const unsigned t = 64u * (unsigned)b + (unsigned)lane; r2[b] = rank_t[t < niter ? t : niter - 1u]; }
                lds_fence();
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const bool okt = 64u * (unsigned)b + (unsigned)lane < niter;
                    q0[b] = __ballot(okt && r2[b] <= r2q0);
                    q1[b] = __ballot(okt && r2[b] <= r2q1);
                }
            } else {
                const WV_GA T* xr = xp + (long long)c * cs + row0(k);
                T xv[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) { const unsigned t = 64u * (unsigned)b + (unsigned)lane; xv[b] = xr[t < niter ? t : niter - 1u]; }
                if (fo) { // (wave-uniform: ess(kind = mad))
#pragma unroll
                    for (int b = 0; b < NB; ++b) xv[b] = fold_T<T>(xv[b], fm, true);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const bool okt = 64u * (unsigned)b + (unsigned)lane < niter;
                    q0[b] = __ballot(okt && xv[b] <= qloT);
                    q1[b] = __ballot(okt && xv[b] <= qhiT);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int b = 0; b < NB; ++b) if (b < nwS) { sb2[(2 * j) * nwU + b] = q0[b]; sb2[(2 * j + 1) * nwU + b] = q1[b]; }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    for (int i = lane; i < 2 * 256; i += 64) icov[i] = 0.0;
    lds_fence();
    double ipt = 0.0, isum = 0.0, iess = nan;
    long long ik = 0;
    bool istarted = false, idone = lane >= 2;
    const int kl = lane & 15, seg = lane >> 4;
    const int wq = (nwS + 3) >> 2, i0 = seg * wq, i1 = (i0 + wq < nwS) ? i0 + wq : nwS;
    long long k0 = 0;
    while (true) {
        long long kend = k0 + 16;
        if (kend > maxlag + 1) kend = maxlag + 1;
        const long long k = k0 + kl;
        double cov0 = 0.0, cov1 = 0.0;
        for (int u = 0; u < 2 * m; ++u) {
            const WV_LA unsigned long long* sbu = sb2 + u * nwU;
            int sxp = 0, akp = 0;
            if (k < kend) {
                // (round 6: the window of a lag from two 32-bit funnel shifts over the words read as halves -- v_alignbit_b32 at full rate
                // instead of a 64-bit shift pair at a quarter of it --, four words per step with their LDS reads in flight together:
                // lag_popcounts of mdt_indicator.inc; a unit's words are followed by zero words, so no read needs a bound)
                typedef unsigned __attribute__((may_alias)) u32a;
                const WV_LA u32a* W = (const WV_LA u32a*)sbu;
                const unsigned r5 = (unsigned)k & 31u;
                int d = 2 * i0 + (int)(k >> 5);
                unsigned w0 = W[d], w1 = W[d + 1];
                int i = i0;
                for (; i + 4 <= i1; i += 4) {
                    unsigned long long v[4];
                    unsigned x2[4], x3[4];
#pragma unroll
                    for (int u2 = 0; u2 < 4; ++u2) { v[u2] = sbu[i + u2]; x2[u2] = W[d + 2 + 2 * u2]; x3[u2] = W[d + 3 + 2 * u2]; }
#pragma unroll
                    for (int u2 = 0; u2 < 4; ++u2) {
                        const unsigned lo = __builtin_amdgcn_alignbit(w1, w0, r5), hi = __builtin_amdgcn_alignbit(x2[u2], w1, r5);
                        const unsigned vl = (unsigned)v[u2], vh = (unsigned)(v[u2] >> 32);
                        sxp += __popc(vl) + __popc(vh);
                        akp += __popc(vl & lo) + __popc(vh & hi);
                        w0 = x2[u2]; w1 = x3[u2];
                    }
                    d += 8;
                }
                for (; i < i1; ++i) {
                    const unsigned long long v = sbu[i];
                    const unsigned x2 = W[d + 2], x3 = W[d + 3];
                    const unsigned lo = __builtin_amdgcn_alignbit(w1, w0, r5), hi = __builtin_amdgcn_alignbit(x2, w1, r5);
                    const unsigned vl = (unsigned)v, vh = (unsigned)(v >> 32);
                    sxp += __popc(vl) + __popc(vh);
                    akp += __popc(vl & lo) + __popc(vh & hi);
                    w0 = x2; w1 = x3;
                    d += 2;
                }
            }
            __builtin_amdgcn_wave_barrier();
            sxp += __shfl_xor(sxp, 16); sxp += __shfl_xor(sxp, 32);
            akp += __shfl_xor(akp, 16); akp += __shfl_xor(akp, 32);
            if (k < kend && seg == 0) {
                const long long Sx = sxp, Ak = akp, ni = (long long)niter;
                const int q = (int)(k >> 6);
                long long head = 0, tail = 0; // ones among the first k / the last k positions
                for (int i = 0; i <= q; ++i) {
                    const long long nb2 = k - ((long long)i << 6);
                    const unsigned long long mk = nb2 >= 64 ? ~0ull : (nb2 <= 0 ? 0ull : ((1ull << nb2) - 1ull));
                    head += __popcll(sbu[i] & mk);
                }
                {
                    const long long p0 = ni - k;
                    for (int i = (int)((p0 > 0 ? p0 : 0) >> 6); i < nwS; ++i) {
                        const long long bpos = (long long)i << 6;
                        const long long sk = p0 - bpos;
                        const unsigned long long mk = sk <= 0 ? ~0ull : (sk >= 64 ? 0ull : ~((1ull << sk) - 1ull));
                        tail += __popcll(sbu[i] & mk);
                    }
                }
                const long long Tk = Sx - head, Hk = Sx - tail;
                const long long N = ni * ni * Ak - ni * Sx * (Hk + Tk) + (ni - k) * Sx * Sx;
                double ck = (double)N / (dn * dn);
                // FFTAutocovMethod normalises by the lag-0 value of the same chain: 0/0 -> NaN for a constant split chain (src/ess_rhat.jl:192)
                if (kp.method == 1 && k > 0 && (Sx == 0 || Sx == ni)) ck = nan;
                if (k == 0) imean[(u & 1) * WV_MAXM + (u >> 1)] = (double)Sx / dn;
                cov0 += (u & 1) ? 0.0 : ck;
                cov1 += (u & 1) ? ck : 0.0;
            }
            __builtin_amdgcn_wave_barrier(); // (every lane joins before the next unit's shuffles)
        }
        if (k < kend && seg == 0) { icov[k] = cov0; icov[256 + k] = cov1; }
        lds_fence();
        if (!idone) {
            const int ind = lane;
            const WV_LA double* cv = icov + ind * 256;
            const double W = cv[0] / dm / (double)(niter - 1u);
            double mb = 0.0;
            for (int j = 0; j < m; ++j) mb += imean[ind * WV_MAXM + j];
            mb /= dm;
            double vb = 0.0;
            for (int j = 0; j < m; ++j) { const double dd = imean[ind * WV_MAXM + j] - mb; vb = fma(dd, dd, vb); }
            vb /= (double)(m > 1 ? m - 1 : m);
            const double varp = kp.cf * W + vb;
            const double inv_varp = 1.0 / varp;
            auto rho = [&](long long kk) -> double {
                const double ma = cv[kk] / dm / dn;
                return 1.0 - inv_varp * (W - ma);
            };
            const long long avail = kend - 1;
            if (!istarted) { ipt = 1.0 + rho(1); isum = ipt; ik = 2; istarted = true; }
            int stop = 0;
            while (ik < maxlag - 1) {
                if (ik + 1 > avail) break;
                const double delta = rho(ik) + rho(ik + 1);
                if (!(delta > 0.0)) { stop = 1; break; }
                ipt = jl_min(delta, ipt);
                isum += ipt;
                ik += 2;
            }
            if (!(ik < maxlag - 1)) stop = 1;
            if (stop && (ik <= avail || !(maxlag > 1))) {
                const double rho_even = (maxlag > 1) ? rho(ik) : 0.0;
                const double tau = jl_max(0.0, 2.0 * isum + jl_max(0.0, rho_even) - 1.0);
                double e = jl_min(1.0 / tau, kp.rel_ess_max);
                if (!kp.relative) e *= kp.ntotal;
                iess = e;
                idone = true;
            }
        }
        if (__ballot(!idone) == 0ull) break;
        k0 = kend;
    }
    const double et = jl_min(readlane_d(iess, 0), readlane_d(iess, 1));
    if (lane == 0) ws->out[2] = et;
    lds_fence();
    WPROF(8);
}

// ------------------------------------------------------------------ the kernel
// (progress words in host-pinned memory: diagnostics builds only -- tools/build_wave_variant.sh NAME -DWV_DEBUG, tools/dev_wave_dbg.py)
#if defined(WV_DEBUG) || defined(WV_PROF)
#define WDBG(code) do { if (wa.dbg && lane == 0) __hip_atomic_store(wa.dbg + gw, (unsigned)(code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#else
#define WDBG(code) do { } while (0)
#endif
template <typename T>
__global__ __launch_bounds__(NTHREADS) void wave_kernel(const KP* __restrict__ kpp, WvArgs wa) {
    const WV_CA KP& kp = *(const WV_CA KP*)(unsigned long long)kpp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef xs_of<T> ZT_t;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned n = (unsigned)kp.n;
    const unsigned gw = blockIdx.x * (blockDim.x >> 6) + (unsigned)w;
    const double nan = __builtin_nan("");
    WDBG(1);
    // ---- the normal-score table of whole ranks, once per workgroup: lz[r] = Phi^-1((r - 3/8) / (n + 1/4)), r = 1 .. n
    {
        ZT_t* lz = (ZT_t*)smem;
        const WV_GA ZT_t* gzd = (const WV_GA ZT_t*)kp.ztab + (2u * n + 1u); // (behind the table by doubled rank: the dense copy for whole ranks)
        if (wa.tab_bytes > 0) for (unsigned i = (unsigned)tid; i <= n; i += blockDim.x) lz[i] = gzd[i];
    }
    __syncthreads(); // the only workgroup barrier of the launch
    WDBG(2);
    const unsigned smem0 = (unsigned)(unsigned long long)(WV_LA unsigned char*)smem;
    const unsigned wbase = rflu(smem0 + (unsigned)wa.tab_bytes + (unsigned)w * (unsigned)wa.perwave);
    WV_LA unsigned char* wb = (WV_LA unsigned char*)(unsigned long long)wbase;
    WV_LA WS* ws = (WV_LA WS*)(wb + OFF_WS);
    if (lane == 0) {
        ws->kp = (unsigned long long)kpp;
        ws->rec = (unsigned long long)(wa.rec + (long long)gw * wa.npad);
        ws->ra = (unsigned long long)(wa.ra + (long long)gw * wa.npad);
        ws->dbg = (unsigned long long)wa.dbg;
        ws->lz = smem0;
        ws->tab_lds = wa.tab_bytes > 0 ? 1u : 0u;
        ws->view_bytes = (unsigned)(wa.perwave - OFF_VIEW);
        ws->gw = gw;
        ws->stop = wa.stop;
    }
#ifdef WV_PROF
    if (lane < 24) ((WV_LA unsigned long long*)(wb + OFF_PROF))[lane] = 0ull;
    if (lane == 0) ((WV_LA unsigned long long*)(wb + OFF_PROF))[24] = (unsigned long long)__builtin_readcyclecounter();
#endif
    lds_fence();

    while (true) {
        int tk = 0;
        if (lane == 0) tk = atomicAdd(kp.status + 1, 1);
        tk = rfl(tk);
        WDBG(0x100 + tk);
        if (tk >= kp.P) break;
        const long long p = tk;
        if (lane == 0) {
            ws->p = p;
            ws->xp = (unsigned long long)((const T*)kp.x + p * kp.pstride);
            ws->fallback = 0; ws->zq_ok = 0; ws->bits_done = 0; ws->anyodd = 0u; ws->nrounds = 0u; ws->foldon = 0u; ws->fmed = 0.0;
            ws->osonly = ((((kp.need_et || kp.est) && kp.ess_ok) || kp.mode == 5) && !kp.need_eb && !kp.need_rb && !kp.need_rt && wa.osonly) ? 1u : 0u;
            ws->out[0] = nan; ws->out[1] = nan; ws->out[2] = nan; ws->out[3] = nan;
        }
        lds_fence();
        WPROF(0);
        const int stop = wa.stop; // diagnostics: the parameter ends after phase `stop`, its outputs stay NaN
        WDBG(10);
        if (kp.mode == 5 && kp.tpos_arr[2 * p] == 0xFFFFFFFFu) { // mcse quantile of a NaN ESS (src/mcse.jl:95-97)
            if (lane == 0) store_out<T>(kp.out_eb, p, nan);
            lds_fence();
            continue;
        }
        if (kp.basic) {
            // raw draws (round 6): ess / rhat / ess_rhat(kind = basic) and mcse(kind = mean) -- no ranking, the series phase alone
            const unsigned nblk = ((unsigned)kp.niter + 63u) >> 6;
            if (kp.est == 5) ph_pooled_sq<T>(wbase);
            if (nblk <= 4u) ph_series<T, 4, true>(wbase);
            else if (nblk <= 8u) ph_series<T, 8, true>(wbase);
            else if (nblk <= (unsigned)((WV_NITER_SHORT + 63) / 64)) ph_series<T, (WV_NITER_SHORT + 63) / 64, true>(wbase);
            else ph_series<T, (WV_NITER + 63) / 64, true>(wbase);
            if (rfl(ws->fallback) != 0) {
                if (lane == 0) { const int i = atomicAdd(wa.fb_count, 1); wa.fb_list[i] = (unsigned)p; }
            } else if (lane == 0) {
                const double ess = ((kp.need_eb || kp.est == 5) && kp.ess_ok) ? ws->out[0] : nan;
                if (kp.est == 5) {
                    // ess(kind = std): the basic ESS of the proxy; mcse(kind = std) = sqrt((E[mu4] / E[var] - E[var]) / S) / 2 (src/mcse.jl:52-65)
                    const double mv = ws->qlo, m4 = ws->qhi;
                    store_out<T>(kp.out_eb, p, kp.mode == 4 ? sqrt((m4 / mv - mv) / ess) / 2.0 : ess);
                } else if (kp.mode == 2) {
                    // mcse(kind = mean) = pooled corrected std / sqrt(ESS) (src/mcse.jl:45-51); every draw belongs to a split chain here
                    // (no row dropped by the split), so the pooled moments follow from the split-chain ones -- Chan's combination, the
                    // expressions of raw_request in mdt_device.hpp
                    const WV_LA double* mean_j = (const WV_LA double*)(wb + OFF_MEAN);
                    const WV_LA double* var_j = mean_j + WV_MAXM;
                    double mu = 0.0;
                    for (int j = 0; j < kp.m; ++j) mu += mean_j[j];
                    mu /= (double)kp.m;
                    double ss = 0.0;
                    for (int j = 0; j < kp.m; ++j) { const double d = mean_j[j] - mu; ss += (double)(kp.niter - 1) * var_j[j] + (double)kp.niter * d * d; }
                    store_out<T>(kp.out_eb, p, sqrt(ss / (double)(kp.n - 1)) / sqrt(ess));
                } else {
                    store_out<T>(kp.out_eb, p, ess);
                    store_out<T>(kp.out_rb, p, ws->out[1]);
                }
            }
            WPROF(9);
            lds_fence(); // (every lane joins before the next ticket: see the end of the loop)
            continue;
        }
        ph_classify<T>(wbase);
        bool fb = rfl(ws->fallback) != 0;
        WDBG(20);
        if (!fb && stop != 1) {
            ph_partition<T>(wbase);
            WDBG(30);
            if (stop != 2) {
                ph_rounds<T>(wbase);
                WDBG(40);
                if (kp.est == 6) {
                    // ess(kind = mad): the median of |x - median(x)| by a second ranking of the same kind -- the three passes read the
                    // folded value where they read x (order statistics only: two or three rounds of the scratch are written)
                    if (lane == 0) { ws->fmed = ws->med; ws->foldon = 1u; }
                    lds_fence();
                    ph_classify<T, true>(wbase);
                    fb = rfl(ws->fallback) != 0;
                    if (!fb) { ph_partition<T, true>(wbase); ph_rounds<T>(wbase); }
                }
                if (!fb && stop != 3) {
                    if (kp.need_rt) { ph_fold<T>(wbase); fb = rfl(ws->fallback) != 0; }
                    WDBG(50);
                    if (!fb && stop != 4) {
                        const unsigned nblk = ((unsigned)kp.niter + 63u) >> 6;
                        if (kp.need_eb || kp.need_rb) {
                            if (nblk <= 4u) ph_series<T, 4>(wbase);
                            else if (nblk <= 8u) ph_series<T, 8>(wbase);
                            else if (nblk <= (unsigned)((WV_NITER_SHORT + 63) / 64)) ph_series<T, (WV_NITER_SHORT + 63) / 64>(wbase);
                            else ph_series<T, (WV_NITER + 63) / 64>(wbase);
                        }
                        WDBG(60);
                        if ((kp.need_et || kp.est) && kp.ess_ok && stop != 5) {
                            if (nblk <= 4u) ph_indicators<T, 4>(wbase);
                            else if (nblk <= 8u) ph_indicators<T, 8>(wbase);
                            else if (nblk <= (unsigned)((WV_NITER_SHORT + 63) / 64)) ph_indicators<T, (WV_NITER_SHORT + 63) / 64>(wbase);
                            else ph_indicators<T, (WV_NITER + 63) / 64>(wbase);
                        }
                    }
                }
            }
        }
        WDBG(70);
        if (fb) { // the general kernel, queued behind this launch, serves the parameter
            if (lane == 0) { const int i = atomicAdd(wa.fb_count, 1); wa.fb_list[i] = (unsigned)p; }
        } else if (kp.est || kp.mode == 5) { // ess(kind = median | mad | quantile(p)): the ESS of the one indicator (the two "tails" are the same series);
                                             // mode 5: half the distance of the two order statistics
            if (lane == 0) store_out<T>(kp.out_eb, p, ws->out[2]);
        } else if (lane == 0) {
            const double eb_ = ws->out[0], rb_ = ws->out[1], et_ = ws->out[2], rt_ = ws->out[3];
            store_out<T>(kp.out_eb, p, eb_);
            store_out<T>(kp.out_rb, p, rb_);
            store_out<T>(kp.out_et, p, et_);
            store_out<T>(kp.out_rt, p, rt_);
            store_out<T>(kp.out_rrank, p, jl_max(rt_, rb_)); // map(max, Rtail, Rbulk), src/ess_rhat.jl:419,622
        }
        WPROF(9);
        // Every lane joins here before the next ticket: without a convergent operation behind the last lane-0 block the compiler
        // sends lanes 1..63 straight back to the loop head (they have nothing to store) and the next iteration runs without lane 0
        // -- no ticket is drawn, parameter 0 is worked on again, for ever.  (Found with the progress words, round 5.)
        lds_fence();
    }
#ifdef WV_PROF
    if (wa.dbg && lane < 24) ((unsigned long long*)(wa.dbg + 8192))[gw * 24 + lane] = ((WV_LA unsigned long long*)(wb + OFF_PROF))[lane];
#endif
    WDBG(0xFFFF);
}

// ------------------------------------------------------------------ host side
size_t kp_bytes() { return sizeof(KP); }

Geo geometry(long long n, long long niter, int f32, const Hooks& hk, int raw) {
    Geo g;
    g.tab_bytes = raw ? 0 : (int)((((n + 1) * (f32 ? 4 : 8)) + 15) / 16 * 16); // (raw draws are never ranked: no score table)
    // above 4096 values the table (64 KB at 8000) would leave 8 waves per CU: the series phase then reads its entries from the dense
    // global copy instead (L2) and the workgroup keeps its 12 waves -- 1000 x 8 x 50 000: 13.6 -> 13.2 ms; below, the LDS table is the
    // faster one by 0.5 - 2 % (round 5, MDT_WAVE_TAB = 1 / 2 forces either)
    if (hk.tab_mode == 2 || (hk.tab_mode == 0 && n > 4096)) g.tab_bytes = 0;
    g.perwave = niter <= WV_NITER_SHORT ? PERWAVE_SHORT : PERWAVE;
    int wv = (LDS_TOTAL - g.tab_bytes) / g.perwave;
    if (wv > NTHREADS / 64) wv = NTHREADS / 64;
    if (hk.waves > 0 && hk.waves < wv) wv = hk.waves; // diagnostics (MDT_WAVE_WAVES): fewer waves per workgroup
    if (wv < 1) wv = 1;
    g.waves = wv;
    g.lds_bytes = g.tab_bytes + wv * g.perwave;
    g.npad = ((n + 63) / 64) * 64;
    return g;
}

size_t scratch_bytes(const Geo& g, long long waves_total) {
    return (size_t)waves_total * (size_t)g.npad * (12 + 4);
}

int prepare_device() { // > 64 KB of dynamic LDS needs the opt-in, per kernel and device (called by mdt_ctx_create on its device)
    if (hipFuncSetAttribute((const void*)wave_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)wave_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return 1;
    return 0;
}

void launch(hipStream_t st, const KP* kp_slot, int f32, const Geo& g, int workgroups, void* scratch, unsigned* fb_list, int* fb_count,
            const Hooks& hk) {
    const long long waves_total = (long long)workgroups * g.waves;
    WvArgs wa;
    wa.rec = (Rec*)scratch;
    wa.ra = (unsigned*)((char*)scratch + (size_t)waves_total * g.npad * 12);
    wa.fb_list = fb_list;
    wa.fb_count = fb_count;
    wa.npad = g.npad;
    wa.tab_bytes = g.tab_bytes;
    wa.perwave = g.perwave;
    wa.stop = hk.stop;
    wa.osonly = hk.osonly;
#if defined(WV_DEBUG) || defined(WV_PROF)
    wa.dbg = hk.dbg; // (diagnostics builds only: a release library never stores through a pointer that came from the environment)
#else
    wa.dbg = nullptr;
#endif
    if (f32) hipLaunchKernelGGL(wave_kernel<float>, dim3((unsigned)workgroups), dim3((unsigned)(64 * g.waves)), (size_t)g.lds_bytes, st, kp_slot, wa);
    else hipLaunchKernelGGL(wave_kernel<double>, dim3((unsigned)workgroups), dim3((unsigned)(64 * g.waves)), (size_t)g.lds_bytes, st, kp_slot, wa);
}

} // namespace mdtw
