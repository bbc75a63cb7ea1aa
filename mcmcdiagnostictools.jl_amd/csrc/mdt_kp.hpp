// mdt_kp.hpp -- the kernel parameter block shared by every translation unit of libmdt_hip.so
// (mdt_hip.hip: the workgroup-per-parameter kernels of mdt_device.hpp; mdt_wave.hip: the wave-per-parameter kernel).
#pragma once
#ifndef MDT_DEVICE_COMMON
#define MDT_DEVICE_COMMON
namespace mdtc {
struct KP {
    const void* x;
    long long cstride, pstride; // elements between chains / parameters of x (dense: D, D*C)
    int strided;                // x is not dense: every workgroup gathers its parameter into xcbuf first
    void* xcbuf;                // per-workgroup scratch for that, stride n_pad elements of T
    unsigned long long* ibits;  // per-workgroup scratch, stride ibits_stride words: the tail-ESS indicator bits of every split chain, taken from the z-series
    long long ibits_stride;
    long long D, C, P, n;       // draws, chains, params, n = D*C
    long long niter;            // D / s
    int s, rem, m;              // split_chains, D % s, m = C*s
    int mode;                   // 0 ess_rhat / bulk_tail, 1 rhat_nested, 2 mcse_mean, 3 rank_normalize stage, 4 mcse_std, 5 mcse quantile pass 2, 6 mcse_mean + rhat_nested
    int basic;                  // kind == basic (no ranking); outputs via out_eb / out_rb
    int need_eb, need_rb, need_et, need_rt; // bulk ESS / bulk Rhat / tail ESS / tail Rhat
    int method;                 // 0 direct, 1 fft, 2 bda
    int ess_ok;                 // niter > 4
    int relative;
    int fold;                   // rank_normalize stage: rank |x - median| instead of x
    const unsigned int* tpos_arr; // mode 5: [P][2] zero-based sorted positions (l-1, u-1) per parameter; 0xFFFFFFFF: NaN
    int fft_lds;                // FFTAutocovMethod: 1 = run the in-LDS transform when it fits (MDT_FFT_LDS=1), 0 = lagged products
    int est;                    // ess(kind = estimator): 0 none, 4 median, 5 std, 6 mad, 7 quantile(p) (src/ess_rhat.jl:626-659); mode 4: mcse(std)
    int winout;                 // 1: chains that do not fit the LDS take the windowed score layout (series_win_phase); 0: time-order scatter (MDT_WINOUT=0: tests, A/B timing)
    int fold_merge;             // 1: rank |x - median| by merging the sorted slice when the ranking of x left one (fold_merge_phase); 0: always the general ranking (MDT_FOLD_MERGE=0: tests, A/B timing)
    int pa_on;                  // 1: tile-major partition where it applies (rank_phase_q, `pa`); 0: histogram + partition passes (MDT_PA=0: tests, A/B timing)
    int rank_variant;           // 0: rounds re-stream the slice (rank_phase_b), 1: partitioned rounds (rank_phase_p), 2: two-level (rank_phase_q)
    int tail_srt;               // 1: a tail-only request with R-hat ranks every round for the sorted copy, its folded ranking is the merge (round 6); 0: order statistics only + the general folded ranking (MDT_TAIL_SRT=0: tests, A/B timing)
    int f4_stag;                // Float32 rounds keep the split-chain slot of every chunk slot in a byte array behind the chunk (round 6): bit 0 the slots of the sorted copy leave from there (0: the three-barrier route), bit 1 a moments-only round adds its scores to the chains' sums in sorted order (0: back to partition order first); MDT_F4_STAG=0..3: tests, A/B timing
    int sq_fast;                // 1: the std proxy takes the one-chain-per-buffer sweep where a split chain fits the LDS (round 6); 0: the tiled series_phase (MDT_SQ_FAST=0: tests, A/B timing)
    int onechunk;               // 1: short slices take the one-chunk path (oc_rank_phase / oc_series_phase, mdt_short.inc) where it applies; 0: never (MDT_ONECHUNK=0: tests, A/B timing)
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
    void* out_mcse;             // mode 6 (mcse(mean) + rhat_nested fused): [P] mcse
    double* rank_out;           // stage: [P*n] tied ranks
    void* z_out;                // stage: [P*n] normal scores (T)
    const int* chain_inds;      // nested: K x S (device)
    int K, S;
    int* status;                // device status words: [0] flags of this launch (bit0 rounds overflow, bit1 NaN hit quantile, bit2 FDist); [1] next-parameter ticket; [2] sticky flags of earlier launches
    unsigned long long* prof;   // optional [grid*16] phase cycle counters (diagnostics), or null
    const unsigned int* order;  // ticket t works on parameter order[t] (costliest first, by the durations the previous launch of the same shape measured), or null: t itself
    unsigned int* cost;         // [P] duration of every parameter of this launch (shader clocks >> 6), for the next launch's order; or null
};
} // namespace mdtc
#endif
