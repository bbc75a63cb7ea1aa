/*
 * mdt_hip.h -- C-ABI of libmdt_hip.so: the MI355X (gfx950) ESS / R-hat engine that stands behind
 * MCMCDiagnosticTools.jl's ess / rhat / ess_rhat / rhat_nested / mcse(kind=mean).
 *
 * Boundary.  The reference is pure Julia and has no FFI for this path; the seam that is replaced
 * is the set of exported functions themselves (SURVEY.md §8b "coarse seam").  A Julia shim
 * (julia/MCMCDiagnosticToolsROC.jl) validates arguments exactly like the
 * reference, flattens the parameter dimensions (src/utils.jl:197-215) and makes ONE ccall per
 * request into the entry points below.  Each entry point cites the reference code it replaces
 * (paths relative to the reference repository).
 *
 * Conventions
 *   - arrays are column-major, shaped (draws, chains, params): x[d + chain_stride*c + param_stride*p] with the
 *     draws of one chain contiguous; the strides default (0) to the dense Julia Array{T,3},
 *     chain_stride = draws, param_stride = draws*chains; (iterations, parameters, chains) storage as MCMCChains
 *     keeps it is chain_stride = draws*params, param_stride = draws, read in place (src/utils.jl:203-211 only
 *     ever takes views); outputs have `params` elements of dtype T;
 *   - every function returns an int status: MDT_OK or one of the MDT_E* codes; the message for
 *     the last failing call on a context is available from mdt_last_error();
 *   - `location` says whether x / outputs are host pointers (the library stages them through its
 *     own device buffers) or device pointers on the context's GPU (zero-copy; the hot path);
 *   - a context is single-owner: one call at a time; distinct contexts may be used from distinct
 *     host threads.  With `sync != 0` (what Julia wants: it reads the results immediately) the
 *     call returns after the stream has drained; with `sync == 0` and device pointers the work
 *     is only enqueued on `stream`; conditions that a synchronous call reports as an error (a NaN under a
 *     quantile, a slice too large for the ranking tables) are then kept as sticky flags on the context:
 *     mdt_ctx_status() returns them;
 *   - no entry point changes the calling thread's current HIP device;
 *   - no exceptions cross the boundary; NaN inputs propagate as in the reference
 *     (constant parameter -> NaN ESS and NaN R-hat, test/ess_rhat.jl:242-257).
 */
#ifndef MDT_HIP_H
#define MDT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDT_VERSION 310 /* 0.3.1: mdt_ctx_reserve (the allocations of a shape before its first request); 0.3.0: mdt_args gained x_shards
                           (pre-sharded input of a multi-GPU context); mdt_shard_bounds; mdt_mcse_nested */

/* status codes (the Julia shim maps them to the reference's exception types) */
#define MDT_OK 0
#define MDT_EARG 1     /* ArgumentError   (src/ess_rhat.jl:286,294,347,453; src/rhat_nested.jl:49-79) */
#define MDT_EDOMAIN 2  /* DomainError     (maxlag <= 0, src/ess_rhat.jl:481) */
#define MDT_EDIM 3     /* DimensionMismatch (src/utils.jl:18-28, src/rhat_nested.jl:69-73) */
#define MDT_ERUNTIME 4 /* HIP runtime error */
#define MDT_ENOMEM 5   /* device or host allocation failed */
#define MDT_EUNSUPPORTED 6 /* valid request the engine cannot serve (message says why) */

/* dtype */
#define MDT_F64 0
#define MDT_F32 1
/* location */
#define MDT_HOST 0
#define MDT_DEVICE 1
/* kind: the `kind` keyword of ess / rhat / ess_rhat / rhat_nested (src/ess_rhat.jl:276-290,
 * 335-349, 438-455; src/rhat_nested.jl:55-65) */
#define MDT_KIND_BASIC 0
#define MDT_KIND_BULK 1
#define MDT_KIND_TAIL 2
#define MDT_KIND_RANK 3
/* estimators of ess(x; kind = f) with an expectand proxy (src/ess_rhat.jl:291-297, 626-659); mdt_ess_rhat with
 * outputs == MDT_OUT_ESS only.  MDT_KIND_QUANTILE is Base.Fix2(Statistics.quantile, p) with p = args->tail_prob. */
#define MDT_KIND_MEDIAN 4   /* Statistics.median: 1[x <= median(x)]            (:630-639) */
#define MDT_KIND_STD 5      /* Statistics.std: (x - mean(x))^2                 (:640-642) */
#define MDT_KIND_MAD 6      /* StatsBase.mad: median proxy of |x - median(x)|  (:643-646) */
#define MDT_KIND_QUANTILE 7 /* Fix2(quantile, p): 1[x <= quantile(x, p)]       (:647-659) */
/* outputs bitmask: ess() asks for ESS only, rhat() for RHAT only, ess_rhat() for both */
#define MDT_OUT_ESS 1
#define MDT_OUT_RHAT 2
/* autocov: the AbstractAutocovMethod subtype (src/ess_rhat.jl:2,38,55,73) */
#define MDT_AUTOCOV_DIRECT 0 /* AutocovMethod     (src/ess_rhat.jl:161-179) */
#define MDT_AUTOCOV_FFT 1    /* FFTAutocovMethod  (src/ess_rhat.jl:103-118,130-152,181-195) */
#define MDT_AUTOCOV_BDA 2    /* BDAAutocovMethod  (src/ess_rhat.jl:120-126,154-159,197-213) */

typedef struct mdt_ctx mdt_ctx; /* opaque: device, scratch, normal-score tables */

typedef struct mdt_args {
    int64_t draws, chains, params; /* size(x, 1), size(x, 2), prod(size(x)[3:end]) */
    int32_t dtype;                 /* MDT_F64 / MDT_F32: eltype of x and of the outputs */
    int32_t location;              /* MDT_HOST / MDT_DEVICE for x and all outputs (MDT_HOST: staged by the library and the call is
                                      synchronous; arrays of 128 MB and more travel in blocks of the params axis, copy under compute) */
    int32_t kind;                  /* MDT_KIND_* */
    int32_t outputs;               /* MDT_OUT_ESS | MDT_OUT_RHAT (mdt_ess_rhat only) */
    int32_t autocov;               /* MDT_AUTOCOV_* : keyword autocov_method */
    int32_t split_chains;          /* keyword split_chains (default 2) */
    int32_t relative;              /* keyword relative (default false) */
    int32_t sync;                  /* nonzero: synchronise the stream before returning */
    int64_t maxlag;                /* keyword maxlag (default 250) */
    double tail_prob;              /* keyword tail_prob (default 1//10) */
    void* stream;                  /* hipStream_t to enqueue on, or NULL for the default stream */
    int64_t chain_stride;          /* elements between the first draws of consecutive chains; 0: draws */
    int64_t param_stride;          /* elements between consecutive parameters; 0: draws*chains */
    const void* const* x_shards;   /* multi-GPU context only, or NULL: the input PRE-SHARDED -- x_shards[i] is block i of the params axis
                                      (mdt_shard_bounds(params, i, ndev)), resident on device_ids[i] (MDT_DEVICE) or in host memory
                                      (MDT_HOST), same draws / chains / strides as `args` says; the `x` argument is then ignored */
    double tail_p_lo, tail_p_hi;   /* both nonzero: the two quantile probabilities of ess(kind = :tail) themselves -- pl, pu of
                                      src/ess_rhat.jl:305-307 as the caller evaluated them in ITS arithmetic (Julia: convert(T,
                                      tail_prob / 2), convert(T, 1 - tail_prob / 2) with a Rational tail_prob); 0, 0: derived here as
                                      tail_prob / 2 and 1 - tail_prob / 2 in double (one ulp from the Rational result at most) */
} mdt_args;

int mdt_version(void);

/* context life cycle; device_id is the HIP device ordinal (one context per GPU and stream per process) */
int mdt_ctx_create(mdt_ctx** out, int device_id);
int mdt_ctx_destroy(mdt_ctx* ctx);
const char* mdt_last_error(const mdt_ctx* ctx);
/*
 * Optional: everything a request of this shape would otherwise allocate and build inside its FIRST call -- the per-workgroup
 * and per-wave scratch, the normal-score table of (draws * chains, dtype) (src/utils.jl:182, 189-193: built once per shape),
 * the ticket tables of `params` parameters -- so that the first mdt_ess_rhat / mdt_bulk_tail / ... on the context costs what
 * every later one costs.  A diagnostic is typically called once per array: call this when the sampler's shape is known.
 * Read from args: draws, chains, params, dtype, stream; split_chains (0: the keyword's default 2 -- it decides which kernel family
 * the shape is handed to and therefore which scratch is needed); chain_stride / param_stride (a strided layout -- MCMCChains
 * storage -- also gets its per-workgroup gather buffer); location (MDT_HOST: the whole-array staging buffers of a host request
 * below 128 MB; larger host arrays travel in blocks whose buffers the first request allocates beside its PCIe copy).  The rest is
 * ignored.  Not covered: the side tables of mdt_rhat_nested / mdt_mcse_quantile (a few KB, allocated by their first request).
 * The reference has no counterpart: its caches are the per-call temporaries of src/ess_rhat.jl:511 and src/utils.jl:170-171.
 * Returns MDT_OK, MDT_EARG, MDT_EDIM, MDT_EUNSUPPORTED (draws * chains beyond the ranking's range), MDT_ENOMEM, MDT_ERUNTIME.
 */
int mdt_ctx_reserve(mdt_ctx* ctx, const mdt_args* args);

/*
 * One context over several GPUs of the node: every parameter is independent (the `for` over
 * _eachparam, src/ess_rhat.jl:517), so mdt_ess_rhat (every kind and estimator) / mdt_bulk_tail / mdt_rhat_nested /
 * mdt_mcse_nested / mdt_mcse_mean / mdt_mcse_std / mdt_mcse_quantile on such a context cut the params axis into ndev
 * contiguous blocks -- block i = mdt_shard_bounds(params, i, ndev): the first params mod ndev blocks hold one parameter
 * more --, run block i on device_ids[i] concurrently (one persistent worker thread and one stream per device, no exchange
 * during compute) and leave every output vector complete in the caller's buffer.
 *   Input, three forms: (a) PRE-SHARDED, args->x_shards[i] = block i already resident on device_ids[i] -- no input moves
 *   at all; the only form for arrays that exceed one GPU (BASELINE config 4 is 320 GB); (b) location == MDT_HOST: each
 *   block is staged to its own GPU over PCIe (x or x_shards); (c) location == MDT_DEVICE with a single x: the array
 *   lives on ONE GPU (found from the pointer), the others fetch their block by a peer copy over xGMI -- a convenience
 *   for arrays that are already there, not a fast path.
 *   Ordering against the producers of the input: (c) waits for args->stream, which must be a stream of the GPU that holds
 *   x (NULL: that GPU's default stream); (a) has one producer per GPU and one stream slot, so the library synchronises
 *   each device_ids[i] (hipDeviceSynchronize, on the worker of that GPU) before block i is read: work queued on any stream
 *   of that GPU when the call is made is complete by then; work the caller queues later, from another thread, is not
 *   waited for.  (b) reads host memory: the caller's.
 *   Outputs: [params] vectors in host memory, or on one of the devices (found from the pointer); every block's results
 *   reach them by one copy per output (in place when the block's GPU owns them).
 * Strided layouts (chain_stride / param_stride) are honoured per block.  Calls on a multi-GPU context are synchronous
 * on return.  ndev == 1 is allowed.
 */
int mdt_ctx_create_multi(mdt_ctx** out, const int* device_ids, int ndev);

/* The partition rule of a multi-GPU context (and of the one-process-per-GPU mirror, dist.shard_bounds): block `shard`
 * of `nshards` is params [lo, hi).  Pure arithmetic, no GPU is touched. */
int mdt_shard_bounds(int64_t params, int32_t shard, int32_t nshards, int64_t* lo, int64_t* hi);

/* flags of mdt_ctx_status */
#define MDT_FLAG_RANK_OVERFLOW 1 /* draws*chains exceeded the ranking tables: the outputs of that parameter are NaN */
#define MDT_FLAG_NAN_QUANTILE 2  /* a quantile of data containing NaN (Statistics.quantile throws ArgumentError, used at src/ess_rhat.jl:655); the ESS is NaN */
#define MDT_FLAG_FDIST 4         /* gelmandiag: FDist would reject its parameter */
/*
 * Waits for the work enqueued through this context and returns the sticky condition flags raised by any launch
 * since the last clearing call -- the way a caller of sync == 0 requests learns what a synchronous call would have
 * returned as MDT_EARG / MDT_EUNSUPPORTED.  clear != 0 resets them.  A synchronous call that reports such an
 * error itself also clears the flag it reported.
 */
int mdt_ctx_status(mdt_ctx* ctx, int32_t* flags, int32_t clear);

/*
 * ess / rhat / ess_rhat  -- replaces src/ess_rhat.jl:276-311 (ess), :335-420 (rhat), :438-624
 * (ess_rhat) together with the whole-array transforms they call (src/utils.jl:148-193) and the
 * per-parameter kernels _rhat_basic! (:362-409) and _ess_rhat_basic! (:488-603).
 * ess_out / rhat_out: [params] of dtype, or NULL when the corresponding bit of `outputs` is 0.
 * niter = draws / split_chains <= 4 is NOT an error: ESS is filled with NaN and R-hat is still
 * computed (src/ess_rhat.jl:472-479; the shim issues the @warn).  maxlag <= 0 with niter > 4
 * and ESS requested -> MDT_EDOMAIN.
 */
int mdt_ess_rhat(mdt_ctx* ctx, const mdt_args* args, const void* x, void* ess_out, void* rhat_out);

/*
 * The BASELINE metric in one pass over x: ess_rhat(x; kind=:bulk) and ess_rhat(x; kind=:tail)
 * fused (src/ess_rhat.jl:604-616), sharing the pooled ranking of x.  Any output may be NULL.
 * rhat(kind=:rank) == max(rhat_bulk, rhat_tail) with Julia's NaN-propagating max (:416-420).
 */
int mdt_bulk_tail(mdt_ctx* ctx, const mdt_args* args, const void* x, void* ess_bulk, void* ess_tail,
                  void* rhat_bulk, void* rhat_tail);

/*
 * rhat_nested -- replaces src/rhat_nested.jl:83-188.  chain_inds is what
 * _validate_superchain_ids (:68-81) returns, made 0-based: a K x S column-major int32 matrix on
 * the HOST whose column k lists the chains of the k-th superchain (K*S == chains, S >= 2).
 */
int mdt_rhat_nested(mdt_ctx* ctx, const mdt_args* args, const void* x, const int32_t* chain_inds,
                    int32_t K, int32_t S, void* rhat_out);

/* mcse(x; kind=Statistics.mean) -- replaces src/mcse.jl:45-51 (basic ESS, then pooled corrected
 * std over all draws and chains divided by sqrt(ESS)).  args->kind is ignored. */
int mdt_mcse_mean(mdt_ctx* ctx, const mdt_args* args, const void* x, void* mcse_out);

/*
 * mcse(x; kind=Statistics.mean) AND rhat_nested(x, superchain_ids; kind) from one visit to every parameter (BASELINE
 * config 5 asks for both on the same array): replaces src/mcse.jl:45-51 and src/rhat_nested.jl:83-188 in one launch --
 * a parameter's slice is read from HBM once, the ranking passes that follow find it in the caches.  Same results, bit
 * for bit, as mdt_mcse_mean and mdt_rhat_nested called one after the other.  args->kind selects the nested R-hat
 * (MDT_KIND_BASIC .. MDT_KIND_RANK); autocov / maxlag / split_chains are those of mcse (rhat_nested has none of its
 * own besides split_chains).  chain_inds, K, S as for mdt_rhat_nested.
 */
int mdt_mcse_nested(mdt_ctx* ctx, const mdt_args* args, const void* x, const int32_t* chain_inds, int32_t K, int32_t S,
                    void* mcse_out, void* rhat_out);

/* mcse(x; kind=Statistics.std) -- replaces src/mcse.jl:52-65 (ESS of the proxy (x - mean)^2, then the delta
 * method sqrt((E[mu4]/E[var] - E[var]) / S) / 2).  args->kind is ignored. */
int mdt_mcse_std(mdt_ctx* ctx, const mdt_args* args, const void* x, void* mcse_out);

/* mcse(x; kind = Statistics.median) and mcse(x; kind = Base.Fix2(Statistics.quantile, p)) -- replaces
 * src/mcse.jl:66-118: ESS of the indicator proxy (pass 1), the Beta(S p + 1, S (1-p) + 1) quadrature points by one
 * device thread per parameter, two order statistics per parameter through a second, order-statistics-only ranking
 * (pass 2); all three on args->stream, nothing returns to the host in between.
 * args->kind: MDT_KIND_MEDIAN or MDT_KIND_QUANTILE (p = args->tail_prob). */
int mdt_mcse_quantile(mdt_ctx* ctx, const mdt_args* args, const void* x, void* mcse_out);

/*
 * Storage-order helper: MCMCChains keeps (iterations, parameters, chains); every entry
 * point above wants (draws, chains, params).  One coalesced device pass, out of place:
 * x_dpc[d + draws*(p + params*c)] -> x_dcp[d + draws*(c + chains*p)], instead of a host-side permutedims.
 * args: draws, chains, params, dtype, stream, sync; location must be MDT_DEVICE.
 */
int mdt_layout_from_dpc(mdt_ctx* ctx, const mdt_args* args, const void* x_dpc, void* x_dcp);

/*
 * bfmi(energy; dims = 1) -- replaces src/bfmi.jl:38-45: per chain,
 * mean(abs2, diff(energy)) / var(energy).  energy is (draws, chains[, params]) column-major; one value per
 * column, bfmi_out[chains*params] in dtype.  One streaming pass (plus an L2-resident second pass for the
 * centred sum).  args: draws, chains, params, dtype, location, stream, sync.
 */
int mdt_bfmi(mdt_ctx* ctx, const mdt_args* args, const void* energy, void* bfmi_out);

/*
 * gelmandiag(samples; alpha) -- replaces the univariate parts of src/gelmandiag.jl:1-60, 76-80: the potential
 * scale reduction factor and its upper (1 - alpha/2) confidence limit per parameter.  One streaming pass for the
 * per-chain means and variances, then one thread per parameter for the few flops that follow and the
 * F-distribution quantile (Distributions.quantile(FDist(...)), :50); nothing runs on the host.  psrf / psrfci:
 * [params] Float64 whatever the input dtype (:41-42), host or device per args->location.  chains < 2 -> MDT_EARG
 * (:3).  gelmandiag_multivariate (:86-113, a params x params eigenproblem) is not provided.
 */
int mdt_gelmandiag(mdt_ctx* ctx, const mdt_args* args, const void* x, double alpha, double* psrf, double* psrfci);

/*
 * Stage output for parity checks: _rank_normalize (src/utils.jl:169-184) of x (or, with
 * fold != 0, of _fold_around_median(x), :148-158).  rank_out: [draws*chains*params] float64
 * tied ranks (StatsBase.tiedrank) or NULL; z_out: same length, dtype, normal scores, or NULL.
 */
int mdt_rank_normalize(mdt_ctx* ctx, const mdt_args* args, const void* x, int32_t fold,
                       double* rank_out, void* z_out);

/*
 * Diagnostics (no reference counterpart): per-phase shader-clock totals summed over workgroups,
 * 16 slots (see the PH_* enum in csrc/mdt_device.hpp).  Only available when the context was
 * created with MDT_PROF=1 in the environment; the counters cost one branch per phase otherwise.
 */
int mdt_debug_phase_cycles(mdt_ctx* ctx, uint64_t* out16, int32_t reset);

/*
 * Diagnostics, host only (no GPU is touched): the library's regularised incomplete beta function I_v(a, b)
 * (inverse == 0) or its inverse in v (inverse != 0) -- the code behind the quadrature points of
 * src/mcse.jl:105-106 and the F quantile of src/gelmandiag.jl:50 (csrc/mdt_beta.hpp, compiled for host and
 * device from one source), so that it can be checked against an independent implementation.
 */
int mdt_debug_beta(double a, double b, double v, int32_t inverse, double* out);

#ifdef __cplusplus
}
#endif
#endif /* MDT_HIP_H */
