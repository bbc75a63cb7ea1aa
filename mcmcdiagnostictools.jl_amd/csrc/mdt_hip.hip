// mdt_hip.hip -- C-ABI of libmdt_hip.so (see include/mdt_hip.h) around the kernels in
// mdt_device.hpp.  Host responsibilities: argument validation with the reference's error
// behaviour, evaluation of the few scalars the reference computes in eltype(x) (log10(ntotal),
// the type-7 quantile index and gamma), scratch and normal-score-table management, launch.
//
// Build (see __graft_entry__.build / csrc/Makefile):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -I include -o libmdt_hip.so mdt_hip.hip
// -ffp-contract=off: the scalar formulas that mirror Julia expressions (quantile, median,
// var+, tau) must round like Julia does; hot loops use explicit fma().
// the device code, once per workgroup geometry (see the head of mdt_device.hpp)
#define MDT_NS mdt_big
#define MDT_NT 1024
#define MDT_LDS_LIMIT (160 * 1024)
#define MDT_WAVES_PER_EU 4
#include "mdt_device.hpp"
#undef MDT_NS
#undef MDT_NT
#undef MDT_LDS_LIMIT
#undef MDT_WAVES_PER_EU
#define MDT_NS mdt_small
#define MDT_NT 512
#define MDT_LDS_LIMIT (80 * 1024)
#define MDT_WAVES_PER_EU 4
#include "mdt_device.hpp"
#undef MDT_NS
#undef MDT_NT
#undef MDT_LDS_LIMIT
#undef MDT_WAVES_PER_EU
#define MDT_NS mdt_tiny
#define MDT_NT 256
#define MDT_LDS_LIMIT (40 * 1024)
#define MDT_WAVES_PER_EU 4
#include "mdt_device.hpp"
#undef MDT_NS
#undef MDT_NT
#undef MDT_LDS_LIMIT
#undef MDT_WAVES_PER_EU
// (a fourth geometry -- 1024 threads, 80 KiB, 8 waves/SIMD at 64 VGPRs -- compiles from the same header and
// passes every test; measured 23.7 ms against 18.3 ms on the headline shape: the barrier-heavy partition and
// the register-hungry lag products lose more than the count / histogram passes gain.  Not instantiated.)
#include "mdt_hip.h"
#include "mdt_beta.hpp"
#include "mdt_wave.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <atomic>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

using mdtc::KP;

// worker threads of a multi-GPU context: created once, parked on a condition variable between requests
struct MultiPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    std::function<int(int)> job; // shard index -> status
    std::vector<int> rcs;
    unsigned long long epoch = 0;
    int pending = 0;
    bool stop = false;
    explicit MultiPool(int n) : rcs(n, MDT_OK) {
        for (int i = 0; i < n; ++i) threads.emplace_back([this, i] { loop(i); });
    }
    void loop(int i) {
        unsigned long long seen = 0;
        for (;;) {
            std::function<int(int)> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || epoch != seen; });
                if (stop) return;
                seen = epoch;
                j = job;
            }
            const int rc = j(i);
            {
                std::lock_guard<std::mutex> lk(mu);
                rcs[i] = rc;
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    void run(const std::function<int(int)>& j) { // every worker runs j(its index); returns when all are back
        std::unique_lock<std::mutex> lk(mu);
        job = j;
        pending = (int)threads.size();
        ++epoch;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~MultiPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : threads) t.join();
    }
};

struct mdt_ctx {
    int device = 0;
    int num_cu = 0;
    int grid = 0;      // workgroups of the small geometry: 2 per CU (big: 1 per CU, tiny: 4 per CU)
    int force_cfg = 0; // MDT_CONFIG=big|small|tiny (tests, A/B timing): 1 big, 2 small, 3 tiny, 0 automatic
    std::string err;
    // per-workgroup scratch
    void* scratch_slab = nullptr; // ONE allocation behind the five per-workgroup scratch arrays below (round 6: five hipMalloc calls were ~8 ms of a first call)
    unsigned int* r2buf = nullptr;
    double* zbuf = nullptr;
    double* xsbuf = nullptr;
    unsigned short* bsbuf = nullptr;
    unsigned short* binbuf = nullptr;
    long long n_pad = 0;
    long long scratch_elems = 0; // capacity of each scratch array, in elements (workgroups x padded slice)
    // normal-score table
    double* ztab = nullptr;
    long long ztab_n = -1;
    int ztab_f32 = -1;
    long long ztab_cap = 0;
    // several GPUs of one node (mdt_ctx_create_multi): one child context per device, params cut into contiguous blocks
    std::vector<mdt_ctx*> children;
    MultiPool* pool = nullptr;        // one worker thread per child, alive as long as the context
    // the stream the last request went to (a context serves one stream at a time; mdt_ctx_status waits on it)
    hipStream_t last_stream = nullptr;
    bool used = false;
    // strided input: per-workgroup dense copy of the parameter being worked on
    void* xcbuf = nullptr;
    long long xc_elems = 0;
    // tail-ESS indicator bits of every split chain (taken from the z-series), per workgroup
    unsigned long long* ibits = nullptr;
    long long ibits_stride = 0, ibits_cap = 0;
    // status words (flags of the launch, ticket, sticky flags) + nested chain indices
    int* status = nullptr;
    unsigned long long* prof = nullptr; // [2*grid*16] phase cycle counters when MDT_PROF is set
    // ticket order of the persistent launch: cost[p] = duration of parameter p in the last launch, order = parameters by
    // decreasing cost (order_kernel, queued behind every launch); used by the next launch with the same number of parameters
    unsigned int* cost = nullptr;
    unsigned int* order = nullptr;
    unsigned int* order_sorted = nullptr; // work area of order_kernel: the parameters by decreasing cost
    long long order_cap = 0;
    long long order_P = -1;             // the order on the device is a permutation of [0, order_P); -1: none
    // wave-per-parameter kernel (mdt_wave.hip): per-wave scratch, and the parameters it leaves to the general kernel
    void* wv_scratch = nullptr;
    size_t wv_cap = 0;
    unsigned int* fb_list = nullptr;
    long long fb_cap = 0;
    KP* kp_ring = nullptr;              // device copies of the kernel parameters, one slot per launch
    unsigned kp_next = 0;
    int* chain_inds = nullptr;
    int chain_inds_cap = 0;
    // mcse quantile: ESS of the first pass and the order-statistic positions of the second
    void* ess_tmp = nullptr;
    unsigned int* tpos_arr = nullptr;
    long long mq_cap = 0;
    // bfmi / gelmandiag: (mean, variance, mean squared difference) of every column
    double* mom = nullptr;
    long long mom_cap = 0;
    // host staging
    void* stage_x = nullptr;
    size_t stage_x_cap = 0;
    // host buffers handed over in blocks of the params axis (run_host_blocks): copies on their own stream, one event per block
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[16] = {};
    void* stage_out = nullptr;
    size_t stage_out_cap = 0;
    // slot k of the output staging area of a multi-GPU block of cnt values (16-byte aligned slots)
    void* stage_out_multi(int k, long long cnt, size_t ts) const { return (char*)stage_out + (size_t)k * ((((size_t)cnt * ts) + 15) & ~(size_t)15); }
};

namespace {

// no entry point leaves the calling thread on another device than it came with (PyTorch and HIP.jl read the
// current device from the runtime)
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define MDT_ON_DEVICE(ctx)                                                                     \
    DeviceGuard guard_((ctx)->device);                                                         \
    if (!guard_.ok) return fail((ctx), MDT_ERUNTIME, "hipSetDevice failed")

// (draws, params, chains) -> (draws, chains, params): both sides are runs of `draws` contiguous values
template <typename T>
__global__ void permute_dpc_kernel(const T* __restrict__ in, T* __restrict__ out, long long D, long long C, long long P) {
    const long long run = blockIdx.x; // = p + P * c in the source
    const long long c = run / P, p = run - c * P;
    const T* src = in + run * D;
    T* dst = out + (c + C * p) * D;
    for (long long d = threadIdx.x; d < D; d += blockDim.x) dst[d] = src[d];
}


// (count, mean, sum of squared deviations) of two disjoint groups -> of their union (Chan, Golub & LeVeque 1979):
// as accurate as the two-pass formula, and every value is read from memory once
__device__ inline void chan_merge(double& na, double& ma, double& Ma, double nb, double mb, double Mb) {
    if (nb == 0.0) return;
    const double n = na + nb, d = mb - ma;
    ma = ma + d * (nb / n);
    Ma = Ma + Mb + d * d * (na * nb / n);
    na = n;
}

// One workgroup per column of `draws` contiguous values, ONE pass over HBM: every thread keeps 16 strided values
// of a 4096-value chunk in registers (16 independent loads in flight), takes their mean and squared deviations
// exactly as the two-pass formula would and merges that into its running (n, mean, M2); the 256 partial
// triples are merged by a fixed tree at the end.  MSD adds sum (x[t+1] - x[t])^2: the right neighbour comes
// from the next lane, lane 63 loads it.  Output: bfmi_out[col] = msd / var (src/bfmi.jl:39) when given, else
// mom[3 col + {0, 1, 2}] = mean, corrected variance, mean squared difference (src/gelmandiag.jl:10-13).
// Streaming, HBM-bound: draws * sizeof(T) algorithmic bytes per column.
template <typename T, bool MSD>
__global__ __launch_bounds__(256) void column_moments_kernel(const T* __restrict__ x, long long D, long long ncol,
                                                             double* __restrict__ mom, T* __restrict__ bfmi_out) {
    constexpr int IPT = 16;
    __shared__ double red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (long long col = blockIdx.x; col < ncol; col += gridDim.x) {
        const T* xc = x + col * D;
        double rn = 0.0, rmean = 0.0, rm2 = 0.0, q = 0.0;
        for (long long c0 = 0; c0 < D; c0 += 256 * IPT) {
            double v[IPT];
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const long long i = c0 + k * 256 + tid;
                v[k] = (double)xc[i < D ? i : D - 1];
            }
            double s = 0.0, cn = 0.0;
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const bool ok = c0 + k * 256 + tid < D;
                s += ok ? v[k] : 0.0;
                cn += ok ? 1.0 : 0.0;
            }
            const double cm = cn > 0.0 ? s / cn : 0.0;
            double m2 = 0.0;
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const bool ok = c0 + k * 256 + tid < D;
                const double e = v[k] - cm;
                m2 += ok ? e * e : 0.0;
            }
            chan_merge(rn, rmean, rm2, cn, cm, m2);
            if (MSD) {
#pragma unroll
                for (int k = 0; k < IPT; ++k) {
                    const long long i = c0 + k * 256 + tid;
                    // value of lane + 1 through the cross-lane data path (DPP wave_shl:1), no index registers
                    const long long vb = __double_as_longlong(v[k]);
                    const int lo = __builtin_amdgcn_update_dpp(0, (int)vb, 0x130, 0xf, 0xf, false);
                    const int hi = __builtin_amdgcn_update_dpp(0, (int)(vb >> 32), 0x130, 0xf, 0xf, false);
                    double nb = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
                    if (lane == 63) nb = (double)xc[i + 1 < D ? i + 1 : D - 1];
                    const double d = nb - v[k];
                    q += (i + 1 < D) ? d * d : 0.0;
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { // fixed merge tree over the lanes
            const double nb = __shfl_xor(rn, o), mb = __shfl_xor(rmean, o), Mb = __shfl_xor(rm2, o);
            // both partners must end with the same triple: merge (lower lane, upper lane) in that order
            double na = rn, ma = rmean, Ma = rm2, n2 = nb, m2b = mb, M2b = Mb;
            if (lane & o) { na = nb; ma = mb; Ma = Mb; n2 = rn; m2b = rmean; M2b = rm2; }
            chan_merge(na, ma, Ma, n2, m2b, M2b);
            rn = na; rmean = ma; rm2 = Ma;
            if (MSD) q += __shfl_xor(q, o);
        }
        if (lane == 0) { red[0][w] = rn; red[1][w] = rmean; red[2][w] = rm2; red[3][w] = q; }
        __syncthreads();
        if (tid == 0) {
            double n = red[0][0], mu = red[1][0], M2 = red[2][0];
            for (int k = 1; k < 4; ++k) chan_merge(n, mu, M2, red[0][k], red[1][k], red[2][k]);
            const double dm1 = (double)(D - 1);
            const double var = M2 / dm1;
            const double msd = ((red[3][0] + red[3][1]) + (red[3][2] + red[3][3])) / dm1;
            if (bfmi_out) bfmi_out[col] = (T)(msd / var); // mean(abs2, diff(energy)) / var(energy), src/bfmi.jl:39
            else { mom[3 * col] = mu; mom[3 * col + 1] = var; mom[3 * col + 2] = msd; }
        }
        __syncthreads();
    }
}

// One thread per parameter: the flops that follow the per-chain moments (src/gelmandiag.jl:5-57, the diagonal
// entries only) and the F quantile of :50, quantile(FDist(d1, d2), q) = d2 y / (d1 (1 - y)) with
// y = betainvcdf(d1/2, d2/2, q).  All Float64 (:41-42).  status |= 4 when FDist would reject its parameter.
__global__ __launch_bounds__(64) void gelman_finish_kernel(const double* __restrict__ mom, long long D, long long C, long long P,
                                                           double alpha, double zq, double* __restrict__ psrf,
                                                           double* __restrict__ psrfci, int* __restrict__ status) {
    const long long p = (long long)blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    const double niters = (double)D, nchains = (double)C;
    const double rfixed = (niters - 1) / niters, rrandomscale = (nchains + 1) / (nchains * niters);
    const double q = 1 - alpha / 2, B_df = nchains - 1;
    const double* mp = mom + 3 * C * p;
    double w = 0, pbar2 = 0, mq = 0; // means over the chains of s2, psibar, psibar^2
    for (long long c = 0; c < C; ++c) { pbar2 += mp[3 * c]; w += mp[3 * c + 1]; mq += mp[3 * c] * mp[3 * c]; }
    w /= nchains; pbar2 /= nchains; mq /= nchains;
    double bb = 0, vw = 0, c_sq = 0, c_s = 0; // corrected (co)variances over the chains
    for (long long c = 0; c < C; ++c) {
        const double dm = mp[3 * c] - pbar2, ds = mp[3 * c + 1] - w;
        bb += dm * dm; vw += ds * ds;
        c_sq += ds * (mp[3 * c] * mp[3 * c] - mq);
        c_s += ds * dm;
    }
    const double b = niters * bb / (nchains - 1);
    const double var_w = vw / (nchains - 1) / nchains;
    const double var_b = (2 / (nchains - 1)) * b * b;
    const double var_wb = (niters / nchains) * (c_sq / (nchains - 1) - 2 * pbar2 * (c_s / (nchains - 1)));
    const double V = rfixed * w + rrandomscale * b;
    const double var_V = rfixed * rfixed * var_w + rrandomscale * rrandomscale * var_b + 2 * rfixed * rrandomscale * var_wb;
    const double df = 2 * V * V / var_V, W_df = 2 * w * w / var_w;
    const double correction = (df + 3) / (df + 1);
    double rrandom = rrandomscale * b / w;
    psrf[p] = sqrt(correction * (rfixed + rrandom));
    if (rrandom == rrandom) {
        if (!(W_df > 0)) { atomicOr(status, 4); psrfci[p] = __builtin_nan(""); return; }
        // start: Wilson-Hilferty chi-square quantile for the numerator, F ~ chi2(d1)/d1 for large d2
        const double t9 = 2.0 / (9.0 * B_df), wh = 1.0 - t9 + zq * sqrt(t9);
        const double f0 = wh * wh * wh;
        const double y = mdt_betainvcdf(B_df / 2, W_df / 2, q, f0 > 0.0 ? B_df * f0 / (W_df + B_df * f0) : -1.0);
        rrandom *= W_df * y / (B_df * (1 - y));
    }
    psrfci[p] = sqrt(correction * (rfixed + rrandom));
}

// mcse(kind = median | quantile(p)), src/mcse.jl:93-108: S -> 0-based positions of the order statistics at the two
// quadrature points of Beta(S p + 1, S (1 - p) + 1); one thread per parameter (a thread per parameter AND point measured 2.2 ms against
// 1.6 per 100 000 parameters: the inversions are bound by Float64 issue, not by their latency; round 6).  NaN ESS -> sentinel positions.
template <typename T>
__global__ __launch_bounds__(64) void mcse_quantile_pos_kernel(const T* __restrict__ ess, long long P, double p, long long n,
                                                               unsigned int* __restrict__ tp) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
    if (i >= P) return;
    const double S = (double)ess[i];
    if (S != S) { tp[2 * i] = 0xFFFFFFFFu; tp[2 * i + 1] = 0xFFFFFFFFu; return; }
    const double alpha = S * p + 1.0, beta = S * (1.0 - p) + 1.0;
    const double pu = mdt_betainvcdf(alpha, beta, 0.8413447460685429);  // normcdf(1)
    const double pl = mdt_betainvcdf(alpha, beta, 0.15865525393145705); // normcdf(-1)
    long long l = (long long)floor(pl * (double)n), u = (long long)ceil(pu * (double)n);
    if (l < 1) l = 1;
    if (u > n) u = n;
    tp[2 * i] = (unsigned int)(l - 1);
    tp[2 * i + 1] = (unsigned int)(u - 1);
}

// Launch prologue on the compute queue: the parameter block arrives as a kernel argument and is stored into its
// ring slot, the ticket (and, unless a first pass left flags to report, the status word) is cleared.  Everything is
// ordered by the stream, so a slot is never rewritten while a launch that reads it is still queued in front.  One small
// kernel instead of a memset and a host-to-device copy: no hand-over between the copy engine and the compute
// queue in front of every launch.
__global__ void kp_setup_kernel(KP* slot, KP v, int* status, int keep_status) {
    if (threadIdx.x == 0) {
        *slot = v;
        // the flags of the previous launch become sticky (mdt_ctx_status) before this launch starts with a clean word;
        // a second pass keeps the word: its caller reports the flags of both passes
        // (keep_status bit 1: the launch works off the list the wave-per-parameter kernel in front of it left: its count stays)
        if (!(keep_status & 1)) { status[2] |= status[0]; status[0] = 0; }
        if (!(keep_status & 2)) status[3] = 0;
        // keep_status bit 2: the parameters the wave-per-parameter kernel left behind (status[3] of them, listed in the `order` table
        // of this slot) become the tickets of the general kernel queued behind it -- a count only the device knows
        if (keep_status & 4) slot->P = status[3];
        status[1] = 0;
    }
}

// the same bookkeeping for the entry points that do not go through mdt_kernel (gelmandiag)
__global__ void status_begin_kernel(int* status) {
    if (threadIdx.x == 0) { status[2] |= status[0]; status[0] = 0; status[1] = 0; }
}

// Parameters by decreasing cost: a counting sort over ORDER_BINS linear cost classes by one workgroup (three passes over
// the costs; P = 10 000: ~10 us, queued BEHIND the launch that measured them).  Inside a class the order is whatever the
// atomics give -- any permutation is a valid ticket order.  Only the END of the ticket sequence has to be sorted (the
// cheapest `tail` parameters, costliest of them first: that is what shortens the tail of the launch); the tickets before
// it are dealt in a strided walk over the sorted sequence instead, so that neighbouring tickets differ in cost: workgroups
// that all work on equally costly parameters stay in lockstep -- every CU in the same phase at the same time, all of them
// on HBM or all of them on the LDS -- and lose what two workgroups per CU are there for.
constexpr int ORDER_BINS = 1024;
__global__ __launch_bounds__(1024) void order_kernel(const unsigned int* __restrict__ cost, unsigned int* __restrict__ sorted,
                                                     unsigned int* __restrict__ order, long long P, long long tail, long long stride) {
    __shared__ unsigned cmax, lh[ORDER_BINS + 1];
    const int tid = threadIdx.x;
    if (tid == 0) cmax = 0u;
    for (int b = tid; b <= ORDER_BINS; b += 1024) lh[b] = 0u;
    __syncthreads();
    unsigned m = 0u;
    for (long long i = tid; i < P; i += 1024) m = cost[i] > m ? cost[i] : m;
    atomicMax(&cmax, m);
    __syncthreads();
    const unsigned long long cm = (unsigned long long)cmax + 1ull;
    // class 0 = the costliest
    auto cls = [&](unsigned c) -> unsigned { return (unsigned)(ORDER_BINS - 1) - (unsigned)(((unsigned long long)c * ORDER_BINS) / cm); };
    for (long long i = tid; i < P; i += 1024) atomicAdd(&lh[cls(cost[i])], 1u);
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0u;
        for (int b = 0; b < ORDER_BINS; ++b) { const unsigned c = lh[b]; lh[b] = run; run += c; }
    }
    __syncthreads();
    for (long long i = tid; i < P; i += 1024) sorted[atomicAdd(&lh[cls(cost[i])], 1u)] = (unsigned)i;
    __syncthreads(); // (one workgroup: its own global stores are visible to it after the barrier)
    const long long head = P - tail; // stride is coprime with head: i -> i * stride mod head is a permutation
    for (long long i = tid; i < P; i += 1024) order[i] = (i < head) ? sorted[(unsigned long long)i * (unsigned long long)stride % (unsigned long long)head] : sorted[i];
}

constexpr unsigned KP_RING = 64;

// Test and A/B switches (MDT_CONFIG, MDT_WAVE, MDT_RANK_VARIANT, ...) change which of several bit-compatible paths serves a
// request; some of them change the last bits of an ESS.  A stray variable in a user's shell must not do that: the switches
// are read only when MDT_TEST_HOOKS=1 was set when the library was first used (tests/conftest.py sets it).
const char* hook(const char* name) {
    static const bool on = [] { const char* e = getenv("MDT_TEST_HOOKS"); return e && *e && strcmp(e, "0") != 0; }();
    return on ? getenv(name) : nullptr;
}

int fail(mdt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, e_ == hipErrorOutOfMemory ? MDT_ENOMEM : MDT_ERUNTIME,            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

size_t tsize(int dtype) { return dtype == MDT_F32 ? 4 : 8; }

// per-workgroup dense copy of a parameter whose chains are not contiguous in x (MCMCChains storage): grown lazily
int ensure_xcbuf(mdt_ctx* ctx, long long wgs) {
    const long long need = wgs * ctx->n_pad;
    if (need > ctx->xc_elems) {
        if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
        if (ctx->xcbuf) (void)hipFree(ctx->xcbuf);
        ctx->xcbuf = nullptr; ctx->xc_elems = 0;
        HIP_TRY(hipMalloc(&ctx->xcbuf, (size_t)need * 8));
        ctx->xc_elems = need;
    }
    return MDT_OK;
}

// the diagnostics of the wave-per-parameter kernel, read behind the test-hooks gate (all zero without it)
mdtw::Hooks wave_hooks() {
    mdtw::Hooks h;
    if (const char* e = hook("MDT_WAVE_WAVES")) h.waves = atoi(e);
    if (const char* e = hook("MDT_WAVE_TAB")) h.tab_mode = atoi(e);
    if (const char* e = hook("MDT_WAVE_STOP")) h.stop = atoi(e);
    if (const char* e = hook("MDT_WAVE_OSONLY")) h.osonly = atoi(e);
    if (const char* e = hook("MDT_WAVE_DBG")) h.dbg = (unsigned*)strtoull(e, nullptr, 0); // (only a -DWV_DEBUG / -DWV_PROF build of mdt_wave.hip stores through it)
    return h;
}

int ensure_scratch(mdt_ctx* ctx, long long n, long long wgs) {
    const long long n_pad = ((n + 63) / 64) * 64;
    const long long need = wgs * n_pad;
    if (need > ctx->scratch_elems) {
        if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream)); // launches still queued work in the old buffers
        if (ctx->scratch_slab) (void)hipFree(ctx->scratch_slab);
        ctx->scratch_slab = nullptr;
        ctx->r2buf = nullptr; ctx->binbuf = nullptr; ctx->zbuf = nullptr; ctx->xsbuf = nullptr; ctx->bsbuf = nullptr; ctx->scratch_elems = 0;
        // one slab, the arrays at 256-byte boundaries inside it (widest element type first)
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t bz = up((size_t)need * sizeof(double)), br = up((size_t)need * sizeof(unsigned int)), bb = up((size_t)need * sizeof(unsigned short));
        HIP_TRY(hipMalloc(&ctx->scratch_slab, 2 * bz + br + 2 * bb));
        char* base = (char*)ctx->scratch_slab;
        ctx->zbuf = (double*)base;
        ctx->xsbuf = (double*)(base + bz);
        ctx->r2buf = (unsigned int*)(base + 2 * bz);
        ctx->binbuf = (unsigned short*)(base + 2 * bz + br);
        ctx->bsbuf = (unsigned short*)(base + 2 * bz + br + bb);
        ctx->scratch_elems = need;
    }
    ctx->n_pad = n_pad; // the stride of this request: every workgroup owns [blockIdx * n_pad, +n_pad)
    ctx->ibits_stride = n_pad / 32 + 256; // 2 indicators x (n / 64 + one word per split chain) words
    if (wgs * ctx->ibits_stride > ctx->ibits_cap) {
        if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
        if (ctx->ibits) (void)hipFree(ctx->ibits);
        ctx->ibits = nullptr; ctx->ibits_cap = 0;
        HIP_TRY(hipMalloc((void**)&ctx->ibits, (size_t)(wgs * ctx->ibits_stride) * sizeof(unsigned long long)));
        ctx->ibits_cap = wgs * ctx->ibits_stride;
    }
    return MDT_OK;
}

int ensure_ztab(mdt_ctx* ctx, long long n, int f32, hipStream_t st) {
    if (ctx->ztab_n == n && ctx->ztab_f32 == f32) return MDT_OK;
    const long long need = 3 * n + 2; // by doubled rank [0, 2n] + dense copy for whole ranks
    if (need > ctx->ztab_cap) {
        if (ctx->ztab) (void)hipFree(ctx->ztab);
        ctx->ztab = nullptr; ctx->ztab_cap = 0;
        // (+ 32 entries of slack: the fold merge reads the scores of a thread's positions as aligned pairs, past the last rank at the end of a slice)
        HIP_TRY(hipMalloc((void**)&ctx->ztab, (size_t)(need + 32) * sizeof(double)));
        ctx->ztab_cap = need;
    }
    const int bs = 256;
    const long long nb = (need + bs - 1) / bs;
    hipLaunchKernelGGL(mdt_big::ztab_kernel, dim3((unsigned)nb), dim3(bs), 0, st, ctx->ztab, n, f32);
    HIP_TRY(hipGetLastError());
    ctx->ztab_n = n;
    ctx->ztab_f32 = f32;
    return MDT_OK;
}

// device staging buffers of a host-memory request, grown lazily
int stage_x_bytes(mdt_ctx* ctx, size_t bytes) {
    if (bytes > ctx->stage_x_cap) {
        if (ctx->stage_x) (void)hipFree(ctx->stage_x);
        ctx->stage_x = nullptr; ctx->stage_x_cap = 0;
        HIP_TRY(hipMalloc(&ctx->stage_x, bytes));
        ctx->stage_x_cap = bytes;
    }
    return MDT_OK;
}
int stage_out_bytes(mdt_ctx* ctx, size_t bytes) {
    if (bytes > ctx->stage_out_cap) {
        if (ctx->stage_out) (void)hipFree(ctx->stage_out);
        ctx->stage_out = nullptr; ctx->stage_out_cap = 0;
        HIP_TRY(hipMalloc(&ctx->stage_out, bytes));
        ctx->stage_out_cap = bytes;
    }
    return MDT_OK;
}

// what a request needs, common to all entry points
struct Req {
    int mode = 0;
    int basic = 0;
    int need_eb = 0, need_rb = 0, need_et = 0, need_rt = 0;
    void* out_eb = nullptr; void* out_rb = nullptr; void* out_et = nullptr; void* out_rt = nullptr;
    void* out_rrank = nullptr;
    void* out_mcse = nullptr; // mode 6
    double* rank_out = nullptr; void* z_out = nullptr;
    int fold = 0;
    int est = 0; // ess(kind = estimator): MDT_KIND_MEDIAN .. MDT_KIND_QUANTILE
    const int32_t* chain_inds = nullptr; int K = 0, S = 0;
    const unsigned int* tpos_arr = nullptr; // mode 5
    int want_ess = 0; // ESS of some kind is requested (drives the maxlag domain check)
    int keep_status = 0; // second pass of a two-pass request: the flags of the first pass are still to be reported
};

// n = draws * chains without overflow; false: beyond what the ranking can index (10^9 values per parameter)
bool slice_size(long long D, long long C, long long* n) {
    const long long lim = 2000000000LL / 2;
    if (D < 0 || C < 0 || D > lim || C > lim) return false;
    *n = D * C; // (<= 10^18)
    return *n <= lim;
}
// The shape half of the dispatch rules -- which kernel family a (slice, parameter count) can be handed to -- shared by run() and
// mdt_ctx_reserve, so that a reservation allocates what the request will use and nothing else.
// Wave-per-parameter kernel (mdt_wave.hip): the slice fits it AND there are enough parameters to keep ~3000 waves busy for a few
// tickets each (measured crossovers: run()).
bool wave_shape_fits(long long n, long long niter, long long m) {
    return n >= 256 && n <= (long long)mdtw::WV_MAXN && m <= mdtw::WV_MAXM && niter >= 8 && niter <= (long long)mdtw::WV_NITER;
}
bool wave_shape_pays(long long n, long long niter, long long P) {
    return P >= ((n > 4096 || niter > (long long)mdtw::WV_NITER_SHORT) ? 3 * n : 700 + (4 * n) / 3);
}
// ... and for raw draws (kind = basic, mcse(mean); round 6).  Measured crossovers (profiles/r06_wave_raw_crossover.txt; ms, wave kernel /
// workgroup-per-parameter kernels): with the autocovariance 1000 x 4 at ~7000 parameters (4000: 0.42 / 0.34, 8000: 0.49 / 0.53, 100 000:
// 2.40 / 4.70), 250 x 4 at ~2000 (0.22 / 0.22; 200 000: 2.60 / 8.22), 1000 x 8 at ~8000 (0.93 / 0.95; 50 000: 3.09 / 4.73); R-hat alone
// from ~2000 parameters at every shape (1000 x 4 x 100 000: 1.37 / 3.54)
bool wave_raw_pays(long long n, long long P, bool acov) { return P >= (acov ? 1500 + n : 2000); }
// tickets by measured cost (order_kernel): only when a workgroup gets more than one ticket
bool lpt_shape(long long P, long long wgs) { return P > wgs && P < 0xFFFFFFFFLL; }

int run_multi(mdt_ctx* ctx, const mdt_args* a, const void* x, const Req& rq);
int run_host_blocks(mdt_ctx* ctx, const mdt_args* a, const void* x, const Req& rq, int nb);
void shard_bounds(long long P, int i, int n, long long* lo, long long* hi);

int run(mdt_ctx* ctx, const mdt_args* a, const void* x, Req rq) {
    if (!ctx) return MDT_EARG;
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    if (!ctx->children.empty()) return run_multi(ctx, a, x, rq);
    if (a->dtype != MDT_F64 && a->dtype != MDT_F32) return fail(ctx, MDT_EARG, "dtype must be MDT_F64 or MDT_F32");
    if (a->location != MDT_HOST && a->location != MDT_DEVICE) return fail(ctx, MDT_EARG, "bad location");
    if (a->draws < 0 || a->chains < 0 || a->params < 0) return fail(ctx, MDT_EDIM, "negative dimension");
    if (a->split_chains < 1) return fail(ctx, MDT_EARG, "split_chains must be >= 1");
    if (a->autocov < 0 || a->autocov > 2) return fail(ctx, MDT_EARG, "unknown autocov method");
    const long long D = a->draws, C = a->chains, P = a->params, s = a->split_chains;
    if (P == 0) return MDT_OK;
    if (!x) return fail(ctx, MDT_EARG, "x is NULL");
    if (D == 0 || C == 0) return fail(ctx, MDT_EDIM, "empty sample dimensions");
    long long n = 0;
    if (!slice_size(D, C, &n)) return fail(ctx, MDT_EUNSUPPORTED, "draws*chains too large");
    const long long niter = D / s, m = C * s;
    // layout: x[d + cs * c + ps * p]; 0 = dense
    const long long cs = a->chain_stride ? a->chain_stride : D, ps = a->param_stride ? a->param_stride : n;
    if (cs < D || ps < 1) return fail(ctx, MDT_EARG, "chain_stride must be >= draws and param_stride >= 1 (the draws of a chain are contiguous)");
    const bool strided = !(cs == D && ps == n);
    // a host buffer of some size whose parameters are contiguous pieces (params outermost): the params axis goes to the
    // device in blocks, block i + 1 over PCIe while block i computes (the stage request returns arrays of the size of x:
    // not this way)
    if (a->location == MDT_HOST && rq.mode != 3 && ps >= (C - 1) * cs + D) {
        const char* emin = hook("MDT_HOST_PIPE_MIN"); // bytes (tests: 1)
        const char* eblk = hook("MDT_HOST_PIPE_BLOCKS");
        const size_t xb = (size_t)((P - 1) * ps + (C - 1) * cs + D) * tsize(a->dtype);
        const size_t min_bytes = emin ? (size_t)atoll(emin) : ((size_t)128 << 20);
        long long nb = eblk ? atoll(eblk) : (long long)(xb / ((size_t)192 << 20));
        nb = std::max<long long>(2, std::min<long long>(16, nb));
        if (xb >= min_bytes && P >= 2 * nb && min_bytes > 0) return run_host_blocks(ctx, a, x, rq, (int)nb);
    }
    const int ess_ok = niter > 4;
    long long maxlag = a->maxlag;
    if (rq.want_ess && ess_ok) {
        if (!(maxlag > 0)) return fail(ctx, MDT_EDOMAIN, "maxlag must be >0."); // src/ess_rhat.jl:481
        if (maxlag > niter - 4) maxlag = niter - 4;                              // :482
    } else {
        maxlag = 1;
    }
    if (!(a->tail_prob > 0.0 && a->tail_prob < 1.0) && rq.need_et) return fail(ctx, MDT_EARG, "tail_prob must be in (0, 1)");
    if (rq.need_et && (a->tail_p_lo != 0.0 || a->tail_p_hi != 0.0) && !(a->tail_p_lo > 0.0 && a->tail_p_lo < a->tail_p_hi && a->tail_p_hi < 1.0))
        return fail(ctx, MDT_EARG, "tail_p_lo / tail_p_hi must be both 0 or 0 < tail_p_lo < tail_p_hi < 1");
    if (rq.est == MDT_KIND_QUANTILE && !(a->tail_prob >= 0.0 && a->tail_prob <= 1.0))
        return fail(ctx, MDT_EARG, "quantile probability must be in [0, 1]"); // Statistics.quantile
    // engine limits
    static_assert(mdt_big::MAXM == mdt_small::MAXM && mdt_big::MAXM == mdt_tiny::MAXM, "one limit for all geometries");
    // more split chains than the LDS moment tables hold (many short chains, what nested R-hat is for): every kind runs
    // with one thread per split chain (many_phase), its moments in the scratch of the sorted copy
    const bool many = m > mdt_big::MAXM;
    if (maxlag > mdt_big::MAXLAG_CAP) return fail(ctx, MDT_EUNSUPPORTED, "effective maxlag > 1000 is not supported");
    if (n > 2000000000LL / 2) return fail(ctx, MDT_EUNSUPPORTED, "draws*chains too large");
    const bool indicators = rq.need_et || rq.est == MDT_KIND_MEDIAN || rq.est == MDT_KIND_MAD || rq.est == MDT_KIND_QUANTILE;
    // (split chains beyond the indicator phases' bit vectors, draws / split_chains > 262144: the indicator is evaluated as a
    // proxy series by the tiled series_phase -- SerInd in mdt_device.hpp)
    // (FFTAutocovMethod beyond 8192 padded points: the same linear autocorrelation from lagged products, see the kernel)
    const bool fft = a->autocov == MDT_AUTOCOV_FFT && rq.want_ess && ess_ok;
    // geometry: two 512-thread workgroups per CU (80 KiB of LDS each) whenever the request fits their
    // tables -- phases of two parameters overlap on every CU; one 1024-thread workgroup per CU otherwise
    const bool fft_lds = fft && hook("MDT_FFT_LDS") != nullptr; // run the in-LDS transform instead of lagged products (tests)
    const bool small_fits = maxlag <= mdt_small::MAXLAG_CAP && (!fft_lds || niter + maxlag + 1 <= mdt_small::FFT_MAXN) &&
                            niter <= (long long)mdt_small::BITW * 64 && n <= (long long)mdt_small::QCH8 * (hook("MDT_SMALL_ROUNDS") ? atoll(hook("MDT_SMALL_ROUNDS")) : 64);
    // short slices: four 256-thread workgroups per CU (40 KiB of LDS each) -- a parameter of a few thousand values
    // is mostly barrier and latency chains, which twice as many resident workgroups hide twice as well
    int force_cfg = ctx->force_cfg; // MDT_CONFIG is honoured per request (tests switch it between calls)
    if (const char* c = hook("MDT_CONFIG")) force_cfg = !strcmp(c, "big") ? 1 : (!strcmp(c, "small") ? 2 : (!strcmp(c, "tiny") ? 3 : 0));
    else force_cfg = 0;
    // (measured: 1000 x 4 x 100 000 19.5 ms against 31.8 with two 512-thread workgroups per CU, 200 x 4 x 500 000 57 against
    // 132, 1000 x 12 x 30 000 28.4 against 36.1; no gain once a split chain no longer fits the z-series buffer)
    const bool tiny_fits = small_fits && n <= 12288 && niter + mdt_tiny::SZ_EDGE + 80 <= mdt_tiny::SZ && niter <= (long long)mdt_tiny::BITW * 64 &&
                           (!fft_lds || niter + maxlag + 1 <= mdt_tiny::FFT_MAXN);
    // the one-chunk path (mdt_short.inc): a slice of at most 4000 values, at most eight split chains, no dropped rows, ranked and
    // carried through the bulk series inside ONE chunk of the 512-thread geometry -- which then serves the request instead of
    // the 256-thread one.  Measured in round 4 (bulk+tail, ms, one-chunk / 256-thread general path): 1000 x 4 x 100 000 14.80 /
    // 16.88, 2000 x 2 x 50 000 7.17 / 8.04, 600 x 4 x 100 000 13.72 / 13.76, 500 x 4 x 100 000 13.36 / 12.77: it pays where the
    // 256-thread geometry needs two rounds (more than 2560 values) and loses below, where four small workgroups per CU hide the
    // per-parameter latency chains better than two large ones.  MDT_ONECHUNK=0: never, 2: wherever it can run (tests, A/B timing)
    const int oc_env = hook("MDT_ONECHUNK") ? atoi(hook("MDT_ONECHUNK")) : 1;
    const bool onechunk = oc_env != 0 && small_fits && rq.mode == 0 && !rq.basic &&
                          rq.est == 0 && n <= (long long)mdt_small::OC_MAX && (oc_env == 2 || n > (long long)mdt_tiny::QCH8) && D % s == 0 && m <= 8 &&
                          niter >= 64 && !fft_lds && niter + maxlag + 80 <= mdt_small::SZ && maxlag < mdt_small::SZ_EDGE &&
                          (rq.need_eb || rq.need_rb) && !(hook("MDT_RANK_VARIANT") && atoi(hook("MDT_RANK_VARIANT")) != 2);
    const bool tiny = force_cfg == 0 ? (tiny_fits && !onechunk) : (force_cfg == 3 && tiny_fits);
    const bool small = tiny ? false : (force_cfg == 1 ? false : small_fits); // 80 KiB tables, two workgroups per CU
    // Short slices, one parameter per WAVE (mdt_wave.hip): rank-normalised requests (kinds bulk / tail / rank and the fused one) on
    // slices of 256 ... 8192 values in at most 16 split chains of 8 ... 1024 draws, effective maxlag <= 255, AutocovMethod or
    // FFTAutocovMethod (evaluated from lag products, as everywhere).  Parameters it cannot serve (a NaN, an infinity, a point mass
    // larger than a round, many ties among the folded values) come back in a list and the general kernel, queued behind it, works
    // that list off.  MDT_WAVE=0: never, 2: wherever it can run (tests, A/B timing; needs MDT_TEST_HOOKS=1 like every switch)
    // A wave works on its parameter for 0.1 - 0.4 ms where a workgroup needs 0.04 - 0.08: the wave kernel wins once there are enough
    // parameters to keep its ~3000 waves busy for a few tickets each, and loses below -- measured crossover (fused bulk+tail):
    // 1000 x 4 at ~6000 parameters (P = 10: 0.31 against 0.12 ms), 500 x 4 at ~4000, 250 x 4 at ~2000; above 4096 values the score
    // table leaves room for 8 waves per CU and the crossover moves out (1000 x 8: 16 000 parameters 5.2 against 4.9 ms, 50 000: 13.5
    // against 14.6).  Hence the parameter count in the rule.
    const int wave_env = hook("MDT_WAVE") ? atoi(hook("MDT_WAVE")) : 1;
    // (split chains of more than 640 draws take the longer per-wave LDS block -- 10 waves beside a 4000-value table --: 2000 x 2 ties at
    // 8000 parameters and wins 15 % at 16 000)
    // Round 6: raw draws too -- ess / rhat / ess_rhat(kind = basic) and mcse(kind = mean) (no row dropped by the split: its pooled
    // standard deviation then follows from the split-chain moments) -- through the series phase of the same kernel, no ranking, no
    // per-wave scratch.  R-hat is the same bits as on the workgroup-per-parameter kernels (the same summation order).
    // ... and the std proxy (x - mean)^2: ess(kind = std) and mcse(kind = std) (src/ess_rhat.jl:640-642, src/mcse.jl:52-65)
    const bool wave_std = rq.basic && rq.est == MDT_KIND_STD && (rq.mode == 0 || rq.mode == 4) && ess_ok;
    const bool wave_raw = wave_std || (rq.basic && rq.est == 0 && (rq.mode == 0 || (rq.mode == 2 && D % s == 0)) && (rq.need_eb || rq.need_rb));
    const bool wave_pays = wave_raw ? wave_raw_pays(n, P, (rq.need_eb || wave_std) && ess_ok) : wave_shape_pays(n, niter, P);
    // ... and the indicator estimators whose threshold is an order statistic of x itself: ess(kind = median | quantile(p))
    // (src/ess_rhat.jl:630-633, 647-659) -- the order-statistics-only ranking and the indicator phase of ess(kind = tail)
    const bool wave_est = (rq.est == MDT_KIND_MEDIAN || rq.est == MDT_KIND_MAD || rq.est == MDT_KIND_QUANTILE) && rq.mode == 0 && !rq.basic && ess_ok;
    // ... and the second pass of mcse(kind = median | quantile(p)) (mode 5: two order statistics at the parameter's own places)
    const bool wave_os2 = rq.mode == 5 && !rq.basic && rq.est == 0;
    const bool wave = wave_env != 0 && (wave_pays || wave_env == 2) && force_cfg == 0 && ((rq.mode == 0 && !rq.basic) || wave_raw || wave_os2) && (rq.est == 0 || wave_est || wave_std) &&
                      (rq.need_eb || rq.need_rb || rq.need_et || rq.need_rt || wave_est || wave_std || wave_os2) && wave_shape_fits(n, niter, m) &&
                      maxlag <= mdtw::WV_MAXLAG && !(a->autocov == MDT_AUTOCOV_BDA && rq.want_ess) && !fft_lds &&
                      !(hook("MDT_RANK_VARIANT") && atoi(hook("MDT_RANK_VARIANT")) != 2) && P < 0x7FFFFFFFLL;

    MDT_ON_DEVICE(ctx);
    hipStream_t st = (hipStream_t)a->stream;
    const size_t ts = tsize(a->dtype);
    const bool f32 = a->dtype == MDT_F32;
    // a context serves one stream at a time: work queued on another stream may still use the scratch and the status words
    if (ctx->used && ctx->last_stream != st) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    // nested R-hat: split chains per superchain and superchains the combine step holds in LDS
    if (rq.mode == 1 || rq.mode == 6) {
        const long long mkmax = tiny ? mdt_tiny::NEST_MK : mdt_small::NEST_MK, smax = tiny ? mdt_tiny::NEST_S : mdt_small::NEST_S;
        static_assert(mdt_small::NEST_MK == mdt_big::NEST_MK && mdt_small::NEST_S == mdt_big::NEST_S, "one limit for the 80 / 160 KiB geometries");
        if ((long long)rq.K * s > mkmax || rq.S > smax)
            return fail(ctx, MDT_EUNSUPPORTED, "rhat_nested: more than 2048 split chains per superchain, or more than 2048 superchains (1024 for slices of <= 12288 values)");
    }

    // host staging
    const void* dx = x;
    constexpr int NOUT = 6;
    void* outs_host[NOUT] = {rq.out_eb, rq.out_rb, rq.out_et, rq.out_rt, rq.out_rrank, rq.out_mcse};
    void* outs_dev[NOUT] = {rq.out_eb, rq.out_rb, rq.out_et, rq.out_rt, rq.out_rrank, rq.out_mcse};
    double* rank_host = rq.rank_out; void* z_host = rq.z_out;
    if (a->location == MDT_HOST) {
        const size_t xb = (size_t)((P - 1) * ps + (C - 1) * cs + D) * ts; // the extent of x, whatever lies between its pieces
        if (int rcs = stage_x_bytes(ctx, xb)) return rcs;
        HIP_TRY(hipMemcpyAsync(ctx->stage_x, x, xb, hipMemcpyHostToDevice, st));
        dx = ctx->stage_x;
        size_t ob = NOUT * (size_t)P * ts;
        if (rq.mode == 3) ob += (size_t)n * P * (8 + ts) + 16; // (+ the 16-byte alignment of the stage arrays)
        if (int rcs = stage_out_bytes(ctx, ob)) return rcs;
        for (int i = 0; i < NOUT; ++i)
            outs_dev[i] = outs_host[i] ? (void*)((char*)ctx->stage_out + (size_t)i * P * ts) : nullptr;
        if (rq.mode == 3) {
            char* base = (char*)ctx->stage_out + NOUT * (size_t)P * ts;
            base = (char*)(((uintptr_t)base + 15) & ~(uintptr_t)15);
            if (rank_host) rq.rank_out = (double*)base;
            if (z_host) rq.z_out = base + (size_t)n * P * 8;
        }
    }

    const long long wgs = tiny ? 2LL * ctx->grid : (small ? ctx->grid : ctx->grid / 2);
    int rc = ensure_scratch(ctx, (many && 2 * m > n) ? 2 * m : n, wgs); // (many split chains: their 2 m moments live in the scratch of the sorted copy)
    if (rc) return rc;
    mdtw::Geo wgeo{};
    mdtw::Hooks whk; // (by value through geometry() and launch(): the wave unit keeps no state of its own)
    int wave_wgs = 0;
    if (wave) {
        whk = wave_hooks();
        wgeo = mdtw::geometry(n, niter, f32 ? 1 : 0, whk, wave_raw ? 1 : 0);
        const int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
        wave_wgs = (int)std::min<long long>(ncu, (P + wgeo.waves - 1) / wgeo.waves);
        // (the scratch is sized for a full grid: the same buffer serves every request of the shape; raw draws pass through none)
        const size_t need = wave_raw ? 0 : mdtw::scratch_bytes(wgeo, (long long)ncu * wgeo.waves);
        if (need > ctx->wv_cap) {
            if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
            if (ctx->wv_scratch) (void)hipFree(ctx->wv_scratch);
            ctx->wv_scratch = nullptr; ctx->wv_cap = 0;
            HIP_TRY(hipMalloc(&ctx->wv_scratch, need));
            ctx->wv_cap = need;
        }
        if (P > ctx->fb_cap) {
            if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
            if (ctx->fb_list) (void)hipFree(ctx->fb_list);
            ctx->fb_list = nullptr; ctx->fb_cap = 0;
            HIP_TRY(hipMalloc((void**)&ctx->fb_list, (size_t)P * sizeof(unsigned int)));
            ctx->fb_cap = P;
        }
    }
    if (strided) { if (int rcx = ensure_xcbuf(ctx, wgs)) return rcx; }
    const bool need_rank = !rq.basic || rq.mode == 3;
    if (need_rank) {
        rc = ensure_ztab(ctx, n, f32 ? 1 : 0, st);
        if (rc) return rc;
    }
    if (rq.mode == 1 || rq.mode == 6) {
        const int cnt = rq.K * rq.S;
        if (cnt > ctx->chain_inds_cap) {
            if (ctx->chain_inds) (void)hipFree(ctx->chain_inds);
            ctx->chain_inds = nullptr; ctx->chain_inds_cap = 0;
            HIP_TRY(hipMalloc((void**)&ctx->chain_inds, (size_t)cnt * sizeof(int)));
            ctx->chain_inds_cap = cnt;
        }
        HIP_TRY(hipMemcpyAsync(ctx->chain_inds, rq.chain_inds, (size_t)cnt * sizeof(int), hipMemcpyHostToDevice, st));
    }
    // (status word 0: flags, word 1: the kernel's parameter ticket -- both cleared by kp_setup_kernel below)

    KP kp;
    memset(&kp, 0, sizeof(kp));
    kp.x = dx;
    kp.cstride = cs; kp.pstride = ps; kp.strided = strided ? 1 : 0; kp.xcbuf = ctx->xcbuf;
    kp.ibits = ctx->ibits; kp.ibits_stride = hook("MDT_NO_IBITS") ? 0 : ctx->ibits_stride; // (MDT_NO_IBITS: indicators from x itself; tests, A/B timing)
    kp.D = D; kp.C = C; kp.P = P; kp.n = n;
    kp.niter = niter; kp.s = (int)s; kp.rem = (int)(D % s); kp.m = (int)m;
    kp.mode = rq.mode; kp.basic = rq.basic; kp.est = rq.est; kp.fft_lds = fft_lds ? 1 : 0;
    kp.need_eb = rq.need_eb; kp.need_rb = rq.need_rb; kp.need_et = rq.need_et; kp.need_rt = rq.need_rt;
    kp.method = a->autocov; kp.ess_ok = ess_ok; kp.relative = a->relative ? 1 : 0; kp.fold = rq.fold;
    kp.maxlag = maxlag;
    const long long ntotal = niter * m;
    // rel_ess_max = log10(oftype(one(T), ntotal)) evaluated in T (src/ess_rhat.jl:514)
    kp.rel_ess_max = f32 ? (double)log10f((float)ntotal) : log10((double)ntotal);
    kp.ntotal = (double)ntotal;
    kp.cf = niter > 0 ? (double)(niter - 1) / (double)niter : -INFINITY; // (niter - 1)//niter
    // type-7 quantile index/gamma in T (Statistics.quantile: aleph = fma(n, p, 1 - p)), :305-309
    for (int i = 0; i < 2; ++i) {
        long long j; double g;
        if (f32) {
            const bool given = a->tail_p_lo != 0.0 && a->tail_p_hi != 0.0 && rq.est != MDT_KIND_QUANTILE; // (the caller's own pl, pu)
            const float p = (rq.est == MDT_KIND_QUANTILE) ? (float)a->tail_prob
                            : given ? (float)((i == 0) ? a->tail_p_lo : a->tail_p_hi) : ((i == 0) ? (float)(a->tail_prob / 2) : (float)(1 - a->tail_prob / 2));
            const float aleph = fmaf((float)n, p, 1.0f - p);
            j = (long long)aleph;
            if (j < 1) j = 1;
            if (j > n - 1) j = n - 1;
            float gg = aleph - (float)j;
            gg = gg < 0 ? 0 : (gg > 1 ? 1 : gg);
            g = (double)gg;
        } else {
            const bool given = a->tail_p_lo != 0.0 && a->tail_p_hi != 0.0 && rq.est != MDT_KIND_QUANTILE;
            const double p = (rq.est == MDT_KIND_QUANTILE) ? a->tail_prob
                             : given ? ((i == 0) ? a->tail_p_lo : a->tail_p_hi) : ((i == 0) ? (a->tail_prob / 2) : (1 - a->tail_prob / 2));
            const double aleph = fma((double)n, p, 1.0 - p);
            j = (long long)aleph;
            if (j < 1) j = 1;
            if (j > n - 1) j = n - 1;
            g = aleph - (double)j;
            g = g < 0 ? 0 : (g > 1 ? 1 : g);
        }
        if (n == 1) j = 1;
        kp.q_j[i] = j;
        kp.q_g[i] = g;
    }
    kp.ztab = ctx->ztab;
    kp.r2buf = ctx->r2buf; kp.zbuf = ctx->zbuf; kp.binbuf = ctx->binbuf; kp.n_pad = ctx->n_pad;
    kp.xsbuf = ctx->xsbuf; kp.bsbuf = ctx->bsbuf;
    {
        const char* rv = hook("MDT_RANK_VARIANT");
        // rounds re-stream the slice for small slices (<= 16-bit slots, few rounds); beyond, the two-level
        // partitioned variant (O(n) traffic, bins re-cut per round; it falls back to the single-level
        // partitioned variant 1 on a huge point mass); MDT_RANK_VARIANT forces one (tests, A/B timing)
        // (two 512-thread workgroups per CU: measured crossover at draws*chains ~ 12-16 K -- below it the
        // histogram + partition passes of the two-level variant cost more than re-streaming a slice that is
        // only one or two rounds long: 1000 x 4 x 100 000 takes 31.8 ms instead of 35.0, 200 x 4 x 500 000 132 instead of 161)
        // (round 3: since the folded ranking is a merge of the sorted copy the two-level variant leaves, it wins at EVERY slice
        // size of the 512- and 256-thread geometries -- 1000 x 4 x 100 000 16.4 ms against 19.3, 2000 x 4 x 50 000 12.9 against
        // 18.1, 1000 x 8 x 50 000 15.1 against 20.6, 50 x 4 x 400 000 38.3 against 39.1 -- and it is the default everywhere: one
        // variant in every geometry also keeps R-hat the same BITS whichever geometry serves a slice (the merge and the general
        // folded ranking take their moments around different shifts); the other two remain as its fallback (a coarse class
        // larger than the chunk) and behind MDT_RANK_VARIANT)
        // (a bulk-only request on a short slice would still be 20 % faster with re-streaming rounds -- 1000 x 4 x 100 000,
        // ess_rhat(kind = bulk): 9.0 ms against 10.9 -- but its ESS would then differ in the last bits from the bulk half of
        // the fused request, which takes the segment-order series: one variant, one set of bits)
        kp.rank_variant = rv ? atoi(rv) : 2;
        // the round tables hold MAXROUNDS rounds of at most one LDS chunk each: beyond that the ranking cannot be served
        // (slices between half of that and that may still overflow on the device: status flag, NaN outputs)
        const long long chunk = tiny ? std::min(mdt_tiny::QCH8, mdt_tiny::CH) : (small ? std::min(mdt_small::QCH8, mdt_small::CH) : std::min(mdt_big::QCH8, mdt_big::CH));
        if (need_rank && n > (long long)mdt_big::MAXROUNDS * chunk)
            return fail(ctx, MDT_EUNSUPPORTED, "draws*chains too large for the on-chip ranking (round table overflow)");
    }
    { const char* fm = hook("MDT_FOLD_MERGE"); kp.fold_merge = fm ? atoi(fm) : 1; }
    { const char* wo = hook("MDT_WINOUT"); kp.winout = wo ? atoi(wo) : 1; }
    { const char* pe = hook("MDT_PA"); kp.pa_on = pe ? atoi(pe) : 1; }
    { const char* te = hook("MDT_TAIL_SRT"); kp.tail_srt = te ? atoi(te) : 1; }
    { const char* qe = hook("MDT_SQ_FAST"); kp.sq_fast = qe ? atoi(qe) : 1; }
    { const char* fe = hook("MDT_F4_STAG"); kp.f4_stag = fe ? atoi(fe) : 3; } // bit 0: slots of the sorted copy, bit 1: moments in sorted order
    kp.onechunk = onechunk ? 1 : 0;
    kp.out_eb = outs_dev[0]; kp.out_rb = outs_dev[1]; kp.out_et = outs_dev[2]; kp.out_rt = outs_dev[3];
    kp.out_rrank = outs_dev[4];
    kp.out_mcse = outs_dev[5];
    kp.rank_out = rq.rank_out; kp.z_out = rq.z_out;
    kp.chain_inds = ctx->chain_inds; kp.K = rq.K; kp.S = rq.S;
    kp.tpos_arr = rq.tpos_arr;
    kp.status = ctx->status;
    kp.prof = ctx->prof;
    // ticket order: by the durations the previous launch with as many parameters measured; this launch measures again
    kp.order = nullptr; kp.cost = nullptr;
    const bool lpt = !wave && lpt_shape(P, wgs) && !hook("MDT_NO_ORDER"); // (one ticket per workgroup at most: nothing to order)
    if (lpt) {
        if (P > ctx->order_cap) {
            if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
            if (ctx->cost) (void)hipFree(ctx->cost);
            if (ctx->order) (void)hipFree(ctx->order);
            if (ctx->order_sorted) (void)hipFree(ctx->order_sorted);
            ctx->cost = nullptr; ctx->order = nullptr; ctx->order_sorted = nullptr; ctx->order_cap = 0; ctx->order_P = -1;
            HIP_TRY(hipMalloc((void**)&ctx->cost, (size_t)P * sizeof(unsigned int)));
            HIP_TRY(hipMalloc((void**)&ctx->order, (size_t)P * sizeof(unsigned int)));
            HIP_TRY(hipMalloc((void**)&ctx->order_sorted, (size_t)P * sizeof(unsigned int)));
            ctx->order_cap = P;
        }
        kp.cost = ctx->cost;
        if (ctx->order_P == P) kp.order = ctx->order;
    }

    const unsigned grid = (unsigned)((P < wgs) ? P : wgs);
    void (*kern)(const KP*) = tiny ? (f32 ? mdt_tiny::mdt_kernel<float> : mdt_tiny::mdt_kernel<double>)
                              : small ? (f32 ? mdt_small::mdt_kernel<float> : mdt_small::mdt_kernel<double>)
                                      : (f32 ? mdt_big::mdt_kernel<float> : mdt_big::mdt_kernel<double>);
    const int nthreads = tiny ? mdt_tiny::NT : (small ? mdt_small::NT : mdt_big::NT);
    const int ldsb = tiny ? mdt_tiny::LDS_BYTES : (small ? mdt_small::LDS_BYTES : mdt_big::LDS_BYTES);
    // the parameter block travels through a small ring in device memory: the kernel reads it with
    // scalar loads on demand instead of pinning ~60 SGPRs for its whole life
    const unsigned si = ctx->kp_next++ % KP_RING;
    KP* slot = ctx->kp_ring + si;
    if (wave) {
        // the wave-per-parameter kernel over all parameters, then the general kernel over the list it left (its length is only
        // known on the device: the second slot's parameter count is patched there); both on the stream, nothing returns to the host
        hipLaunchKernelGGL(kp_setup_kernel, dim3(1), dim3(64), 0, st, slot, kp, ctx->status, rq.keep_status & 1);
        mdtw::launch(st, (const KP*)slot, f32 ? 1 : 0, wgeo, wave_wgs, ctx->wv_scratch, ctx->fb_list, ctx->status + 3, whk);
        KP kb = kp;
        kb.order = ctx->fb_list; kb.cost = nullptr; kb.P = 0;
        KP* slot2 = ctx->kp_ring + (ctx->kp_next++ % KP_RING);
        hipLaunchKernelGGL(kp_setup_kernel, dim3(1), dim3(64), 0, st, slot2, kb, ctx->status, 3 | 4);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), ldsb, st, (const KP*)slot2);
    } else {
        hipLaunchKernelGGL(kp_setup_kernel, dim3(1), dim3(64), 0, st, slot, kp, ctx->status, rq.keep_status & 1);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), ldsb, st, (const KP*)slot);
    }
    if (lpt) { // (same stream: it runs when every duration is in, and before the next launch reads the order)
        // the cheapest 2 x wgs parameters close the launch in sorted order; the others are walked with a stride near head / 7.3
        // (a handful of tickets per workgroup -- BASELINE config 3 has 3.9 --: the makespan is a bin-packing problem, the whole
        // sequence is sorted, longest first)
        long long tail = P <= 6 * wgs ? P : std::min<long long>(P, 2 * wgs), head = P - tail, stride = 1;
        if (const char* e = hook("MDT_ORDER_TAIL")) { tail = std::min<long long>(P, std::max<long long>(0, atoll(e))); head = P - tail; } // (A/B timing)
        if (head > 2 && !hook("MDT_ORDER_PURE")) {
            stride = std::max<long long>(2, (long long)((double)head / 7.3));
            auto gcd = [](long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; };
            while (gcd(stride, head) != 1) ++stride;
        }
        hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, st, (const unsigned int*)ctx->cost, ctx->order_sorted, ctx->order, P, tail, stride);
        ctx->order_P = P;
    }
    HIP_TRY(hipGetLastError());
    ctx->last_stream = st;
    ctx->used = true;

    if (a->location == MDT_HOST) {
        for (int i = 0; i < NOUT; ++i)
            if (outs_host[i]) HIP_TRY(hipMemcpyAsync(outs_host[i], outs_dev[i], (size_t)P * ts, hipMemcpyDeviceToHost, st));
        if (rq.mode == 3) {
            if (rank_host) HIP_TRY(hipMemcpyAsync(rank_host, rq.rank_out, (size_t)n * P * 8, hipMemcpyDeviceToHost, st));
            if (z_host) HIP_TRY(hipMemcpyAsync(z_host, rq.z_out, (size_t)n * P * ts, hipMemcpyDeviceToHost, st));
        }
    }
    if (a->sync || a->location == MDT_HOST) {
        int status = 0;
        HIP_TRY(hipMemcpyAsync(&status, ctx->status, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (status & 3) { // reported here: not left behind as a sticky flag
            HIP_TRY(hipMemsetAsync(ctx->status, 0, sizeof(int), st));
            if (status & 1) return fail(ctx, MDT_EUNSUPPORTED, "draws*chains too large for the on-chip ranking (round table overflow)");
            return fail(ctx, MDT_EARG, "quantiles are undefined for collections containing NaNs"); // Statistics.quantile
        }
    }
    return MDT_OK;
}

// ---- a host buffer in blocks of the params axis: the copy of block i + 1 (own stream) overlaps the launch of block i.
// Every block is an ordinary device-resident request on the staged copy (same strides, outputs into the staging area);
// the status flags accumulate over the blocks and are reported once, like the plain path does.
int run_host_blocks(mdt_ctx* ctx, const mdt_args* a, const void* x, const Req& rq, int nb) {
    MDT_ON_DEVICE(ctx);
    hipStream_t st = (hipStream_t)a->stream;
    const size_t ts = tsize(a->dtype);
    const long long D = a->draws, C = a->chains, P = a->params;
    const long long cs = a->chain_stride ? a->chain_stride : D, ps = a->param_stride ? a->param_stride : D * C;
    if (ctx->used && ctx->last_stream != st) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    const size_t xb = (size_t)((P - 1) * ps + (C - 1) * cs + D) * ts;
    if (int rcs = stage_x_bytes(ctx, xb)) return rcs;
    constexpr int NOUT = 6;
    if (int rcs = stage_out_bytes(ctx, NOUT * (size_t)P * ts)) return rcs;
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    void* outs_host[NOUT] = {rq.out_eb, rq.out_rb, rq.out_et, rq.out_rt, rq.out_rrank, rq.out_mcse};
    bool first = true;
    for (int b = 0; b < nb; ++b) {
        long long lo, hi;
        shard_bounds(P, b, nb, &lo, &hi);
        const long long cnt = hi - lo;
        if (cnt == 0) continue;
        if (!ctx->copy_ev[b]) HIP_TRY(hipEventCreateWithFlags(&ctx->copy_ev[b], hipEventDisableTiming));
        const size_t off = (size_t)lo * (size_t)ps * ts, bytes = (size_t)((cnt - 1) * ps + (C - 1) * cs + D) * ts;
        HIP_TRY(hipMemcpyAsync((char*)ctx->stage_x + off, (const char*)x + off, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
        HIP_TRY(hipEventRecord(ctx->copy_ev[b], ctx->copy_stream));
        HIP_TRY(hipStreamWaitEvent(st, ctx->copy_ev[b], 0));
        mdt_args ab = *a;
        ab.location = MDT_DEVICE;
        ab.params = cnt;
        ab.sync = 0;
        Req rb = rq;
        void** outs_b[NOUT] = {&rb.out_eb, &rb.out_rb, &rb.out_et, &rb.out_rt, &rb.out_rrank, &rb.out_mcse};
        for (int i = 0; i < NOUT; ++i)
            *outs_b[i] = outs_host[i] ? (void*)((char*)ctx->stage_out + ((size_t)i * P + (size_t)lo) * ts) : nullptr;
        if (rq.tpos_arr) rb.tpos_arr = rq.tpos_arr + 2 * lo;
        rb.keep_status = first ? rq.keep_status : 1;
        first = false;
        if (int rc = run(ctx, &ab, (const char*)ctx->stage_x + off, rb)) {
            // (the checks of the request run inside run(): on a refusal the copy of this block from the caller's buffer is
            // still in flight and earlier blocks are queued -- the caller may free x as soon as we return)
            (void)hipStreamSynchronize(ctx->copy_stream);
            (void)hipStreamSynchronize(st);
            return rc;
        }
    }
    for (int i = 0; i < NOUT; ++i)
        if (outs_host[i]) HIP_TRY(hipMemcpyAsync(outs_host[i], (char*)ctx->stage_out + (size_t)i * P * ts, (size_t)P * ts, hipMemcpyDeviceToHost, st));
    int status = 0;
    HIP_TRY(hipMemcpyAsync(&status, ctx->status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (status & 3) { // reported here: not left behind as a sticky flag
        HIP_TRY(hipMemsetAsync(ctx->status, 0, sizeof(int), st));
        if (status & 1) return fail(ctx, MDT_EUNSUPPORTED, "draws*chains too large for the on-chip ranking (round table overflow)");
        return fail(ctx, MDT_EARG, "quantiles are undefined for collections containing NaNs"); // Statistics.quantile
    }
    return MDT_OK;
}

// ---- several GPUs: the params axis in contiguous blocks, one worker thread and one child context per device.
// The partition rule is ONE rule everywhere (mdt_shard_bounds, dist.shard_bounds of the Python mirror, the Julia shim): the
// first P mod N blocks hold one parameter more.
void shard_bounds(long long P, int i, int n, long long* lo, long long* hi) {
    const long long q = P / n, r = P % n;
    *lo = (long long)i * q + std::min<long long>(i, r);
    *hi = *lo + q + (i < r ? 1 : 0);
}

// One request over the children of a multi-GPU context.  fn(child, args of the block, x of the block, outs of the block)
// runs on the child's worker thread; this helper finds every block's input -- already on the child's GPU (pre-sharded:
// args->x_shards, or the block of a single array that lives on that GPU), in host memory, or on another GPU (fetched by
// one peer copy over xGMI) -- and brings the block's outputs back to the caller's vectors (written in place when they live
// on the child's GPU or in host memory, one peer copy per output otherwise).
constexpr int MULTI_NOUT = 6;
using ShardFn = std::function<int(mdt_ctx*, const mdt_args*, const void*, void* const*)>;
int for_shards(mdt_ctx* ctx, const mdt_args* a, const void* x, void* const* outs, int nout, const ShardFn& fn) {
    if (a->dtype != MDT_F64 && a->dtype != MDT_F32) return fail(ctx, MDT_EARG, "dtype must be MDT_F64 or MDT_F32");
    if (a->location != MDT_HOST && a->location != MDT_DEVICE) return fail(ctx, MDT_EARG, "bad location");
    if (a->draws < 0 || a->chains < 0 || a->params < 0) return fail(ctx, MDT_EDIM, "negative dimension");
    const long long P = a->params, D = a->draws, C = a->chains;
    if (P == 0) return MDT_OK;
    if (!x && !a->x_shards) return fail(ctx, MDT_EARG, "x is NULL");
    if (D == 0 || C == 0) return fail(ctx, MDT_EDIM, "empty sample dimensions");
    const long long cs = a->chain_stride ? a->chain_stride : D, ps = a->param_stride ? a->param_stride : D * C;
    if (cs < D || ps < 1) return fail(ctx, MDT_EARG, "chain_stride must be >= draws and param_stride >= 1 (the draws of a chain are contiguous)");
    const int nd = (int)ctx->children.size();
    const size_t ts = tsize(a->dtype);
    // where the single array / the gathered outputs live (device memory: found from the pointers)
    int x_dev = -1, out_dev = -1;
    if (a->location == MDT_DEVICE) {
        hipPointerAttribute_t at;
        if (!a->x_shards) {
            if (hipPointerGetAttributes(&at, x) != hipSuccess) return fail(ctx, MDT_EARG, "x is not a device pointer");
            x_dev = at.device;
        }
        for (int k = 0; k < nout; ++k)
            if (outs[k]) {
                if (hipPointerGetAttributes(&at, outs[k]) != hipSuccess) return fail(ctx, MDT_EARG, "an output is not a device pointer");
                out_dev = at.device;
                break;
            }
        // The producers of the input have finished before any child reads it.  A single x: args->stream is a stream of the
        // device that HOLDS x (a NULL stream is the default stream of whatever device is current, so that device is made
        // current first).  Pre-sharded input has one producer per GPU and only one stream slot in the args: every worker
        // synchronises its own device before it touches its block (below).
        if (!a->x_shards) {
            DeviceGuard gx(x_dev);
            if (!gx.ok || hipStreamSynchronize((hipStream_t)a->stream) != hipSuccess) return fail(ctx, MDT_ERUNTIME, "cannot synchronise the stream of x");
        }
    }
    auto job = [&](int i) -> int {
        mdt_ctx* child = ctx->children[i];
        long long lo, hi;
        shard_bounds(P, i, nd, &lo, &hi);
        const long long cnt = hi - lo;
        if (cnt == 0) return MDT_OK;
        mdt_args b = *a;
        b.params = cnt;
        b.sync = 1;
        b.stream = nullptr; // (every child works on its own device's default stream)
        b.x_shards = nullptr;
        mdt_ctx* ctx = child; // (HIP_TRY reports into the child)
        MDT_ON_DEVICE(child);
        const void* xs;
        if (a->x_shards) {
            xs = a->x_shards[i];
            if (!xs) return fail(child, MDT_EARG, "x_shards holds a NULL block");
            if (a->location == MDT_DEVICE) {
                hipPointerAttribute_t at;
                if (hipPointerGetAttributes(&at, xs) != hipSuccess || at.device != child->device)
                    return fail(child, MDT_EARG, "x_shards[i] must live on device_ids[i] of the context");
                // whatever stream of this GPU produced the block (non-blocking ones included) is done before it is read
                HIP_TRY(hipDeviceSynchronize());
            }
        } else {
            xs = (const char*)x + (size_t)lo * (size_t)ps * ts;
            if (a->location == MDT_DEVICE && x_dev != child->device) {
                // the block lives on another GPU: fetch its extent (whatever lies between its pieces) by one peer copy
                const size_t xb = (size_t)((cnt - 1) * ps + (C - 1) * cs + D) * ts;
                if (int rc = stage_x_bytes(child, xb)) return rc;
                HIP_TRY(hipMemcpyPeer(child->stage_x, child->device, xs, x_dev, xb));
                xs = child->stage_x;
            }
        }
        void* lo_outs[MULTI_NOUT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        const bool out_local = a->location == MDT_HOST || out_dev == child->device || out_dev < 0;
        if (!out_local) { if (int rc = stage_out_bytes(child, (size_t)nout * cnt * ts + (size_t)MULTI_NOUT * 16)) return rc; }
        for (int k = 0; k < nout; ++k)
            if (outs[k]) lo_outs[k] = out_local ? (void*)((char*)outs[k] + (size_t)lo * ts)
                                                 : (void*)((char*)child->stage_out_multi(k, cnt, ts));
        if (int rc = fn(child, &b, xs, lo_outs)) return rc;
        if (!out_local)
            for (int k = 0; k < nout; ++k)
                if (outs[k]) HIP_TRY(hipMemcpyPeer((char*)outs[k] + (size_t)lo * ts, out_dev, lo_outs[k], child->device, (size_t)cnt * ts));
        return MDT_OK;
    };
    ctx->pool->run(job);
    for (int i = 0; i < nd; ++i)
        if (ctx->pool->rcs[i]) return fail(ctx, ctx->pool->rcs[i], "device " + std::to_string(ctx->children[i]->device) + ": " + ctx->children[i]->err);
    return MDT_OK;
}

int run_multi(mdt_ctx* ctx, const mdt_args* a, const void* x, const Req& rq) {
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    if ((rq.mode > 2 && rq.mode != 4 && rq.mode != 6) || rq.tpos_arr) return fail(ctx, MDT_EUNSUPPORTED, "this entry point does not run on a multi-GPU context");
    void* Req::*slots[MULTI_NOUT] = {&Req::out_eb, &Req::out_rb, &Req::out_et, &Req::out_rt, &Req::out_rrank, &Req::out_mcse};
    void* outs[MULTI_NOUT];
    for (int k = 0; k < MULTI_NOUT; ++k) outs[k] = rq.*slots[k];
    return for_shards(ctx, a, x, outs, MULTI_NOUT, [&](mdt_ctx* child, const mdt_args* b, const void* xs, void* const* lo_outs) {
        Req r = rq;
        for (int k = 0; k < MULTI_NOUT; ++k) r.*slots[k] = lo_outs[k];
        return run(child, b, xs, r);
    });
}

} // namespace

extern "C" {

int mdt_version(void) { return MDT_VERSION; }

int mdt_ctx_create(mdt_ctx** out, int device_id) {
    if (!out) return MDT_EARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MDT_ERUNTIME;
    if (device_id < 0 || device_id >= ndev) return MDT_EARG;
    mdt_ctx* ctx = new (std::nothrow) mdt_ctx();
    if (!ctx) return MDT_ENOMEM;
    ctx->device = device_id;
    DeviceGuard guard(device_id);
    if (!guard.ok) { delete ctx; return MDT_ERUNTIME; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete ctx; return MDT_ERUNTIME; }
    ctx->num_cu = prop.multiProcessorCount;
    ctx->grid = 2 * (ctx->num_cu > 0 ? ctx->num_cu : 256); // two 512-thread or one 1024-thread workgroup per CU (LDS-bound)
    if (const char* g = hook("MDT_GRID")) { const int v = atoi(g); if (v > 0) ctx->grid = 2 * v; } // diagnostics: workgroups of the big geometry
    if (const char* c = hook("MDT_CONFIG")) ctx->force_cfg = !strcmp(c, "big") ? 1 : (!strcmp(c, "small") ? 2 : (!strcmp(c, "tiny") ? 3 : 0));
    if (hipMalloc((void**)&ctx->status, 4 * sizeof(int)) != hipSuccess || hipMemset(ctx->status, 0, 4 * sizeof(int)) != hipSuccess) { delete ctx; return MDT_ENOMEM; }
    // > 64 KB of dynamic LDS needs the opt-in, per kernel variant
    const void* kerns[] = {(const void*)mdt_big::mdt_kernel<double>, (const void*)mdt_big::mdt_kernel<float>,
                           (const void*)mdt_small::mdt_kernel<double>, (const void*)mdt_small::mdt_kernel<float>,
                           (const void*)mdt_tiny::mdt_kernel<double>, (const void*)mdt_tiny::mdt_kernel<float>};
    const int klds[] = {mdt_big::LDS_BYTES, mdt_big::LDS_BYTES, mdt_small::LDS_BYTES, mdt_small::LDS_BYTES, mdt_tiny::LDS_BYTES, mdt_tiny::LDS_BYTES};
    for (int ki = 0; ki < 6; ++ki) {
        const void* k = kerns[ki];
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, klds[ki]) != hipSuccess) {
            (void)hipFree(ctx->status);
            delete ctx;
            return MDT_ERUNTIME;
        }
    }
    if (mdtw::kp_bytes() != sizeof(KP) || mdtw::prepare_device() != 0) { // (a variant build that linked a stale partner unit: another KP layout)
        (void)hipFree(ctx->status);
        delete ctx;
        return MDT_ERUNTIME;
    }
    if (hipMalloc((void**)&ctx->kp_ring, KP_RING * sizeof(KP)) != hipSuccess) {
        (void)hipFree(ctx->status);
        delete ctx;
        return MDT_ENOMEM;
    }
    if (hook("MDT_PROF")) {
        if (hipMalloc((void**)&ctx->prof, (size_t)ctx->grid * 2 * 16 * sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(ctx->prof, 0, (size_t)ctx->grid * 2 * 16 * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipFree(ctx->status);
            delete ctx;
            return MDT_ENOMEM;
        }
    }
    *out = ctx;
    return MDT_OK;
}

int mdt_ctx_reserve(mdt_ctx* ctx, const mdt_args* a) {
    if (!ctx) return MDT_EARG;
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    if (!ctx->children.empty()) { // every block on its own GPU, sized for its share of the params axis
        for (size_t i = 0; i < ctx->children.size(); ++i) {
            long long lo, hi;
            shard_bounds(a->params, (int)i, (int)ctx->children.size(), &lo, &hi);
            mdt_args b = *a;
            b.params = hi - lo;
            b.stream = nullptr;
            const int rc = mdt_ctx_reserve(ctx->children[i], &b);
            if (rc) return fail(ctx, rc, ctx->children[i]->err);
        }
        return MDT_OK;
    }
    if (a->dtype != MDT_F64 && a->dtype != MDT_F32) return fail(ctx, MDT_EARG, "dtype must be MDT_F64 or MDT_F32");
    if (a->draws <= 0 || a->chains <= 0 || a->params < 0) return fail(ctx, MDT_EDIM, "empty sample dimensions");
    if (a->location != MDT_HOST && a->location != MDT_DEVICE) return fail(ctx, MDT_EARG, "bad location");
    long long n = 0;
    if (!slice_size(a->draws, a->chains, &n)) return fail(ctx, MDT_EUNSUPPORTED, "draws*chains too large"); // (checked before the product is formed)
    const long long D = a->draws, P = a->params;
    const long long s = a->split_chains >= 1 ? a->split_chains : 2, niter = D / s, m = a->chains * s; // (0: the keyword's default)
    const long long cs = a->chain_stride ? a->chain_stride : D, ps = a->param_stride ? a->param_stride : n;
    if (cs < D || ps < 1) return fail(ctx, MDT_EARG, "chain_stride must be >= draws and param_stride >= 1 (the draws of a chain are contiguous)");
    const bool strided = cs != D || ps != n;
    const size_t ts = tsize(a->dtype);
    MDT_ON_DEVICE(ctx);
    hipStream_t st = (hipStream_t)a->stream;
    if (ctx->used && ctx->last_stream != st) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    // the largest grid a request of this slice size can take (four 256-thread workgroups per CU serve slices of <= 12288 values)
    const long long wgs = n <= 12288 ? 2LL * ctx->grid : ctx->grid;
    if (int rc = ensure_scratch(ctx, n, wgs)) return rc;
    if (int rc = ensure_ztab(ctx, n, a->dtype == MDT_F32 ? 1 : 0, st)) return rc;
    if (strided) { if (int rc = ensure_xcbuf(ctx, wgs)) return rc; } // (MCMCChains storage: the per-workgroup dense copy of a parameter)
    // the same shape predicates as run(): the wave-per-parameter kernel's scratch and list only when such a request would be handed
    // to it, the ticket tables only when a workgroup gets more than one ticket
    const bool wave = wave_shape_fits(n, niter, m) && wave_shape_pays(n, niter, P);
    if (wave) {
        const mdtw::Hooks whk = wave_hooks();
        const mdtw::Geo g = mdtw::geometry(n, niter, a->dtype == MDT_F32 ? 1 : 0, whk);
        const int ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
        const size_t need = mdtw::scratch_bytes(g, (long long)ncu * g.waves);
        if (need > ctx->wv_cap) {
            if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
            if (ctx->wv_scratch) (void)hipFree(ctx->wv_scratch);
            ctx->wv_scratch = nullptr; ctx->wv_cap = 0;
            HIP_TRY(hipMalloc(&ctx->wv_scratch, need));
            ctx->wv_cap = need;
        }
        if (P > ctx->fb_cap) {
            if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
            if (ctx->fb_list) (void)hipFree(ctx->fb_list);
            ctx->fb_list = nullptr; ctx->fb_cap = 0;
            HIP_TRY(hipMalloc((void**)&ctx->fb_list, (size_t)std::max<long long>(P, 1) * sizeof(unsigned int)));
            ctx->fb_cap = P;
        }
    }
    // (a request the wave kernel takes draws no ordered tickets; the grid of the request is wgs or half of it: the smaller one decides)
    if (!wave && lpt_shape(P, std::min<long long>(wgs, ctx->grid / 2)) && P > ctx->order_cap) {
        if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
        if (ctx->cost) (void)hipFree(ctx->cost);
        if (ctx->order) (void)hipFree(ctx->order);
        if (ctx->order_sorted) (void)hipFree(ctx->order_sorted);
        ctx->cost = nullptr; ctx->order = nullptr; ctx->order_sorted = nullptr; ctx->order_cap = 0; ctx->order_P = -1;
        HIP_TRY(hipMalloc((void**)&ctx->cost, (size_t)P * sizeof(unsigned int)));
        HIP_TRY(hipMalloc((void**)&ctx->order, (size_t)P * sizeof(unsigned int)));
        HIP_TRY(hipMalloc((void**)&ctx->order_sorted, (size_t)P * sizeof(unsigned int)));
        ctx->order_cap = P;
    }
    // host arrays below the block pipeline's threshold are staged whole: input and the output vectors (larger ones travel in
    // blocks of the params axis through buffers of their own -- run_host_blocks -- beside a PCIe copy that dwarfs their allocation)
    if (a->location == MDT_HOST && P > 0) {
        const size_t xb = (size_t)((P - 1) * ps + (a->chains - 1) * cs + D) * ts;
        if (xb < ((size_t)128 << 20)) {
            if (int rc = stage_x_bytes(ctx, xb)) return rc;
            if (int rc = stage_out_bytes(ctx, 6 * (size_t)P * ts)) return rc;
        }
    }
    ctx->last_stream = st;
    ctx->used = true;
    HIP_TRY(hipStreamSynchronize(st)); // the score table is built when this returns
    return MDT_OK;
}

int mdt_ctx_destroy(mdt_ctx* ctx) {
    if (!ctx) return MDT_OK;
    if (!ctx->children.empty()) {
        delete ctx->pool; // (joins the worker threads)
        for (mdt_ctx* c : ctx->children) (void)mdt_ctx_destroy(c);
        delete ctx;
        return MDT_OK;
    }
    DeviceGuard guard(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->scratch_slab) (void)hipFree(ctx->scratch_slab);
    if (ctx->ztab) (void)hipFree(ctx->ztab);
    if (ctx->status) (void)hipFree(ctx->status);
    if (ctx->prof) (void)hipFree(ctx->prof);
    if (ctx->kp_ring) (void)hipFree(ctx->kp_ring);
    if (ctx->wv_scratch) (void)hipFree(ctx->wv_scratch);
    if (ctx->fb_list) (void)hipFree(ctx->fb_list);
    if (ctx->chain_inds) (void)hipFree(ctx->chain_inds);
    if (ctx->ess_tmp) (void)hipFree(ctx->ess_tmp);
    if (ctx->tpos_arr) (void)hipFree(ctx->tpos_arr);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : ctx->copy_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->stage_x) (void)hipFree(ctx->stage_x);
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    if (ctx->mom) (void)hipFree(ctx->mom);
    if (ctx->xcbuf) (void)hipFree(ctx->xcbuf);
    if (ctx->ibits) (void)hipFree(ctx->ibits);
    if (ctx->cost) (void)hipFree(ctx->cost);
    if (ctx->order) (void)hipFree(ctx->order);
    if (ctx->order_sorted) (void)hipFree(ctx->order_sorted);
    delete ctx;
    return MDT_OK;
}

const char* mdt_last_error(const mdt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mdt_ctx_create_multi(mdt_ctx** out, const int* device_ids, int ndev) {
    if (!out) return MDT_EARG;
    *out = nullptr;
    if (!device_ids || ndev < 1) return MDT_EARG;
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j]) return MDT_EARG; // one block per GPU
    mdt_ctx* ctx = new (std::nothrow) mdt_ctx();
    if (!ctx) return MDT_ENOMEM;
    ctx->device = device_ids[0];
    for (int i = 0; i < ndev; ++i) {
        mdt_ctx* c = nullptr;
        const int rc = mdt_ctx_create(&c, device_ids[i]);
        if (rc != MDT_OK) {
            for (mdt_ctx* d : ctx->children) (void)mdt_ctx_destroy(d);
            delete ctx;
            return rc;
        }
        ctx->children.push_back(c);
    }
    // peer access for blocks that live on another GPU of the node (ignored where it is already on or unavailable:
    // hipMemcpyPeer works either way)
    for (int i = 0; i < ndev; ++i) {
        DeviceGuard g(device_ids[i]);
        for (int j = 0; j < ndev; ++j) {
            int can = 0;
            if (i != j && hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(device_ids[j], 0);
        }
        (void)hipGetLastError();
    }
    ctx->pool = new (std::nothrow) MultiPool(ndev);
    if (!ctx->pool) { for (mdt_ctx* d : ctx->children) (void)mdt_ctx_destroy(d); delete ctx; return MDT_ENOMEM; }
    *out = ctx;
    return MDT_OK;
}

int mdt_shard_bounds(int64_t params, int32_t shard, int32_t nshards, int64_t* lo, int64_t* hi) {
    if (!lo || !hi || nshards < 1 || shard < 0 || shard >= nshards || params < 0) return MDT_EARG;
    long long l, h;
    shard_bounds(params, shard, nshards, &l, &h);
    *lo = l; *hi = h;
    return MDT_OK;
}

int mdt_ctx_status(mdt_ctx* ctx, int32_t* flags, int32_t clear) {
    if (!ctx || !flags) return MDT_EARG;
    *flags = 0;
    if (!ctx->children.empty()) {
        for (mdt_ctx* c : ctx->children) {
            int32_t f = 0;
            const int rc = mdt_ctx_status(c, &f, clear);
            if (rc) return fail(ctx, rc, c->err);
            *flags |= f;
        }
        return MDT_OK;
    }
    MDT_ON_DEVICE(ctx);
    int st[3] = {0, 0, 0};
    if (ctx->used) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    HIP_TRY(hipMemcpy(st, ctx->status, sizeof(st), hipMemcpyDeviceToHost));
    *flags = st[0] | st[2];
    if (clear && *flags) {
        st[0] = 0; st[2] = 0;
        HIP_TRY(hipMemcpy(ctx->status, st, sizeof(st), hipMemcpyHostToDevice));
    }
    return MDT_OK;
}

int mdt_ess_rhat(mdt_ctx* ctx, const mdt_args* a, const void* x, void* ess_out, void* rhat_out) {
    if (!ctx) return MDT_EARG;
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    const int want_ess = (a->outputs & MDT_OUT_ESS) != 0, want_rhat = (a->outputs & MDT_OUT_RHAT) != 0;
    if (!want_ess && !want_rhat) return fail(ctx, MDT_EARG, "outputs selects nothing");
    if ((want_ess && !ess_out) || (want_rhat && !rhat_out)) return fail(ctx, MDT_EARG, "requested output pointer is NULL");
    Req rq;
    rq.want_ess = want_ess;
    switch (a->kind) {
    case MDT_KIND_BASIC:
        rq.basic = 1; rq.need_eb = want_ess; rq.need_rb = want_rhat;
        rq.out_eb = want_ess ? ess_out : nullptr; rq.out_rb = want_rhat ? rhat_out : nullptr;
        break;
    case MDT_KIND_BULK:
        rq.need_eb = want_ess; rq.need_rb = want_rhat;
        rq.out_eb = want_ess ? ess_out : nullptr; rq.out_rb = want_rhat ? rhat_out : nullptr;
        break;
    case MDT_KIND_TAIL:
        rq.need_et = want_ess; rq.need_rt = want_rhat;
        rq.out_et = want_ess ? ess_out : nullptr; rq.out_rt = want_rhat ? rhat_out : nullptr;
        break;
    case MDT_KIND_RANK: // ess = bulk ESS, rhat = max(rhat_tail, rhat_bulk)  (src/ess_rhat.jl:617-624)
        rq.need_eb = want_ess; rq.need_rb = want_rhat; rq.need_rt = want_rhat;
        rq.out_eb = want_ess ? ess_out : nullptr; rq.out_rrank = want_rhat ? rhat_out : nullptr;
        break;
    case MDT_KIND_MEDIAN: case MDT_KIND_STD: case MDT_KIND_MAD: case MDT_KIND_QUANTILE:
        // ess(x; kind = estimator) through its expectand proxy (src/ess_rhat.jl:291-297, 626-659); rhat and
        // ess_rhat take symbols only (:335-349, :438-455)
        if (want_rhat) return fail(ctx, MDT_EARG, "estimator kinds are supported by `ess` only");
        rq.est = a->kind; rq.out_eb = ess_out;
        rq.basic = a->kind == MDT_KIND_STD; // no ranking for the std proxy
        break;
    default:
        return fail(ctx, MDT_EARG, "the `kind` is not supported"); // src/ess_rhat.jl:286,347,453
    }
    return run(ctx, a, x, rq);
}

int mdt_bulk_tail(mdt_ctx* ctx, const mdt_args* a, const void* x, void* ess_bulk, void* ess_tail,
                  void* rhat_bulk, void* rhat_tail) {
    if (!ctx) return MDT_EARG;
    Req rq;
    rq.need_eb = ess_bulk != nullptr; rq.need_et = ess_tail != nullptr;
    rq.need_rb = rhat_bulk != nullptr; rq.need_rt = rhat_tail != nullptr;
    rq.out_eb = ess_bulk; rq.out_et = ess_tail; rq.out_rb = rhat_bulk; rq.out_rt = rhat_tail;
    rq.want_ess = rq.need_eb || rq.need_et;
    if (!(rq.need_eb || rq.need_et || rq.need_rb || rq.need_rt)) return fail(ctx, MDT_EARG, "all outputs are NULL");
    return run(ctx, a, x, rq);
}

int mdt_rhat_nested(mdt_ctx* ctx, const mdt_args* a, const void* x, const int32_t* chain_inds,
                    int32_t K, int32_t S, void* rhat_out) {
    if (!ctx) return MDT_EARG;
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    if (!chain_inds || !rhat_out) return fail(ctx, MDT_EARG, "NULL pointer");
    if (S < 2) return fail(ctx, MDT_EARG, "at least 2 superchains are required"); // src/rhat_nested.jl:76-77
    if (K < 1 || (long long)K * S != a->chains)
        return fail(ctx, MDT_EDIM, "`superchain_ids` does not cover the chains of `samples`"); // :69-73
    for (int i = 0; i < K * S; ++i)
        if (chain_inds[i] < 0 || chain_inds[i] >= a->chains) return fail(ctx, MDT_EARG, "chain index out of range");
    Req rq;
    rq.mode = 1;
    rq.chain_inds = chain_inds; rq.K = K; rq.S = S;
    switch (a->kind) {
    case MDT_KIND_BASIC: rq.basic = 1; rq.need_rb = 1; rq.out_rb = rhat_out; break;
    case MDT_KIND_BULK: rq.need_rb = 1; rq.out_rb = rhat_out; break;
    case MDT_KIND_TAIL: rq.need_rt = 1; rq.out_rt = rhat_out; break;
    case MDT_KIND_RANK: rq.need_rb = 1; rq.need_rt = 1; rq.out_rrank = rhat_out; break; // map(max, Rbulk, Rtail), :124
    default: return fail(ctx, MDT_EARG, "the `kind` is not supported by `rhat_nested`"); // :64
    }
    return run(ctx, a, x, rq);
}

int mdt_mcse_nested(mdt_ctx* ctx, const mdt_args* a, const void* x, const int32_t* chain_inds, int32_t K, int32_t S,
                    void* mcse_out, void* rhat_out) {
    if (!ctx) return MDT_EARG;
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    if (!chain_inds || !rhat_out || !mcse_out) return fail(ctx, MDT_EARG, "NULL pointer");
    if (S < 2) return fail(ctx, MDT_EARG, "at least 2 superchains are required"); // src/rhat_nested.jl:76-77
    if (K < 1 || (long long)K * S != a->chains)
        return fail(ctx, MDT_EDIM, "`superchain_ids` does not cover the chains of `samples`"); // :69-73
    for (int i = 0; i < K * S; ++i)
        if (chain_inds[i] < 0 || chain_inds[i] >= a->chains) return fail(ctx, MDT_EARG, "chain index out of range");
    Req rq;
    rq.mode = 6;
    rq.chain_inds = chain_inds; rq.K = K; rq.S = S;
    rq.out_mcse = mcse_out; rq.want_ess = 1; // (mcse needs the basic ESS: the maxlag domain check of src/ess_rhat.jl:481 applies)
    switch (a->kind) {
    case MDT_KIND_BASIC: rq.basic = 1; rq.need_rb = 1; rq.out_rb = rhat_out; break;
    case MDT_KIND_BULK: rq.need_rb = 1; rq.out_rb = rhat_out; break;
    case MDT_KIND_TAIL: rq.need_rt = 1; rq.out_rt = rhat_out; break;
    case MDT_KIND_RANK: rq.need_rb = 1; rq.need_rt = 1; rq.out_rrank = rhat_out; break; // map(max, Rbulk, Rtail), src/rhat_nested.jl:124
    default: return fail(ctx, MDT_EARG, "the `kind` is not supported by `rhat_nested`"); // :64
    }
    return run(ctx, a, x, rq);
}

int mdt_mcse_mean(mdt_ctx* ctx, const mdt_args* a, const void* x, void* mcse_out) {
    if (!ctx) return MDT_EARG;
    if (!mcse_out) return fail(ctx, MDT_EARG, "NULL output");
    Req rq;
    rq.mode = 2; rq.basic = 1; rq.need_eb = 1; rq.out_eb = mcse_out; rq.want_ess = 1;
    return run(ctx, a, x, rq);
}

int mdt_mcse_std(mdt_ctx* ctx, const mdt_args* a, const void* x, void* mcse_out) {
    if (!ctx) return MDT_EARG;
    if (!mcse_out) return fail(ctx, MDT_EARG, "NULL output");
    Req rq;
    rq.mode = 4; rq.basic = 1; rq.est = MDT_KIND_STD; rq.out_eb = mcse_out; rq.want_ess = 1;
    return run(ctx, a, x, rq);
}

int mdt_mcse_quantile(mdt_ctx* ctx, const mdt_args* a, const void* x, void* mcse_out) {
    if (!ctx) return MDT_EARG;
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    if (!mcse_out) return fail(ctx, MDT_EARG, "NULL output");
    if (!ctx->children.empty()) { // every block runs its two passes on its own GPU
        void* outs[1] = {mcse_out};
        return for_shards(ctx, a, x, outs, 1, [](mdt_ctx* child, const mdt_args* b, const void* xs, void* const* lo_outs) {
            return mdt_mcse_quantile(child, b, xs, lo_outs[0]);
        });
    }
    if (a->kind != MDT_KIND_MEDIAN && a->kind != MDT_KIND_QUANTILE)
        return fail(ctx, MDT_EARG, "mdt_mcse_quantile: kind must be MDT_KIND_MEDIAN or MDT_KIND_QUANTILE");
    const long long P = a->params, n = a->draws * a->chains;
    if (P <= 0) return run(ctx, a, x, Req());
    const double p = (a->kind == MDT_KIND_MEDIAN) ? 0.5 : a->tail_prob;
    const size_t ts = tsize(a->dtype);
    MDT_ON_DEVICE(ctx);
    hipStream_t st = (hipStream_t)a->stream;
    if (P > ctx->mq_cap) {
        if (ctx->ess_tmp) (void)hipFree(ctx->ess_tmp);
        if (ctx->tpos_arr) (void)hipFree(ctx->tpos_arr);
        ctx->ess_tmp = nullptr; ctx->tpos_arr = nullptr; ctx->mq_cap = 0;
        HIP_TRY(hipMalloc(&ctx->ess_tmp, (size_t)P * 8));
        HIP_TRY(hipMalloc((void**)&ctx->tpos_arr, (size_t)P * 2 * sizeof(unsigned int)));
        ctx->mq_cap = P;
    }
    // pass 1: S = ess(x; kind = median | quantile(p))  (src/mcse.jl:73, 84), kept on the device
    mdt_args a1 = *a;
    a1.location = MDT_DEVICE;
    a1.sync = 0;
    const void* dx = x;
    if (a->location == MDT_HOST) { // stage x once for both passes: its whole extent, whatever lies between its pieces (as run() does)
        const long long cs = a->chain_stride ? a->chain_stride : a->draws, ps = a->param_stride ? a->param_stride : n;
        if (cs < a->draws || ps < 1) return fail(ctx, MDT_EARG, "chain_stride must be >= draws and param_stride >= 1 (the draws of a chain are contiguous)");
        if (a->draws <= 0 || a->chains <= 0) return fail(ctx, MDT_EDIM, "empty sample dimensions");
        const size_t xb = (size_t)((P - 1) * ps + (a->chains - 1) * cs + a->draws) * ts;
        if (int rcs = stage_x_bytes(ctx, xb)) return rcs;
        HIP_TRY(hipMemcpyAsync(ctx->stage_x, x, xb, hipMemcpyHostToDevice, st));
        dx = ctx->stage_x;
    }
    Req r1;
    r1.est = a->kind; r1.want_ess = 1;
    r1.out_eb = ctx->ess_tmp;
    int rc = run(ctx, &a1, dx, r1);
    if (rc) return rc;
    // positions of the two quadrature points of the Beta error distribution (src/mcse.jl:93-108): one thread
    // per parameter, same stream
    if (a->dtype == MDT_F32)
        hipLaunchKernelGGL(mcse_quantile_pos_kernel<float>, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, st, (const float*)ctx->ess_tmp, P, p, n, ctx->tpos_arr);
    else
        hipLaunchKernelGGL(mcse_quantile_pos_kernel<double>, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, st, (const double*)ctx->ess_tmp, P, p, n, ctx->tpos_arr);
    HIP_TRY(hipGetLastError());
    // pass 2: the two order statistics, (xu - xl) / 2  (src/mcse.jl:109-117)
    Req r2;
    r2.mode = 5; r2.out_eb = mcse_out; r2.tpos_arr = ctx->tpos_arr; r2.keep_status = 1;
    if (a->location == MDT_HOST) { // x is already staged: run on the device copy, bring the result back
        const size_t ob = (size_t)P * ts;
        if (int rcs = stage_out_bytes(ctx, ob)) return rcs;
        r2.out_eb = ctx->stage_out;
        a1.sync = 1;
        rc = run(ctx, &a1, dx, r2);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(mcse_out, ctx->stage_out, ob, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return MDT_OK;
    }
    return run(ctx, a, x, r2);
}

int mdt_layout_from_dpc(mdt_ctx* ctx, const mdt_args* a, const void* x_dpc, void* x_dcp) {
    if (!ctx) return MDT_EARG;
    if (!ctx->children.empty()) return fail(ctx, MDT_EUNSUPPORTED, "this entry point does not run on a multi-GPU context");
    if (!a || !x_dpc || !x_dcp) return fail(ctx, MDT_EARG, "NULL pointer");
    if (a->location != MDT_DEVICE) return fail(ctx, MDT_EARG, "mdt_layout_from_dpc works on device buffers");
    if (a->dtype != MDT_F64 && a->dtype != MDT_F32) return fail(ctx, MDT_EARG, "dtype must be MDT_F64 or MDT_F32");
    if (a->draws < 0 || a->chains < 0 || a->params < 0) return fail(ctx, MDT_EDIM, "negative dimension");
    if (x_dpc == x_dcp) return fail(ctx, MDT_EARG, "the permutation is out of place");
    if (a->chain_stride || a->param_stride) return fail(ctx, MDT_EARG, "mdt_layout_from_dpc takes dense buffers (chain_stride = param_stride = 0)");
    const long long runs = a->chains * a->params;
    if (runs == 0 || a->draws == 0) return MDT_OK;
    if (runs > 2147483647LL) return fail(ctx, MDT_EUNSUPPORTED, "chains*params too large");
    MDT_ON_DEVICE(ctx);
    hipStream_t st = (hipStream_t)a->stream;
    if (a->dtype == MDT_F64)
        hipLaunchKernelGGL(permute_dpc_kernel<double>, dim3((unsigned)runs), dim3(256), 0, st, (const double*)x_dpc, (double*)x_dcp,
                           (long long)a->draws, (long long)a->chains, (long long)a->params);
    else
        hipLaunchKernelGGL(permute_dpc_kernel<float>, dim3((unsigned)runs), dim3(256), 0, st, (const float*)x_dpc, (float*)x_dcp,
                           (long long)a->draws, (long long)a->chains, (long long)a->params);
    HIP_TRY(hipGetLastError());
    if (a->sync) HIP_TRY(hipStreamSynchronize(st));
    return MDT_OK;
}

// launches the column pass: (mean, var, msd) of every column into ctx->mom, or bfmi straight into bfmi_dev;
// x is staged when it lives in host memory
static int column_moments(mdt_ctx* ctx, const mdt_args* a, const void* x, long long ncol, void* bfmi_dev) {
    hipStream_t st = (hipStream_t)a->stream;
    const size_t ts = tsize(a->dtype);
    const long long D = a->draws;
    const void* dx = x;
    if (a->location == MDT_HOST) {
        const size_t xb = (size_t)D * ncol * ts;
        if (int rcs = stage_x_bytes(ctx, xb)) return rcs;
        HIP_TRY(hipMemcpyAsync(ctx->stage_x, x, xb, hipMemcpyHostToDevice, st));
        dx = ctx->stage_x;
    }
    if (!bfmi_dev && ncol > ctx->mom_cap) {
        if (ctx->mom) (void)hipFree(ctx->mom);
        ctx->mom = nullptr; ctx->mom_cap = 0;
        HIP_TRY(hipMalloc((void**)&ctx->mom, (size_t)ncol * 3 * sizeof(double)));
        ctx->mom_cap = ncol;
    }
    const unsigned grid = (unsigned)std::min<long long>(ncol, 16LL * (ctx->num_cu > 0 ? ctx->num_cu : 256));
    if (a->dtype == MDT_F64) {
        auto k = bfmi_dev ? column_moments_kernel<double, true> : column_moments_kernel<double, false>;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, st, (const double*)dx, D, ncol, ctx->mom, (double*)bfmi_dev);
    } else {
        auto k = bfmi_dev ? column_moments_kernel<float, true> : column_moments_kernel<float, false>;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, st, (const float*)dx, D, ncol, ctx->mom, (float*)bfmi_dev);
    }
    HIP_TRY(hipGetLastError());
    return MDT_OK;
}

static int diag_args_ok(mdt_ctx* ctx, const mdt_args* a) {
    if (!a) return fail(ctx, MDT_EARG, "args is NULL");
    if (a->dtype != MDT_F64 && a->dtype != MDT_F32) return fail(ctx, MDT_EARG, "dtype must be MDT_F64 or MDT_F32");
    if (a->location != MDT_HOST && a->location != MDT_DEVICE) return fail(ctx, MDT_EARG, "bad location");
    if (a->draws < 0 || a->chains < 0 || a->params < 0) return fail(ctx, MDT_EDIM, "negative dimension");
    // the column pass reads x as dense (draws, chains, params): a strided request would mix chains and parameters silently
    if (a->chain_stride || a->param_stride) return fail(ctx, MDT_EARG, "this entry point takes the dense layout only (chain_stride = param_stride = 0)");
    return MDT_OK;
}

int mdt_bfmi(mdt_ctx* ctx, const mdt_args* a, const void* energy, void* bfmi_out) {
    if (!ctx) return MDT_EARG;
    if (!ctx->children.empty()) return fail(ctx, MDT_EUNSUPPORTED, "this entry point does not run on a multi-GPU context");
    int rc = diag_args_ok(ctx, a);
    if (rc) return rc;
    const long long ncol = a->chains * a->params;
    if (ncol == 0) return MDT_OK;
    if (!energy || !bfmi_out) return fail(ctx, MDT_EARG, "NULL pointer");
    if (a->draws == 0) return fail(ctx, MDT_EDIM, "empty sample dimensions");
    MDT_ON_DEVICE(ctx);
    hipStream_t st = (hipStream_t)a->stream;
    const size_t ob = (size_t)ncol * tsize(a->dtype);
    void* od = bfmi_out;
    if (a->location == MDT_HOST) {
        if ((rc = stage_out_bytes(ctx, ob))) return rc;
        od = ctx->stage_out;
    }
    if ((rc = column_moments(ctx, a, energy, ncol, od))) return rc;
    if (a->location == MDT_HOST) HIP_TRY(hipMemcpyAsync(bfmi_out, od, ob, hipMemcpyDeviceToHost, st));
    if (a->sync || a->location == MDT_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MDT_OK;
}

int mdt_gelmandiag(mdt_ctx* ctx, const mdt_args* a, const void* x, double alpha, double* psrf, double* psrfci) {
    if (!ctx) return MDT_EARG;
    if (!ctx->children.empty()) return fail(ctx, MDT_EUNSUPPORTED, "this entry point does not run on a multi-GPU context");
    int rc = diag_args_ok(ctx, a);
    if (rc) return rc;
    const long long D = a->draws, C = a->chains, P = a->params;
    if (C < 2) return fail(ctx, MDT_EARG, "Gelman diagnostic requires at least 2 chains"); // src/gelmandiag.jl:3
    if (P == 0) return MDT_OK;
    if (!x || !psrf || !psrfci) return fail(ctx, MDT_EARG, "NULL pointer");
    if (D == 0) return fail(ctx, MDT_EDIM, "empty sample dimensions");
    MDT_ON_DEVICE(ctx);
    hipStream_t st = (hipStream_t)a->stream;
    double* od0 = psrf; double* od1 = psrfci;
    if (a->location == MDT_HOST) {
        if ((rc = stage_out_bytes(ctx, (size_t)P * 16))) return rc;
        od0 = (double*)ctx->stage_out; od1 = od0 + P;
    }
    if (ctx->used && ctx->last_stream != st) HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    hipLaunchKernelGGL(status_begin_kernel, dim3(1), dim3(64), 0, st, ctx->status);
    ctx->last_stream = st; ctx->used = true;
    if ((rc = column_moments(ctx, a, x, C * P, nullptr))) return rc;
    double zq = 0.0; // standard normal quantile of 1 - alpha/2 (a start value only): Newton on erfc
    {
        const double qq = 1 - alpha / 2;
        if (qq > 0.0 && qq < 1.0)
            for (int it = 0; it < 60; ++it) {
                const double f = 0.5 * erfc(-zq * 0.7071067811865476) - qq, pdf = 0.3989422804014327 * exp(-0.5 * zq * zq);
                if (!(pdf > 1e-300)) break;
                double step = f / pdf;
                step = step > 1.0 ? 1.0 : (step < -1.0 ? -1.0 : step);
                zq -= step;
                if (fabs(step) < 1e-12) break;
            }
    }
    hipLaunchKernelGGL(gelman_finish_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, st, (const double*)ctx->mom, D, C, P, alpha,
                       zq, od0, od1, ctx->status);
    HIP_TRY(hipGetLastError());
    if (a->location == MDT_HOST) {
        HIP_TRY(hipMemcpyAsync(psrf, od0, (size_t)P * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(psrfci, od1, (size_t)P * 8, hipMemcpyDeviceToHost, st));
    }
    if (a->sync || a->location == MDT_HOST) {
        int status = 0;
        HIP_TRY(hipMemcpyAsync(&status, ctx->status, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (status & 4) {
            HIP_TRY(hipMemsetAsync(ctx->status, 0, sizeof(int), st));
            return fail(ctx, MDT_EARG, "FDist: the condition nu2 > zero(nu2) is not satisfied"); // Distributions.FDist
        }
    }
    return MDT_OK;
}

int mdt_debug_beta(double a, double b, double v, int32_t inverse, double* out) {
    if (!out || !(a > 0.0) || !(b > 0.0) || !(v >= 0.0 && v <= 1.0)) return MDT_EARG;
    *out = inverse ? mdt_betainvcdf(a, b, v) : mdt_ibeta(a, b, v);
    return MDT_OK;
}

int mdt_debug_phase_cycles(mdt_ctx* ctx, uint64_t* out16, int32_t reset) {
    if (!ctx || !out16) return MDT_EARG;
    if (!ctx->children.empty()) return fail(ctx, MDT_EUNSUPPORTED, "this entry point does not run on a multi-GPU context");
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    if (!ctx->prof) return fail(ctx, MDT_EUNSUPPORTED, "phase counters are off (create the context with MDT_PROF=1 in the environment)");
    MDT_ON_DEVICE(ctx);
    HIP_TRY(hipDeviceSynchronize());
    const size_t cnt = (size_t)ctx->grid * 2 * 16; // (the tiny geometry runs 2 * grid workgroups)
    unsigned long long* h = (unsigned long long*)malloc(cnt * sizeof(unsigned long long));
    if (!h) return MDT_ENOMEM;
    hipError_t e = hipMemcpy(h, ctx->prof, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        for (size_t i = 0; i < cnt; ++i) out16[i % 16] += h[i];
        if (reset) e = hipMemset(ctx->prof, 0, cnt * sizeof(unsigned long long));
    }
    free(h);
    if (e != hipSuccess) return fail(ctx, MDT_ERUNTIME, hipGetErrorString(e));
    return MDT_OK;
}

int mdt_rank_normalize(mdt_ctx* ctx, const mdt_args* a, const void* x, int32_t fold, double* rank_out,
                       void* z_out) {
    if (!ctx) return MDT_EARG;
    if (!rank_out && !z_out) return fail(ctx, MDT_EARG, "all outputs are NULL");
    Req rq;
    rq.mode = 3; rq.fold = fold ? 1 : 0; rq.rank_out = rank_out; rq.z_out = z_out;
    return run(ctx, a, x, rq);
}

} // extern "C"
