// microbenchmark (round 4): what v_mfma_f64_16x16x4_f64 costs on gfx950, whether it co-executes with v_fma_f64 of other
// waves of the same SIMD, and the lag products of one split chain (src/ess_rhat.jl:161-179) as a blocked Hankel product
// G = U^T V on the matrix pipe against the vector form the kernel used through round 3 (acc_lags<8>).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/mfma_f64_bench tools/mfma_f64_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- 1. issue rate: NACC independent accumulators, back to back
template <int NACC>
__global__ void mfma_rate(double* out, long long* cyc, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}
template <int NACC>
__global__ void valu_rate(double* out, long long* cyc, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, b, acc[i]);
        asm volatile("" : "+v"(a));
    }
    const long long c1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}
// ---- 2. co-execution: waves 0..3 of a 512-thread workgroup (one per SIMD) issue MFMAs, waves 4..7 issue v_fma_f64
// mode 0: both; 1: only the MFMA waves work; 2: only the VALU waves work
__global__ void coexec(double* out, long long* cyc, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    double s = 0;
    const long long c0 = clock64();
    if (w < 4) {
        if (mode != 2) {
            d4 acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        if (mode != 1) {
            double acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = 0;
            for (int it = 0; it < iters * 4; ++it) { // 16 x 4 v_fma_f64 per 4 MFMAs: the same pipe time if both run at 16 FMA / clk / SIMD
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = fma(a, b, acc[i]);
                asm volatile("" : "+v"(a));
            }
            for (int i = 0; i < 16; ++i) s += acc[i];
        }
    }
    const long long c1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = c1 - c0;
}

// ---- 2b. what ELSE shares the pipe: waves 4..7 run one of several instruction mixes beside the MFMA waves
// kind 0: v_fma_f64, 1: 32-bit integer (v_add / v_xor / v_lshl), 2: v_cmp_lt_f64 + v_cndmask, 3: v_fma_f32, 4: ds_read_b64
template <int KIND>
__global__ void coexec2(double* out, long long* cyc, int iters, int mode) {
    __shared__ double lds[4096];
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-3;
    __syncthreads();
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    double s = 0;
    const long long c0 = clock64();
    if (w < 4) {
        if (mode != 2) {
            d4 acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else if (mode != 1) {
        if (KIND == 1) {
            unsigned v[16];
            for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 2654435761u + i;
            for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (v[i] + 0x9E3779B9u) ^ (v[i] << 3);
            }
            for (int i = 0; i < 16; ++i) s += v[i];
        } else if (KIND == 2) {
            double v[16];
            for (int i = 0; i < 16; ++i) v[i] = a + i;
            for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (v[i] < b) ? a : v[(i + 1) & 15];
                asm volatile("" : "+v"(b));
            }
            for (int i = 0; i < 16; ++i) s += v[i];
        } else if (KIND == 3) {
            float v[16];
            float fa = (float)a, fb = (float)b;
            for (int i = 0; i < 16; ++i) v[i] = 0.f;
            for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(fa, fb, v[i]);
                asm volatile("" : "+v"(fa));
            }
            for (int i = 0; i < 16; ++i) s += v[i];
        } else if (KIND == 4) {
            int idx = threadIdx.x & 63;
            for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s += lds[(idx + 64 * i) & 4095];
                asm volatile("" : "+v"(idx));
            }
        }
    }
    const long long c1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = c1 - c0;
}

// ---- 3. the lag products of one chain of n values held in LDS, lags [0, 32)
constexpr int NT = 512, NWAVE = 8, SZ = 6400;
// vector form of rounds 1-3: waves = 4 lag groups of 8 x 2 time segments, two t per lane, 6 ds_read_b128 per 16 FMAs
__global__ __launch_bounds__(512, 2) void lag_valu(const double* z, double* P, int n, int reps, long long* cyc) {
    __shared__ double buf[SZ];
    __shared__ double part[32 * 2];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const double* zc = z + (size_t)blockIdx.x * n;
    for (int i = tid; i < SZ; i += NT) buf[i] = i < n ? zc[i] : 0.0;
    __syncthreads();
    const int G = 4, nseg = 2, g = w % G, q = w / G;
    const int seglen = (((n + nseg - 1) / nseg) + 1) & ~1;
    const int t0 = q * seglen, t1 = (t0 + seglen < n) ? t0 + seglen : n;
    const int kb = 8 * g;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long c0 = clock64();
    for (int r = 0; r < reps; ++r) {
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
        __syncthreads();
    }
    const long long c1 = clock64();
    for (int i = 0; i < 8; ++i) {
        double v = acc[i];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) part[(g * 8 + i) * nseg + q] = v;
    }
    __syncthreads();
    if (tid < 32) P[(size_t)blockIdx.x * 64 + tid] = (part[tid * 2] + part[tid * 2 + 1]) / reps;
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}
// matrix form: G[i][j] += sum_r u[16 r + i] u[16 r + j + c]: per 64 values ONE operand u[64 R + lane] and one shifted
// operand per tile c = 0, 16, 32; lag k of residue i sits in column k + i of the 16 x 48 strip; the waves are 8 time segments
template <int TILES>
__global__ __launch_bounds__(512, 2) void lag_mfma(const double* z, double* P, int n, int reps, long long* cyc) {
    __shared__ double buf[SZ];
    __shared__ double gt[NWAVE][16][16 * TILES + 1];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const double* zc = z + (size_t)blockIdx.x * n;
    for (int i = tid; i < SZ; i += NT) buf[i] = i < n ? zc[i] : 0.0;
    __syncthreads();
    const int nblk = (n + 63) >> 6;
    const int bper = (nblk + NWAVE - 1) / NWAVE;
    const int b0 = w * bper, b1 = (b0 + bper < nblk) ? b0 + bper : nblk;
    d4 acc[TILES];
    for (int i = 0; i < TILES; ++i) acc[i] = d4{0, 0, 0, 0};
    const long long c0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (int b = b0; b < b1; ++b) {
            const double* p = buf + (b << 6) + lane;
            const double a = p[0];
#pragma unroll
            for (int i = 0; i < TILES; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, p[16 * i], acc[i], 0, 0, 0);
        }
        __syncthreads();
    }
    const long long c1 = clock64();
    // C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
    for (int i = 0; i < TILES; ++i)
        for (int rg = 0; rg < 4; ++rg) gt[w][(lane >> 4) + 4 * rg][16 * i + (lane & 15)] = acc[i][rg];
    __syncthreads();
    if (tid < 16 * (TILES - 1)) {
        double sm = 0.0;
        for (int q = 0; q < NWAVE; ++q)
            for (int i = 0; i < 16; ++i) sm += gt[q][i][tid + i];
        P[(size_t)blockIdx.x * 64 + tid] = sm / reps;
    }
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    printf("%s, %d CUs, clock %d kHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
    double* out; long long* cyc; CK(hipMalloc(&out, 2048 * 1024 * 8)); CK(hipMalloc(&cyc, 4096 * 8 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<long long> hc(4096 * 8);
    auto timeit = [&](const char* name, int grid, int ncyc, double work_per_block, const char* unit, auto f) {
        float ms;
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc.data(), cyc, (size_t)ncyc * 8, hipMemcpyDeviceToHost));
        double mc = 0; for (int i = 0; i < ncyc; ++i) mc += (double)hc[i]; mc /= ncyc;
        printf("%-44s grid %5d  %8.3f ms  mean clock64 ticks %12.0f  -> %.3e %s/s chip\n", name, grid, ms, mc, work_per_block * grid / (ms * 1e-3), unit);
        return mc;
    };
    const int iters = 20000;
    // one wave per SIMD (256-thread blocks, one block per CU), then 2 and 4 waves per SIMD
    for (int bs : {256, 512, 1024}) {
        const int grid = 256;
        const double nw = bs / 64.0;
        double mc;
        mc = timeit("mfma f64 16x16x4, 1 acc", grid, grid, nw * iters * 1.0 * 2048, "flop", [&] { hipLaunchKernelGGL(mfma_rate<1>, dim3(grid), dim3(bs), 0, 0, out, cyc, iters); });
        printf("    bs %d: ticks per MFMA per wave %.2f\n", bs, mc / iters / 1.0);
        mc = timeit("mfma f64 16x16x4, 4 acc", grid, grid, nw * iters * 4.0 * 2048, "flop", [&] { hipLaunchKernelGGL(mfma_rate<4>, dim3(grid), dim3(bs), 0, 0, out, cyc, iters); });
        printf("    bs %d: ticks per MFMA per wave %.2f\n", bs, mc / iters / 4.0);
        mc = timeit("v_fma_f64, 8 acc", grid, grid, nw * iters * 8.0 * 128, "flop", [&] { hipLaunchKernelGGL(valu_rate<8>, dim3(grid), dim3(bs), 0, 0, out, cyc, iters); });
        printf("    bs %d: ticks per v_fma_f64 per wave %.2f\n", bs, mc / iters / 8.0);
    }
    for (int mode : {1, 2, 0}) {
        const int grid = 256;
        float ms;
        auto f = [&] { hipLaunchKernelGGL(coexec, dim3(grid), dim3(512), 0, 0, out, cyc, 5000, mode); };
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hc.data(), cyc, (size_t)grid * 8 * 8, hipMemcpyDeviceToHost));
        double m0 = 0, m1 = 0; for (int i = 0; i < grid; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? m0 : m1) += (double)hc[i * 8 + w];
        printf("coexec mode %d (%s): %.3f ms; mean ticks MFMA waves %.0f, VALU waves %.0f\n", mode, mode == 0 ? "both" : mode == 1 ? "MFMA waves only" : "VALU waves only", ms, m0 / grid / 4, m1 / grid / 4);
    }
    auto co2 = [&](const char* name, auto kern) {
        for (int mode : {1, 2, 0}) {
            const int grid = 256;
            float ms;
            auto f = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, cyc, 5000, mode); };
            f(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hc.data(), cyc, (size_t)grid * 8 * 8, hipMemcpyDeviceToHost));
            double m0 = 0, m1 = 0; for (int i = 0; i < grid; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? m0 : m1) += (double)hc[i * 8 + w];
            printf("coexec with %-28s mode %d (%s): %.3f ms; mean ticks MFMA waves %.0f, other waves %.0f\n", name, mode, mode == 0 ? "both" : mode == 1 ? "MFMA waves only" : "other waves only", ms, m0 / grid / 4, m1 / grid / 4);
        }
    };
    co2("32-bit integer VALU", coexec2<1>);
    co2("v_cmp_f64 + v_cndmask", coexec2<2>);
    co2("v_fma_f32", coexec2<3>);
    co2("ds_read_b64", coexec2<4>);
    // lag products
    const int n = 5000, nch = 512, reps = 200;
    std::vector<double> hz((size_t)nch * n);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : hz) { double acc = 0; for (int k = 0; k < 12; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; acc += (double)(s >> 11) / 9007199254740992.0; } v = acc - 6.0; }
    for (int c = 0; c < nch; ++c) for (int t = 1; t < n; ++t) hz[(size_t)c * n + t] = 0.7 * hz[(size_t)c * n + t - 1] + 0.714 * hz[(size_t)c * n + t];
    double* dz; double *dP0, *dP1; CK(hipMalloc(&dz, hz.size() * 8)); CK(hipMalloc(&dP0, nch * 64 * 8)); CK(hipMalloc(&dP1, nch * 64 * 8));
    CK(hipMemcpy(dz, hz.data(), hz.size() * 8, hipMemcpyHostToDevice));
    const double useful = (double)reps * 32.0 * n * 2.0;
    timeit("lag products, vector form (32 lags)", nch, nch, useful, "useful flop", [&] { hipLaunchKernelGGL(lag_valu, dim3(nch), dim3(512), 0, 0, dz, dP0, n, reps, cyc); });
    timeit("lag products, matrix form (3 tiles)", nch, nch, useful, "useful flop", [&] { hipLaunchKernelGGL(lag_mfma<3>, dim3(nch), dim3(512), 0, 0, dz, dP1, n, reps, cyc); });
    std::vector<double> p0(nch * 64), p1(nch * 64);
    CK(hipMemcpy(p0.data(), dP0, p0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), dP1, p1.size() * 8, hipMemcpyDeviceToHost));
    double worst0 = 0, worst1 = 0;
    for (int c = 0; c < 8; ++c) for (int k = 0; k < 32; ++k) {
        long double ref = 0; for (int t = 0; t + k < n; ++t) ref += (long double)hz[(size_t)c * n + t] * hz[(size_t)c * n + t + k];
        worst0 = fmax(worst0, fabs((double)(p0[c * 64 + k] - ref)) / fabs((double)ref));
        worst1 = fmax(worst1, fabs((double)(p1[c * 64 + k] - ref)) / fabs((double)ref));
    }
    printf("lag products against a long double host sum: vector form max rel err %.2e, matrix form %.2e\n", worst0, worst1);
    timeit("lag products, matrix form (5 tiles = 64 lags)", nch, nch, (double)reps * 64.0 * n * 2.0, "useful flop", [&] { hipLaunchKernelGGL(lag_mfma<5>, dim3(nch), dim3(512), 0, 0, dz, dP1, n, reps, cyc); });
    return 0;
}
