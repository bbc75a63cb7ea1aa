// microbenchmark (round 4): issue rates of the vector instructions the ranking is made of -- f64 compares against f32 / u32 / u64
// compares, f64 add / min, conversions -- as cycles per wave-instruction at 1, 2 and 4 waves per SIMD (independent streams).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_bench tools/valu_rate_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 16; // independent streams per wave

#define KERNEL(name, decl, body, fin)                                                            \
    __global__ void name(double* out, long long* cyc, int iters) {                               \
        decl;                                                                                    \
        const long long c0 = clock64();                                                          \
        for (int it = 0; it < iters; ++it) {                                                     \
            _Pragma("unroll") for (int i = 0; i < N; ++i) { body; }                              \
        }                                                                                        \
        const long long c1 = clock64();                                                          \
        double s = 0;                                                                            \
        for (int i = 0; i < N; ++i) s += (double)(fin);                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
        if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;                                         \
    }

// every body is ONE vector instruction per stream (checked in the ISA: tools/valu_rate_bench.s)
KERNEL(k_cmp_f64, double a[N]; unsigned long long m[N]; double b = 1.0 + threadIdx.x * 1e-3; for (int i = 0; i < N; ++i) { a[i] = i + threadIdx.x * 1e-2; m[i] = 0; },
       asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b)), m[i] & 1)
KERNEL(k_cmp_f32, float a[N]; unsigned long long m[N]; float b = 1.0f + threadIdx.x * 1e-3f; for (int i = 0; i < N; ++i) { a[i] = i + threadIdx.x * 1e-2f; m[i] = 0; },
       asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b)), m[i] & 1)
KERNEL(k_cmp_u32, unsigned a[N]; unsigned long long m[N]; unsigned b = threadIdx.x * 77u; for (int i = 0; i < N; ++i) { a[i] = i * 1000u + threadIdx.x; m[i] = 0; },
       asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b)), m[i] & 1)
KERNEL(k_cmp_u64, unsigned long long a[N]; unsigned long long m[N]; unsigned long long b = threadIdx.x * 77ull; for (int i = 0; i < N; ++i) { a[i] = i * 1000ull + threadIdx.x; m[i] = 0; },
       asm volatile("v_cmp_lt_u64 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b)), m[i] & 1)
KERNEL(k_add_f64, double a[N]; double b = 1.0 + threadIdx.x * 1e-3; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_min_f64, double a[N]; double b = 1.0 + threadIdx.x * 1e-3; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_mul_f64, double a[N]; double b = 1.0 + threadIdx.x * 1e-9; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_add_f32, float a[N]; float b = 1.0f + threadIdx.x * 1e-3f; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_add_u32, unsigned a[N]; unsigned b = threadIdx.x; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_cndmask, unsigned a[N]; unsigned b = threadIdx.x; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_cndmask_s, unsigned a[N]; unsigned b = threadIdx.x; unsigned long long msk = 0x5555555555555555ull; asm volatile("" : "+s"(msk)); for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk)), a[i])
KERNEL(k_cmp_cnd, unsigned a[N]; unsigned b = threadIdx.x; for (int i = 0; i < N; ++i) a[i] = i * 3 + threadIdx.x,
       asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"), a[i])
KERNEL(k_cmp_cnd2, unsigned a[N]; unsigned a2[N]; unsigned b = threadIdx.x; for (int i = 0; i < N; ++i) { a[i] = i * 3 + threadIdx.x; a2[i] = i; },
       asm volatile("v_cmp_lt_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(a[i]), "+v"(a2[i]) : "v"(b) : "vcc"), a[i] + a2[i])
KERNEL(k_cmp_cnd4, unsigned a[N]; unsigned a2[N]; unsigned b = threadIdx.x; for (int i = 0; i < N; ++i) { a[i] = i * 3 + threadIdx.x; a2[i] = i; },
       asm volatile("v_cmp_lt_u32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(a[i]), "+v"(a2[i]) : "v"(b) : "vcc"), a[i] + a2[i])
KERNEL(k_cmps_cnd2, unsigned a[N]; unsigned a2[N]; unsigned b = threadIdx.x; unsigned long long m[N]; for (int i = 0; i < N; ++i) { a[i] = i * 3 + threadIdx.x; a2[i] = i; m[i] = 0; },
       asm volatile("v_cmp_lt_u32 %3, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %3\n\tv_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(a[i]), "+v"(a2[i]) : "v"(b), "s"(m[i])), a[i] + a2[i])
KERNEL(k_max_u32, unsigned a[N]; unsigned b = threadIdx.x; for (int i = 0; i < N; ++i) a[i] = i * 3 + threadIdx.x,
       asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_bfi, unsigned a[N]; unsigned b = threadIdx.x; unsigned c = 0xFF00FF00u; for (int i = 0; i < N; ++i) a[i] = i * 3 + threadIdx.x,
       asm volatile("v_bfi_b32 %0, %2, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c)), a[i])
KERNEL(k_fma_f64, double a[N]; double b = 1.0 + threadIdx.x * 1e-9; double c = 0.5; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)), a[i])
KERNEL(k_pk_fma_f32, double a[N]; double b = 1.0 + threadIdx.x * 1e-9; double c = 0.5; for (int i = 0; i < N; ++i) a[i] = i,
       asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)), a[i])
KERNEL(k_lshl_add, unsigned a[N]; unsigned b = threadIdx.x; for (int i = 0; i < N; ++i) a[i] = i * 3 + threadIdx.x,
       asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)), a[i])
KERNEL(k_cvt_f64_i32, double a[N]; int b[N]; for (int i = 0; i < N; ++i) { a[i] = 0; b[i] = i + threadIdx.x; },
       asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(b[i])), a[i])
KERNEL(k_cvt_i32_f64, int a[N]; double b[N]; for (int i = 0; i < N; ++i) { a[i] = 0; b[i] = i + threadIdx.x * 0.5; },
       asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(b[i])), a[i])
KERNEL(k_mov_dpp, unsigned a[N]; for (int i = 0; i < N; ++i) a[i] = i + threadIdx.x,
       asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i])), a[i])

int main() {
    CK(hipSetDevice(0));
    double* out; long long* cyc; CK(hipMalloc(&out, 256 * 1024 * 8)); CK(hipMalloc(&cyc, 256 * 8));
    std::vector<long long> hc(256);
    const int iters = 20000;
    struct { const char* name; void (*k)(double*, long long*, int); } ks[] = {
        {"v_cmp_lt_f64", k_cmp_f64}, {"v_cmp_lt_f32", k_cmp_f32}, {"v_cmp_lt_u32", k_cmp_u32}, {"v_cmp_lt_u64", k_cmp_u64},
        {"v_add_f64", k_add_f64}, {"v_min_f64", k_min_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f32", k_add_f32}, {"v_add_u32", k_add_u32},
        {"v_cndmask_b32", k_cndmask}, {"v_cndmask s[]", k_cndmask_s}, {"cmp+cndmask pair", k_cmp_cnd}, {"cmp+2 cndmask vcc", k_cmp_cnd2}, {"cmp+4 cndmask vcc", k_cmp_cnd4}, {"cmp+2 cndmask s[]", k_cmps_cnd2}, {"v_max_u32", k_max_u32}, {"v_bfi_b32", k_bfi},
        {"v_fma_f64", k_fma_f64}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_lshl_add_u32", k_lshl_add}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_mov_b32_dpp", k_mov_dpp}};
    // (clock64 of wave 0 is not a throughput: with several waves per SIMD the arbiter favours the oldest wave, which then finishes
    // its loop at the one-wave rate while the launch takes proportionally longer -- throughput is taken from the wall clock)
    printf("%-16s %s\n", "instruction", "wave-instructions per ns and SIMD (wall clock) at 1 / 2 / 4 waves per SIMD; in brackets clock64 ticks per instruction of wave 0");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& k : ks) {
        printf("%-18s", k.name);
        for (int bs : {256, 512, 1024}) {
            hipLaunchKernelGGL(k.k, dim3(256), dim3(bs), 0, 0, out, cyc, iters);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.k, dim3(256), dim3(bs), 0, 0, out, cyc, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
            double mc = 0; for (int i = 0; i < 256; ++i) mc += (double)hc[i]; mc /= 256;
            const double winstr = (double)iters * N * (bs / 64.0) * 256; // wave-instructions of the launch
            printf("  %6.3f [%5.2f]", winstr / (ms * 1e6) / 1024.0, mc / iters / N);
        }
        printf("\n");
    }
    return 0;
}
