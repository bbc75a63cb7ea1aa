// microbenchmark: per-workgroup random scatter of 16-B and 8-B elements into a private 640 KB / 320 KB region
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void scat16(double2* dst, const unsigned* perm, int n, int reps) {
    double2* d = dst + (size_t)blockIdx.x * n;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < n; i += blockDim.x) { double2 v; v.x = i; v.y = r; d[perm[i]] = v; }
}
__global__ void scat8(double* dst, const unsigned* perm, int n, int reps) {
    double* d = dst + (size_t)blockIdx.x * n;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < n; i += blockDim.x) d[perm[i]] = (double)i + r;
}
__global__ void gath8(double* out, const double* src, const unsigned* perm, int n, int reps) {
    const double* s = src + (size_t)blockIdx.x * n;
    double acc = 0;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += s[perm[i]];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void copy8(double* dst, const double* src, int n, int reps) {
    double* d = dst + (size_t)blockIdx.x * n; const double* s = src + (size_t)blockIdx.x * n;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i] + r;
}
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    const int n = 40000, reps = 20;
    std::vector<unsigned> p(n);
    for (int i = 0; i < n; ++i) p[i] = i;
    unsigned s = 12345;
    for (int i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; int j = s % (i + 1); std::swap(p[i], p[j]); }
    unsigned* dp; CK(hipMalloc(&dp, n * 4)); CK(hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice));
    for (int grid : {256, 512, 1024}) for (int bs : {256, 512, 1024}) {
        double2* d16; double *d8, *s8, *o; CK(hipMalloc(&d16, (size_t)grid * n * 16)); CK(hipMalloc(&d8, (size_t)grid * n * 8)); CK(hipMalloc(&s8, (size_t)grid * n * 8)); CK(hipMalloc(&o, (size_t)grid * bs * 8));
        CK(hipMemset(s8, 0, (size_t)grid * n * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms;
        auto run = [&](const char* name, auto f) {
            f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("grid %4d bs %4d %-8s %8.3f ms  -> %.2f us per 40000-element pass per WG-slot, %.1f G elem/s chip\n", grid, bs, name, ms, ms * 1e3 / reps / (grid / 256.0), (double)grid * n * reps / ms / 1e6);
        };
        run("scat16", [&] { hipLaunchKernelGGL(scat16, dim3(grid), dim3(bs), 0, 0, d16, dp, n, reps); });
        run("scat8", [&] { hipLaunchKernelGGL(scat8, dim3(grid), dim3(bs), 0, 0, d8, dp, n, reps); });
        run("gath8", [&] { hipLaunchKernelGGL(gath8, dim3(grid), dim3(bs), 0, 0, o, s8, dp, n, reps); });
        run("copy8", [&] { hipLaunchKernelGGL(copy8, dim3(grid), dim3(bs), 0, 0, d8, s8, n, reps); });
        CK(hipFree(d16)); CK(hipFree(d8)); CK(hipFree(s8)); CK(hipFree(o));
    }
    return 0;
}
