// Microbenchmark (diagnostics, needs a GPU): cycles per LDS read instruction by width and address pattern --
// all 64 lanes one address (broadcast), 2 / 4 / 8 distinct addresses in lane groups, consecutive addresses.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_bcast_bench tools/lds_bcast_bench.hip && tools/lds_bcast_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W> struct V;
template <> struct V<4> { typedef unsigned t; };
template <> struct V<8> { typedef unsigned long long t; };
typedef unsigned u3v __attribute__((ext_vector_type(3)));
template <> struct V<12> { typedef u3v t; };
template <> struct V<16> { typedef uint4 t; };
template <int W>
__global__ void k(long long* out, int groups, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((unsigned*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // groups = 0: consecutive addresses; g > 0: lanes in g groups, one address per group (groups spread over the banks)
    unsigned off = groups == 0 ? lane * 16 : (lane * groups / 64) * 272;
    unsigned acc = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        typename V<W>::t v[16];
        // sixteen independent reads in flight, one wait (throughput, not latency)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned a = off + ((u * 1040 + it * 16) & 16383);
            if constexpr (W == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u]) : "v"(a));
            else if constexpr (W == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(a));
            else if constexpr (W == 12) asm volatile("ds_read_b96 %0, %1" : "=v"(v[u]) : "v"(a));
            else asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (W == 4) acc += v[u];
            else if constexpr (W == 8) acc += (unsigned)v[u];
            else acc += v[u].x;
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) out[0] = 0;
}
template <int W>
void run(const char* nm, int waves) {
    long long* d; hipMalloc(&d, 8 * 1024);
    for (int g : {1, 2, 4, 8, 16, 32, 0}) {
        const int iters = 200;
        hipLaunchKernelGGL(k<W>, dim3(256), dim3(64 * waves), 0, 0, d, g, iters);
        hipLaunchKernelGGL(k<W>, dim3(256), dim3(64 * waves), 0, 0, d, g, iters);
        long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-8s waves/CU %2d  %-22s %7.1f cycles per read instruction (per wave), %6.2f per CU\n", nm, waves, g == 0 ? "consecutive" : (g == 1 ? "one address" : (g == 2 ? "2 addresses" : g == 4 ? "4 addresses" : g == 8 ? "8 addresses" : g == 16 ? "16 addresses" : "32 addresses")), (double)h / (iters * 16), (double)h / (iters * 16) / waves);
    }
    hipFree(d);
}
int main() {
    for (int waves : {1, 16}) {
        run<4>("b32", waves); run<8>("b64", waves); run<12>("b96", waves); run<16>("b128", waves);
    }
    return 0;
}
