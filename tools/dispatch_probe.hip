// How fast can the chip start workgroups?  Times grids of short-lived workgroups with the footprint of a single-K-tile GEMM
// (256 threads, 24 KiB dynamic LDS) against variants without LDS / with fewer, longer workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void touch(const float4* __restrict__ in, float4* __restrict__ out, int per_thread) {
    extern __shared__ float4 sh[];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * per_thread;
    float4 acc = {0, 0, 0, 0};
    for (int i = 0; i < per_thread; ++i) {
        const float4 v = in[base + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (threadIdx.x == 0) sh[0] = acc;
    __syncthreads();
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const size_t bytes = 256ull << 20;
    float4 *in, *out;
    hipMalloc(&in, bytes);
    hipMalloc(&out, bytes);
    hipMemset(in, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    struct Cfg { int blocks, lds, per; const char* name; } cfgs[] = {
        {3136, 24576, 6, "3136 WG x 256 thr, 24 KiB LDS, 96 B/thread (1x1-conv GEMM footprint)"},
        {3136, 0, 6, "3136 WG x 256 thr, no LDS"},
        {6272, 16384, 3, "6272 WG, 16 KiB LDS, 48 B/thread"},
        {784, 24576, 24, "784 WG, 24 KiB LDS, 384 B/thread"},
        {256, 24576, 72, "256 WG, 24 KiB LDS, 1152 B/thread (one per CU)"},
        {12544, 24576, 1, "12544 WG, 24 KiB LDS, 16 B/thread"},
    };
    for (auto& c : cfgs) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(touch, dim3(c.blocks), dim3(256), c.lds, 0, in, out, c.per);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(touch, dim3(c.blocks), dim3(256), c.lds, 0, in, out, c.per);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        const double rd = (double)c.blocks * 256 * c.per * 16, wr = (double)c.blocks * 256 * 16;
        printf("%-70s %7.1f us  (%.0f MB read + %.0f MB written -> %.2f TB/s)\n", c.name, best * 1e3, rd / 1e6, wr / 1e6,
               (rd + wr) / (best * 1e-3) / 1e12);
    }
    return 0;
}
