// What does a launch cost on this stack, eager vs hipGraph, one stream vs fork/join?  Decides how the step's ~1000 launches
// should be issued.  Build: hipcc --offload-arch=gfx950 -O2 tools/launch_probe.hip -o gpurun_out/launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void work(float* p, int iters) {
    float v = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = v;
}
struct BigArgs { float* p; int iters; char pad[360]; };     // a GemmArgs-sized kernarg block
__global__ __launch_bounds__(256) void work_big(BigArgs a) {
    float v = a.p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < a.iters; ++i) v = v * 1.0001f + 0.5f;
    a.p[blockIdx.x * 256 + threadIdx.x] = v;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    float* buf;
    CK(hipMalloc(&buf, 1 << 22));
    CK(hipMemset(buf, 0, 1 << 22));
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev[2];
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const int N = 1000;
    struct Cfg { int blocks, iters; const char* name; } cfgs[] = {
        {256, 8, "tiny kernel (256 WG, ~2 us)"}, {1024, 2000, "medium kernel (1024 WG, ~10 us)"}};
    for (auto& c : cfgs) {
        // ---- eager, one stream
        for (int w = 0; w < 2; ++w) {
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(c.blocks), dim3(256), 0, s, buf, c.iters);
            const double t1 = now_us();
            CK(hipStreamSynchronize(s));
            const double t2 = now_us();
            if (w) printf("%-34s eager 1 stream      : host %.2f us/launch, wall %.2f us/launch\n", c.name, (t1 - t0) / N, (t2 - t0) / N);
        }
        for (int w = 0; w < 2; ++w) {
            BigArgs a; a.p = buf; a.iters = c.iters;
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work_big, dim3(c.blocks), dim3(256), 0, s, a);
            const double t1 = now_us();
            CK(hipStreamSynchronize(s));
            const double t2 = now_us();
            if (w) printf("%-34s eager, 376 B kernarg: host %.2f us/launch, wall %.2f us/launch\n", c.name, (t1 - t0) / N, (t2 - t0) / N);
        }
        // ---- eager with a fork/join to a side stream every 10 launches
        for (int w = 0; w < 2; ++w) {
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) {
                if (i % 10 == 5) {
                    hipEventRecord(ev[0], s); hipStreamWaitEvent(side, ev[0], 0);
                    hipLaunchKernelGGL(work, dim3(c.blocks), dim3(256), 0, side, buf + (1 << 19), c.iters);
                    hipEventRecord(ev[1], side); hipStreamWaitEvent(s, ev[1], 0);
                } else {
                    hipLaunchKernelGGL(work, dim3(c.blocks), dim3(256), 0, s, buf, c.iters);
                }
            }
            const double t1 = now_us();
            CK(hipStreamSynchronize(s));
            const double t2 = now_us();
            if (w) printf("%-34s eager fork/join /10  : host %.2f us/launch, wall %.2f us/launch\n", c.name, (t1 - t0) / N, (t2 - t0) / N);
        }
        // ---- graph, one stream
        for (int variant = 0; variant < 3; ++variant) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) {
                if (variant == 1 && i % 10 == 5) {
                    hipEventRecord(ev[0], s); hipStreamWaitEvent(side, ev[0], 0);
                    hipLaunchKernelGGL(work, dim3(c.blocks), dim3(256), 0, side, buf + (1 << 19), c.iters);
                    hipEventRecord(ev[1], side); hipStreamWaitEvent(s, ev[1], 0);
                } else if (variant == 2 && i % 10 == 5) {
                    hipMemsetAsync(buf + (1 << 20), 0, 4096, s);
                } else {
                    hipLaunchKernelGGL(work, dim3(c.blocks), dim3(256), 0, s, buf, c.iters);
                }
            }
            CK(hipStreamEndCapture(s, &g));
            const double ti0 = now_us();
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const double ti1 = now_us();
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            double host = 0, wall = 0;
            const int R = 5;
            for (int rep = 0; rep < R; ++rep) {
                const double t0 = now_us();
                CK(hipGraphLaunch(ge, s));
                const double t1 = now_us();
                CK(hipStreamSynchronize(s));
                const double t2 = now_us();
                host += t1 - t0; wall += t2 - t0;
            }
            const char* vn[] = {"graph 1 stream       ", "graph fork/join /10  ", "graph + memset /10   "};
            printf("%-34s %s: host %.2f us/node, wall %.2f us/node (instantiate %.0f us)\n", c.name, vn[variant], host / R / N,
                   wall / R / N, ti1 - ti0);
            hipGraphExecDestroy(ge);
            hipGraphDestroy(g);
        }
    }
    return 0;
}
