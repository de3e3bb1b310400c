// Launch overhead of a chain of dependent small kernels: plain stream launches versus one hipGraph launch.
// Decides whether replaying the interpreter's per-level launches from a graph would help small repetition shards
// (32-128 repetitions per GPU, where a level is ~2 us of GPU work and the host cannot enqueue that fast).
// Build: hipcc --offload-arch=gfx950 -O3 graph_mb.hip -o graph_mb.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k(uint32_t* p, uint32_t n, uint32_t iters) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = p[i];
    for (uint32_t t = 0; t < iters; t++) v = v * 1664525u + 1013904223u;
    p[i] = v;
}
int main() {
    const int N = 164;
    uint32_t* d; (void)hipMalloc(&d, 1 << 22); (void)hipMemset(d, 1, 1 << 22);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (uint32_t work : {1u << 14, 1u << 18, 1u << 20}) {
        auto run_stream = [&] {
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k, dim3((work + 255) / 256), dim3(256), 0, st, d, work, 20u);
        };
        run_stream(); (void)hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) run_stream();
        (void)hipStreamSynchronize(st);
        double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (20.0 * N);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        run_stream();
        (void)hipStreamEndCapture(st, &g);
        hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (e != hipSuccess) { printf("instantiate failed %s\n", hipGetErrorString(e)); return 1; }
        (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) (void)hipGraphLaunch(ge, st);
        (void)hipStreamSynchronize(st);
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (20.0 * N);
        printf("chain of %d kernels of %7u threads: stream %.2f us per kernel, graph %.2f us per kernel\n", N, work, us_stream, us_graph);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
