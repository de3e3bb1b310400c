#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../reverie_amd/csrc/b3.h"
template <int N>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t cv[N][8], m[N][16];
    for (int i = 0; i < N; i++) { for (int k = 0; k < 8; k++) cv[i][k] = threadIdx.x * 7 + i + k; for (int k = 0; k < 16; k++) m[i][k] = blockIdx.x + k * 3 + i; }
    for (int it = 0; it < iters; it++) {
        b3::compress_n<N>(cv, m, it, 64, 0);
        for (int i = 0; i < N; i++) m[i][it & 15] ^= cv[i][0];
    }
    uint32_t acc = 0;
    for (int i = 0; i < N; i++) for (int k = 0; k < 8; k++) acc ^= cv[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int N> void run(const char* name, int blocks) {
    uint32_t* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 64;
    hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double comp = (double)blocks * 256 * N * iters;
    printf("%s blocks=%d: %.3f ms, %.3g compress/s\n", name, blocks, ms, comp / (ms * 1e-3));
    hipFree(d);
}
int main() {
    run<1>("N=1", 8192); run<2>("N=2", 8192); run<4>("N=4", 4096); run<1>("N=1", 16384);
    return 0;
}
