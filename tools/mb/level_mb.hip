// Floor of one dependency level inside a single 1024-thread workgroup (the narrow-run kernel):
//   A: LDS reads + ~120 VALU ops + LDS write + __syncthreads()           (everything forwarded through LDS)
//   B: A + one dependent global load per level (L2 / HBM round trip)      (today's per-gate kernel)
//   C: A + global stores (fire and forget) + a global load issued one level AHEAD (prefetch, consumed next level)
// Build: hipcc --offload-arch=gfx950 -O3 level_mb.hip -o level_mb.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int VARIANT>
__global__ __launch_bounds__(1024) void k(uint32_t* __restrict__ g, uint32_t n_rows, int levels, uint32_t* out) {
    __shared__ uint32_t cache[2][32][64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = lane, pre = 0;
    for (int i = threadIdx.x; i < 2 * 32 * 64; i += 1024) (&cache[0][0][0])[i] = i;
    __syncthreads();
    uint32_t row = (wave * 977u + 13u) % n_rows;
    if (VARIANT == 2) pre = g[(size_t)row * 64 + lane];
    for (int l = 0; l < levels; l++) {
        uint32_t x = cache[(l + 1) & 1][(wave + l) & 31][lane] ^ cache[(l + 1) & 1][(wave * 3 + l) & 31][lane];
        if (VARIANT == 1) x ^= g[(size_t)row * 64 + lane];
        if (VARIANT == 2) x ^= pre;
#pragma unroll
        for (int k2 = 0; k2 < 40; k2++) x = (x ^ (x >> 3)) + acc + k2;  // ~120 dependent ALU ops
        acc ^= x;
        cache[l & 1][wave][lane] = x;
        row = (row * 1103515245u + 12345u + x % 7u) % n_rows;
        if (VARIANT >= 1) g[(size_t)((row + 7) % n_rows) * 64 + lane] = x;
        if (VARIANT == 2) pre = g[(size_t)row * 64 + lane];
        __syncthreads();
    }
    out[threadIdx.x] = acc;
}
template <int V> void run(const char* name, uint32_t* d, uint32_t n_rows, uint32_t* o) {
    const int levels = 4000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(1024), 0, 0, d, n_rows, levels, o);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(1024), 0, 0, d, n_rows, levels, o);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-44s rows=%8u  %.3f us per level\n", name, n_rows, ms * 1e3 / levels);
}
int main() {
    uint32_t *d, *o;
    const uint32_t big = 1u << 17;  // 32 MB of rows (SHA-256's working set): mostly beyond the 4 MB L2
    (void)hipMalloc(&d, (size_t)big * 256); (void)hipMalloc(&o, 4096);
    (void)hipMemset(d, 1, (size_t)big * 256);
    run<0>("A: LDS forward only", d, big, o);
    for (uint32_t n : {4096u, big}) {
        run<1>("B: + dependent global load per level", d, n, o);
        run<2>("C: + global load issued one level ahead", d, n, o);
    }
    return 0;
}
