// VALU issue rate versus waves per SIMD on gfx950: can ONE wave per SIMD keep the integer VALU busy?
// (decides whether the bitsliced-AES kernel can run at half its register footprint and share each CU
// with the memory-bound interpreter).  Occupancy is pinned with dynamic LDS: 256-thread workgroups,
// LDS bytes chosen so that exactly w workgroups fit per CU  =>  w waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 occ_mb.hip -o occ_mb.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP>
__device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, uint32_t c) {
    if (OP == 0) { uint32_t r; asm("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
template <int OP, int ILP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    extern __shared__ uint32_t lds[];
    uint32_t x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = threadIdx.x * (i + 3) + blockIdx.x;
    const uint32_t y = threadIdx.x ^ 0x1234567u, z = blockIdx.x * 77u + 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 128 / ILP; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) x[i] = f<OP>(x[i], y + r, z);
    }
    uint32_t acc = 0;
    for (int i = 0; i < ILP; i++) acc ^= x[i];
    if (acc == 0x12345) lds[threadIdx.x] = acc;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int OP, int ILP> void run(const char* name, int w) {
    const int blocks = 256 * w * 4, iters = 2000;
    const size_t lds = (size_t)(160 * 1024 / w) - 1024;
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipFuncSetAttribute((const void*)k<OP, ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), lds, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), lds, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 256 * iters * 128;
    printf("%-14s ilp=%d waves/SIMD=%d  %.3g lane-ops/s  (%.1f lanes/clk/SIMD at 2.4 GHz)  err=%d\n", name, ILP, w, ops / (ms * 1e-3),
           ops / (ms * 1e-3) / (1024 * 2.4e9), (int)hipGetLastError());
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 3, 4, 8}) {
        run<0, 8>("v_xor_b32", w);
        run<0, 2>("v_xor_b32", w);
        run<0, 1>("v_xor_b32", w);
        run<1, 8>("v_bitop3_b32", w);
        run<1, 1>("v_bitop3_b32", w);
    }
    return 0;
}
