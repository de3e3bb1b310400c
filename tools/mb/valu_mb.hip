// Integer-VALU issue-rate calibration for gfx950: lane-ops/s of the instructions the AES and BLAKE3
// kernels are made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_mb.hip -o valu_mb.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP>
__device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, uint32_t c) {
    if (OP == 0) { uint32_t r; asm("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (OP == 1) return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    if (OP == 2) { uint32_t r; asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    if (OP == 3) { uint32_t r; asm("v_alignbit_b32 %0, %1, %1, 7" : "=v"(r) : "v"(a)); return r ^ b; }
    if (OP == 4) return __builtin_amdgcn_perm(a, b, 0x01000302u);
    if (OP == 5) { uint32_t r; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
    if (OP == 6) { uint32_t r; asm("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    return a;
}
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 3) + blockIdx.x;
    const uint32_t y = threadIdx.x ^ 0x1234567u, z = blockIdx.x * 77u + 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = f<OP>(x[i], y + r, z);
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int OP> void run(const char* name) {
    const int blocks = 256 * 8, iters = 2000;
    uint32_t* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 256 * iters * 128;
    printf("%-16s %.3g lane-ops/s  (%.1f lanes/clk/SIMD at 2.4 GHz)\n", name, ops / (ms * 1e-3), ops / (ms * 1e-3) / (1024 * 2.4e9));
    hipFree(d);
}
int main() {
    run<0>("v_xor_b32"); run<1>("v_bitop3_b32"); run<2>("v_add_u32"); run<3>("v_alignbit_b32"); run<4>("v_perm_b32");
    run<5>("v_add3_u32"); run<6>("v_lshl_or_b32");
    return 0;
}
