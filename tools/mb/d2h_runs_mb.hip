// Micro-benchmark: the opening kernels' write pattern straight into page-locked host memory.  A workgroup owns output
// bytes [t0, t0 + run) of each of 40 vectors at byte-granular addresses (record stride odd, as in the proof) and writes
// them either bytewise (a lane per byte, what ex_flush did) or as an aligned 16-byte body with bytewise head / tail.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mb/d2h_runs_mb.hip -o tools/mb/d2h_runs_mb.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr size_t REC = 1254903, VEC = 626774, BASE = 33000 + 137;

template <int WIDE>
__global__ __launch_bounds__(256) void k_runs(uint8_t* __restrict__ out, uint32_t run, const uint8_t* __restrict__ src) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_buf[];  // [40][run + 16]
    const uint32_t stride = run + 16;
    const size_t t0 = (size_t)blockIdx.x * run;
    const uint32_t nb = (uint32_t)(VEC - t0 < run ? VEC - t0 : run);
    // stage something that depends on device memory (so the kernel is not write-only)
    for (uint32_t i = threadIdx.x; i < 40 * stride; i += 256) s_buf[i] = src[(t0 + i) & 0xFFFFF];
    __syncthreads();
    if (!WIDE) {
        for (uint32_t idx = threadIdx.x; idx < 40 * run; idx += 256) {
            const uint32_t k = idx / run, i = idx % run;
            if (i < nb) out[BASE + k * REC + t0 + i] = s_buf[k * stride + i];
        }
    } else {
        // s_buf row k holds the run shifted by the destination's misalignment: LDS byte j <-> address (dst & ~15) + j
        for (uint32_t k = threadIdx.x >> 6; k < 40; k += 4) {
            const size_t dst = BASE + k * REC + t0;
            const uint32_t mis = (uint32_t)(dst & 15);
            uint8_t* basep = out + (dst - mis);
            const uint32_t lane = threadIdx.x & 63;
            const uint32_t end = mis + nb;  // bytes [mis, end) of the shifted row are valid
            for (uint32_t j = lane * 16; j < end; j += 64 * 16) {
                if (j >= mis && j + 16 <= end) {
                    *(v4u*)(basep + j) = *(const v4u*)(s_buf + k * stride + j);
                } else {
                    for (uint32_t b = 0; b < 16; b++)
                        if (j + b >= mis && j + b < end) basep[j + b] = s_buf[k * stride + j + b];
                }
            }
        }
    }
}
int main() {
    const size_t total = 50196120 + (1 << 20);
    uint8_t *d, *h, *src;
    CK(hipMalloc(&d, total));
    CK(hipMalloc(&src, 2 << 20));
    CK(hipMemset(src, 3, 2 << 20));
    CK(hipHostMalloc(&h, total, hipHostMallocDefault));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipFuncSetAttribute((const void*)k_runs<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10));
    CK(hipFuncSetAttribute((const void*)k_runs<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10));
    for (int host = 0; host < 2; host++)
        for (uint32_t run : {128u, 256u, 512u, 1024u, 2048u})
            for (int wide = 0; wide < 2; wide++) {
                const unsigned grid = (unsigned)((VEC + run - 1) / run);
                const size_t lds = 40 * (run + 16);
                double best = 1e9;
                for (int rep = 0; rep < 4; rep++) {
                    hipStreamSynchronize(st);
                    auto t0 = std::chrono::steady_clock::now();
                    if (wide) hipLaunchKernelGGL(k_runs<1>, dim3(grid), dim3(256), lds, st, host ? h : d, run, src);
                    else hipLaunchKernelGGL(k_runs<0>, dim3(grid), dim3(256), lds, st, host ? h : d, run, src);
                    hipStreamSynchronize(st);
                    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (rep && ms < best) best = ms;
                }
                printf("%s  run %4u B  %-8s grid %5u: %.3f ms  %.1f GB/s\n", host ? "host  " : "device", run, wide ? "16-byte" : "bytewise", grid, best, 40.0 * VEC / best / 1e6);
                fflush(stdout);
            }
    return 0;
}
