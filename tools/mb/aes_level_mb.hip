// what does a LEVEL cost when the lane-distributed cipher (aes_col4_dev.h) runs inside the level launches?
// A level of the benchmark circuit has 32 768 Mul gates = 512 CTR blocks x 4 quad groups = 2 048 wavefront trips = 2 per SIMD.
// This microbenchmark launches the product's cipher body that way -- N dependent launches of 256 workgroups x 8 wavefronts, ONE trip
// per wavefront, the 88 KiB key image staged per launch -- and, optionally, the memory side of a fused Mul step around it:
// G random row gathers per gate (64-byte segments per 16 lanes out of a 16.8 MB window, issued BEFORE the cipher), the online row
// and lambda_new stored after it.  Timing only (random key image; the cipher's bytes are checked by the product's tests).
//   variants (argv[1]): 0 = stage through registers, everything before the first round (the shipped generator's prologue)
//                       1 = LDS-DMA (global_load_lds_dwordx4), areas 0-1 waited for first, the rest before round 2
//   argv[2] = stores: 0 none, 1 both mask rows (the shipped generator), 2 fused (online row + lambda_new per gate)
//   argv[3] = G gathers per gate (0, 2 .. 6)        argv[4] = levels (153)        argv[5] = 1: ONE launch, `levels` trips per wavefront
//   argv[6] = workgroups per launch (256 = one trip per wavefront slot at 2 per SIMD)
// Build: hipcc --offload-arch=gfx950 -O3 -I reverie_amd/csrc tools/mb/aes_level_mb.hip -o tools/mb/aes_level_mb.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "aes_col4_dev.h"
using namespace rv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int WAVES = 8;
constexpr uint32_t NQ = 64;

template <int STAGE, int STORES, int G, bool PERSIST>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_level(const uint4* __restrict__ img, uint32_t first_block, uint32_t trips,
                                                                                                 const uint32_t* __restrict__ rows, uint32_t window,
                                                                                                 uint32_t* __restrict__ out_rows, uint32_t* __restrict__ on) {
    extern __shared__ uint4 lds[];
    const uint32_t qg = blockIdx.x & 3, chunk = blockIdx.x >> 2;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* src = img + (size_t)qg * C4_IMG_U4;
    if (STAGE == 0) {
        constexpr uint32_t T = WAVES * 64, FULL = C4_IMG_U4 / T;
        uint4 v[FULL];
#pragma unroll
        for (uint32_t i = 0; i < FULL; i++) v[i] = src[threadIdx.x + i * T];
#pragma unroll
        for (uint32_t i = 0; i < FULL; i++) lds[threadIdx.x + i * T] = v[i];
        __syncthreads();
    } else {
        // 88 pieces of 1 KiB (one (area, plane) row of 64 lanes x 16 B each): wavefront w takes pieces w, w + 8, ...; pieces 0..15 = areas 0, 1
#pragma unroll
        for (uint32_t i = 0; i < 11; i++) {
            const uint32_t piece = wave + i * WAVES;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + piece * 64 + lane),
                                             (void __attribute__((address_space(3)))*)(lds + piece * 64), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // this wavefront's pieces of areas 0, 1 (its first two)
        __builtin_amdgcn_s_barrier();
    }
    const uint32_t c = lane & 3;
    const uint4* rkl = lds + lane;
    const uint32_t cs = lane >> 4, ql = lane & 15, qs = qg * 16 + ql;
    const uint32_t from = 4 * (4 * ql + cs);
    for (uint32_t t = 0; t < trips; t++) {
        const uint32_t jl = (chunk * WAVES + wave) + t * (gridDim.x / 4) * WAVES;
        const uint32_t j = first_block + jl;
        // the fused Mul step's operand rows: gate i = 16*cs + e of the block, G rows each, this lane's quad word
        uint32_t opnd[16][G > 0 ? G : 1];
        if (G > 0) {
#pragma unroll
            for (int e = 0; e < 16; e++)
#pragma unroll
                for (int g = 0; g < G; g++) {
                    uint32_t h = (j * 64u + 16u * cs + (uint32_t)e) * 2654435761u + (uint32_t)g * 0x9E3779B9u;
                    h ^= h >> 15;
                    h *= 0x85EBCA6Bu;
                    h ^= h >> 13;
                    opnd[e][g] = rows[(size_t)(h % window) * NQ + qs];
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t s[32];
        c4_rounds_0_1(j, c, rkl, s);
        if (STAGE == 1 && t == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll 1
        for (int r = 2; r < 10; r++) c4_round(s, rkl + r * 8 * 64);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            c4_sbox8(s[8 * r + 7], s[8 * r + 6], s[8 * r + 5], s[8 * r + 4], s[8 * r + 3], s[8 * r + 2], s[8 * r + 1], s[8 * r + 0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t o[32];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 kv = rkl[(10 * 8 + k) * 64];
            const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
            for (int r = 0; r < 4; r++) o[8 * r + (7 - k)] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)(s[8 * r + k] ^ kw[r]));
        }
        // o[m] = mask 32*cs + m of the block for quad qs: gate e of this lane's sixteen has lambda_ab = o[2e], lambda_new = o[2e + 1]
        if (STORES == 1) {
            uint32_t* out = out_rows + ((size_t)j * 128 + 32 * cs) * NQ + qs;
#pragma unroll
            for (int m = 0; m < 32; m++) __builtin_nontemporal_store(o[m], &out[(size_t)m * NQ]);
        } else if (STORES == 2) {
            uint32_t* out = out_rows + ((size_t)j * 64 + 16 * cs) * NQ + qs;
            uint32_t* onp = on + ((size_t)j * 64 + 16 * cs) * NQ + qs;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t lx = 0, ly = 0;
#pragma unroll
                for (int g = 0; g < (G > 0 ? G : 1); g++) {
                    if (g & 1) ly ^= G > 0 ? opnd[e][g] : 0u;
                    else lx ^= G > 0 ? opnd[e][g] : 0u;
                }
                // (a stand-in for the Mul arithmetic: three reconstructions and the share)
                uint32_t a = lx ^ (lx >> 4), b = ly ^ (ly >> 4), cc = o[2 * e] ^ (o[2 * e] >> 4);
                a ^= a >> 2, b ^= b >> 2, cc ^= cc >> 2;
                a ^= a >> 1, b ^= b >> 1, cc ^= cc >> 1;
                a = (a & 0x01010101u) * 0xFFu, b = (b & 0x01010101u) * 0xFFu, cc = (cc & 0x01010101u) * 0xFFu;
                const uint32_t sv = (ly & a) ^ (lx & b) ^ o[2 * e] ^ o[2 * e + 1] ^ ((a & b) ^ cc);
                __builtin_nontemporal_store(sv, &onp[(size_t)e * NQ]);
                out[(size_t)e * NQ] = o[2 * e + 1];
            }
        } else {
            uint32_t acc = 0;
#pragma unroll
            for (int m = 0; m < 32; m++) acc |= o[m];
            if (G > 0)
#pragma unroll
                for (int e = 0; e < 16; e++)
#pragma unroll
                    for (int g = 0; g < G; g++) acc |= opnd[e][g];
            if (acc == 0x12345678u) out_rows[0] = acc;
        }
    }
}

static int g_grid = 256;
template <int STAGE, int STORES, int G>
static float run(bool persist, int levels, const uint4* d_img, const uint32_t* d_rows, uint32_t window, uint32_t* d_out, uint32_t* d_on) {
    auto kern = persist ? k_level<STAGE, STORES, G, true> : k_level<STAGE, STORES, G, false>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4_LDS_BYTES));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)kern));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        CK(hipEventRecord(e0));
        if (persist) {
            hipLaunchKernelGGL(kern, dim3(g_grid), dim3(WAVES * 64), C4_LDS_BYTES, 0, d_img, 0u, (uint32_t)levels, d_rows, window, d_out, d_on);
        } else {
            for (int l = 0; l < levels; l++)
                hipLaunchKernelGGL(kern, dim3(g_grid), dim3(WAVES * 64), C4_LDS_BYTES, 0, d_img, (uint32_t)l * (uint32_t)(g_grid / 4 * WAVES), 1u, d_rows, window, d_out, d_on);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    printf("stage %d stores %d gathers %d %s: %d VGPRs, %zu B scratch: %.3f ms for %d levels = %.2f us per level\n", STAGE, STORES, G,
           persist ? "ONE launch" : "launch per level", fa.numRegs, (size_t)fa.localSizeBytes, best, levels, 1e3 * best / levels);
    return best;
}

int main(int argc, char** argv) {
    const int stage = argc > 1 ? atoi(argv[1]) : 0, stores = argc > 2 ? atoi(argv[2]) : 0, G = argc > 3 ? atoi(argv[3]) : 0;
    const int levels = argc > 4 ? atoi(argv[4]) : 153;
    const bool persist = argc > 5 && atoi(argv[5]);
    g_grid = argc > 6 ? atoi(argv[6]) : 256;  // 256 workgroups x 8 wavefronts = 2 048 trips = 512 blocks x 4 quad groups per launch
    std::vector<uint32_t> img((size_t)4 * C4_IMG_U4 * 4);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& w : img) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)(x >> 24); }
    uint4* d_img;
    CK(hipMalloc(&d_img, img.size() * 4));
    CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    const uint32_t window = 65536;
    uint32_t *d_rows, *d_out, *d_on;
    CK(hipMalloc(&d_rows, (size_t)window * NQ * 4));
    CK(hipMemset(d_rows, 0x5a, (size_t)window * NQ * 4));
    const size_t out_rows = (size_t)(levels + 1) * (size_t)(g_grid / 4 * WAVES) * 128;  // (blocks x 128 rows)
    CK(hipMalloc(&d_out, out_rows * NQ * 4));
    CK(hipMalloc(&d_on, out_rows / 2 * NQ * 4));
#define RUN(ST, SR, GG) if (stage == ST && stores == SR && G == GG) run<ST, SR, GG>(persist, levels, d_img, d_rows, window, d_out, d_on);
#define RUNS(GG) RUN(0, 0, GG) RUN(0, 1, GG) RUN(0, 2, GG) RUN(1, 0, GG) RUN(1, 1, GG) RUN(1, 2, GG)
    RUNS(0) RUNS(2) RUNS(3) RUNS(4) RUNS(6)
    return 0;
}
