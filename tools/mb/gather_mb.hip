// Gather-pattern calibration for the interpreter (gfx950): how fast can "gates" that XOR two random
// operand rows of the previous layer and write one row run, as a function of
//   (a) the layout: WIDE = one 256-B row per gate (64 quads contiguous, today's layout) versus
//       SLICED = 8 independent 32-repetition slices, slice s handled only by workgroups with
//       blockIdx % 8 == s (round-robin workgroup -> XCD dispatch), 32 B per gate per slice;
//   (b) the operand window (rows of the "previous layer"): 65536 rows (16.8 MB wide / 2.1 MB per slice,
//       fits one XCD's 4 MB L2) versus 4M rows (1 GB: HBM).
// Each "gate" = 16-B record {a, b, dst, pad} + 2 row reads + 1 row write (+ optional streaming read of 2 more
// rows, like lambda_ab / lambda_new, and a streaming write, like the online transcript row).
// Build: hipcc --offload-arch=gfx950 -O3 gather_mb.hip -o gather_mb.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Rec { uint32_t a, b, dst, m; };

// WIDE: wave = 1 gate x 64 quads, U gates in flight
template <int U, bool STREAM>
__global__ __launch_bounds__(256) void k_wide(const Rec* __restrict__ recs, uint32_t n, const uint32_t* __restrict__ win,
                                              const uint32_t* __restrict__ masks, uint32_t* __restrict__ out, uint32_t* __restrict__ on) {
    const uint32_t q = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t g0 = wave * U; g0 + U <= n; g0 += n_waves * U) {
        Rec r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = recs[g0 + u];
        uint32_t x[U], y[U], s0[U], s1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            x[u] = win[(size_t)r[u].a * 64 + q];
            y[u] = win[(size_t)r[u].b * 64 + q];
            if (STREAM) {
                s0[u] = __builtin_nontemporal_load(&masks[(size_t)r[u].m * 64 + q]);
                s1[u] = __builtin_nontemporal_load(&masks[(size_t)(r[u].m + 1) * 64 + q]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t v = x[u] ^ y[u];
            if (STREAM) {
                v ^= s0[u] & s1[u];
                __builtin_nontemporal_store(v ^ s0[u], &on[(size_t)(g0 + u) * 64 + q]);
            }
            out[(size_t)r[u].dst * 64 + q] = v;
        }
    }
}

// SLICED: layout [8][rows][8 words]; workgroup slice = blockIdx % 8; wave = 8 gates x 8 quads, U steps in flight
template <int U, bool STREAM>
__global__ __launch_bounds__(256) void k_sliced(const Rec* __restrict__ recs, uint32_t n, const uint32_t* __restrict__ win,
                                                uint32_t win_rows, const uint32_t* __restrict__ masks, uint32_t mask_rows,
                                                uint32_t* __restrict__ out, uint32_t out_rows, uint32_t* __restrict__ on) {
    const uint32_t slice = blockIdx.x & 7;
    const uint32_t lane = threadIdx.x & 63, q = lane & 7, sub = lane >> 3;
    const uint32_t wave = (blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = (gridDim.x >> 3) * (blockDim.x >> 6);
    const uint32_t* w = win + (size_t)slice * win_rows * 8;
    const uint32_t* mk = masks + (size_t)slice * mask_rows * 8;
    uint32_t* o = out + (size_t)slice * out_rows * 8;
    uint32_t* onw = on + (size_t)slice * n * 8;
    for (uint32_t g0 = wave * U * 8; g0 + U * 8 <= n; g0 += n_waves * U * 8) {
        Rec r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = recs[g0 + u * 8 + sub];
        uint32_t x[U], y[U], s0[U], s1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            x[u] = w[(size_t)r[u].a * 8 + q];
            y[u] = w[(size_t)r[u].b * 8 + q];
            if (STREAM) {
                s0[u] = __builtin_nontemporal_load(&mk[(size_t)r[u].m * 8 + q]);
                s1[u] = __builtin_nontemporal_load(&mk[(size_t)(r[u].m + 1) * 8 + q]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t v = x[u] ^ y[u];
            if (STREAM) {
                v ^= s0[u] & s1[u];
                __builtin_nontemporal_store(v ^ s0[u], &onw[(size_t)(g0 + u * 8 + sub) * 8 + q]);
            }
            o[(size_t)r[u].dst * 8 + q] = v;
        }
    }
}


// QUARTER: the access pattern a fused AES + Mul kernel would have: a 512-thread workgroup owns 16 quads
// (64 B of every row) and 88 KiB of LDS (so ONE workgroup = 8 wavefronts per CU); lane = (gate sub-index 0..3,
// quad 0..15); per step a lane has U gates in flight: 2 operand gathers of 64-byte segments each, then a
// 64-byte segment of the online row and of the output row written.  No streaming mask reads (they would be
// in registers).  Records come through LDS-free vector loads (16 B per gate, 16 lanes share one).
template <int U>
__global__ __launch_bounds__(512) void k_quarter(const Rec* __restrict__ recs, uint32_t n, const uint32_t* __restrict__ win,
                                                 uint32_t* __restrict__ out, uint32_t* __restrict__ on) {
    extern __shared__ uint32_t lds[];
    const uint32_t qg = blockIdx.x & 3, chunk = blockIdx.x >> 2, n_chunks = gridDim.x >> 2;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = qg * 16 + (lane & 15), sub = lane >> 4;
    if (lane == 999) lds[threadIdx.x] = n;
    const uint32_t per = (n + n_chunks - 1) / n_chunks;
    const uint32_t lo = chunk * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t g0 = lo + (wave * 4 + sub) * U; g0 + U <= hi; g0 += 32 * U) {
        Rec r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = recs[g0 + u];
        uint32_t x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            x[u] = win[(size_t)r[u].a * 64 + q];
            y[u] = win[(size_t)r[u].b * 64 + q];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t v = x[u] & y[u];
            on[(size_t)(g0 + u) * 64 + q] = v ^ r[u].m;
            out[(size_t)r[u].dst * 64 + q] = v;
        }
    }
}

static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    const uint32_t n = 1u << 21;  // gates per launch
    for (int stream = 0; stream < 2; stream++)
        for (uint32_t win_rows : {65536u, 1u << 22}) {
            std::vector<Rec> recs(n);
            uint64_t s = 42;
            for (uint32_t i = 0; i < n; i++) recs[i] = Rec{(uint32_t)(sm(s) % win_rows), (uint32_t)(sm(s) % win_rows), i, 2 * i};
            Rec* d_recs; uint32_t *d_win, *d_masks, *d_out, *d_on;
            CK(hipMalloc(&d_recs, n * sizeof(Rec)));
            CK(hipMalloc(&d_win, (size_t)win_rows * 256));
            CK(hipMalloc(&d_masks, (size_t)(2 * n + 2) * 256));
            CK(hipMalloc(&d_out, (size_t)n * 256));
            CK(hipMalloc(&d_on, (size_t)n * 256));
            CK(hipMemcpy(d_recs, recs.data(), n * sizeof(Rec), hipMemcpyHostToDevice));
            CK(hipMemset(d_win, 1, (size_t)win_rows * 256));
            CK(hipMemset(d_masks, 2, (size_t)(2 * n + 2) * 256));
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            const double bytes = (double)n * (16 + 3 * 256 + (stream ? 3 * 256 : 0));
            for (int variant = 0; variant < 2; variant++)
                for (int blocks : {2048, 4096, 8192}) {
                    float best = 1e9f;
                    for (int rep = 0; rep < 4; rep++) {
                        hipEventRecord(a);
                        if (variant == 0) {
                            if (stream) hipLaunchKernelGGL((k_wide<4, true>), dim3(blocks), dim3(256), 0, 0, d_recs, n, d_win, d_masks, d_out, d_on);
                            else hipLaunchKernelGGL((k_wide<4, false>), dim3(blocks), dim3(256), 0, 0, d_recs, n, d_win, d_masks, d_out, d_on);
                        } else {
                            if (stream) hipLaunchKernelGGL((k_sliced<4, true>), dim3(blocks), dim3(256), 0, 0, d_recs, n, d_win, win_rows, d_masks, 2 * n + 2, d_out, n, d_on);
                            else hipLaunchKernelGGL((k_sliced<4, false>), dim3(blocks), dim3(256), 0, 0, d_recs, n, d_win, win_rows, d_masks, 2 * n + 2, d_out, n, d_on);
                        }
                        hipEventRecord(b); hipEventSynchronize(b);
                        float ms; hipEventElapsedTime(&ms, a, b);
                        if (ms < best) best = ms;
                    }
                    CK(hipGetLastError());
                    printf("%-6s stream=%d window=%8u rows blocks=%5d  %.3f ms  %.2f ns/gate... %.2f TB/s algorithmic (%s)\n", variant ? "SLICED" : "WIDE", stream,
                           win_rows, blocks, best, best * 1e6 / n, bytes / (best * 1e-3) / 1e12, variant ? "slice gathers 8x the records" : "");
                }
            if (!stream) {
                hipFuncSetAttribute((const void*)k_quarter<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 88 * 1024);
                hipFuncSetAttribute((const void*)k_quarter<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 88 * 1024);
                for (int U : {8, 16})
                    for (int blocks : {256 * 4, 512 * 4}) {
                        float best = 1e9f;
                        for (int rep = 0; rep < 4; rep++) {
                            hipEventRecord(a);
                            if (U == 8) hipLaunchKernelGGL((k_quarter<8>), dim3(blocks), dim3(512), 88 * 1024, 0, d_recs, n, d_win, d_out, d_on);
                            else hipLaunchKernelGGL((k_quarter<16>), dim3(blocks), dim3(512), 88 * 1024, 0, d_recs, n, d_win, d_out, d_on);
                            hipEventRecord(b); hipEventSynchronize(b);
                            float ms; hipEventElapsedTime(&ms, a, b);
                            if (ms < best) best = ms;
                        }
                        CK(hipGetLastError());
                        const double qb = (double)n * (16 + 4 * 256);
                        printf("QUARTER U=%2d window=%8u rows blocks=%5d  %.3f ms  %.3f ns/gate  %.2f TB/s algorithmic (2 gathers + 2 row writes, 8 waves/CU)\n", U, win_rows,
                               blocks, best, best * 1e6 / n, qb / (best * 1e-3) / 1e12);
                    }
            }
            hipFree(d_recs); hipFree(d_win); hipFree(d_masks); hipFree(d_out); hipFree(d_on);
        }
    return 0;
}
