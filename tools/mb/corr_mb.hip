// Does it pay to keep a row's correction bits (32 B) next to its 256-B share row instead of in a separate array?
// The interpreter's AND gate gathers two operand rows AND their 32-byte corr-bit rows (four random accesses), reads
// two streaming mask rows, writes an online row, an output corr row and preprocessing bits.  Variants:
//   NONE    no corr accesses at all (= gather_mb WIDE stream): the ceiling
//   SPLIT   corr in its own array [rows][32 B] (today's layout)
//   SCALAR  today's layout, the 32-byte corr rows fetched with scalar loads (the gate is wave-uniform)
//   JOINED  row stride 288 / 320 bytes, corr bits at byte 256 of the row: the operand gather and its corr read hit
//           the same DRAM page / adjacent sectors
// Build: hipcc --offload-arch=gfx950 -O3 corr_mb.hip -o corr_mb.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Rec { uint32_t a, b, dst, m; };
// MODE 0 NONE, 1 SPLIT, 2 JOINED; SW = row stride in words (64 for NONE / SPLIT)
template <int U, int MODE, int SW, bool WR>
__global__ __launch_bounds__(256) void k(const Rec* __restrict__ recs, uint32_t n, uint32_t* __restrict__ win, uint8_t* __restrict__ corr,
                                         const uint32_t* __restrict__ masks, uint32_t* __restrict__ on, uint8_t* __restrict__ pre,
                                         uint32_t out_base) {
    const uint32_t q = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t g0 = wave * U; g0 + U <= n; g0 += n_waves * U) {
        Rec r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = recs[g0 + u];
        uint32_t x[U], y[U], s0[U], s1[U], cx[U], cy[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            x[u] = win[(size_t)r[u].a * SW + q];
            y[u] = win[(size_t)r[u].b * SW + q];
            cx[u] = cy[u] = 0;
            if (MODE == 1) {
                cx[u] = corr[(size_t)r[u].a * 32 + (q >> 1)];
                cy[u] = corr[(size_t)r[u].b * 32 + (q >> 1)];
            }
            if (MODE == 3) {  // SPLIT layout, but the 32-byte corr row comes through scalar loads (wave-uniform address)
                const uint32_t* ca = (const uint32_t*)(corr + (size_t)r[u].a * 32);
                const uint32_t* cb = (const uint32_t*)(corr + (size_t)r[u].b * 32);
                const bool b3 = q & 8, b4 = q & 16, b5 = q & 32;
                const uint32_t wa = b5 ? (b4 ? (b3 ? ca[7] : ca[6]) : (b3 ? ca[5] : ca[4])) : (b4 ? (b3 ? ca[3] : ca[2]) : (b3 ? ca[1] : ca[0]));
                const uint32_t wb = b5 ? (b4 ? (b3 ? cb[7] : cb[6]) : (b3 ? cb[5] : cb[4])) : (b4 ? (b3 ? cb[3] : cb[2]) : (b3 ? cb[1] : cb[0]));
                cx[u] = wa >> (4 * (q & 6));  // same byte value as the vector path: byte q>>1
                cy[u] = wb >> (4 * (q & 6));
                cx[u] &= 0xFF; cy[u] &= 0xFF;
            }
            if (MODE == 2) {
                cx[u] = ((const uint8_t*)(win + (size_t)r[u].a * SW + 64))[q >> 1];
                cy[u] = ((const uint8_t*)(win + (size_t)r[u].b * SW + 64))[q >> 1];
            }
            s0[u] = masks[(size_t)r[u].m * SW + q];
            s1[u] = masks[(size_t)(r[u].m + 1) * SW + q];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t v = (x[u] & cy[u]) ^ (y[u] & cx[u]) ^ s0[u] ^ s1[u] ^ x[u] ^ y[u];
            on[(size_t)(g0 + u) * 64 + q] = v;
            const uint32_t nb = (v >> 3) & 0xF, other = __shfl_xor(nb, 1);
            if (WR && !(q & 1)) {
                const uint8_t byte = (uint8_t)(nb | (other << 4));
                if (MODE != 0) pre[(size_t)(g0 + u) * 32 + (q >> 1)] = byte;
                // output corr of the gate: row dst = out_base + m + 1 style (every other row), sequential
                if (MODE == 1 || MODE == 3) corr[(size_t)(out_base + r[u].dst) * 32 + (q >> 1)] = byte;
                if (MODE == 2) ((uint8_t*)(win + (size_t)(out_base + r[u].dst) * SW + 64))[q >> 1] = byte;
            }
        }
    }
}
static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
template <int MODE, int SW, bool WR>
void run(const char* name, const Rec* d_recs, uint32_t n, uint32_t win_rows) {
    uint32_t *d_win, *d_masks, *d_on; uint8_t *d_corr, *d_pre;
    const size_t rows = (size_t)win_rows + n + 2;  // window, then the rows the gates' corr outputs go to
    CK(hipMalloc(&d_win, rows * SW * 4)); CK(hipMalloc(&d_corr, rows * 32)); CK(hipMalloc(&d_masks, (size_t)(2 * n + 2) * SW * 4));
    CK(hipMalloc(&d_on, (size_t)n * 256)); CK(hipMalloc(&d_pre, (size_t)n * 32));
    CK(hipMemset(d_win, 1, rows * SW * 4)); CK(hipMemset(d_corr, 1, rows * 32)); CK(hipMemset(d_masks, 2, (size_t)(2 * n + 2) * SW * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {4096, 8192}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL((k<4, MODE, SW, WR>), dim3(blocks), dim3(256), 0, 0, d_recs, n, d_win, d_corr, d_masks, d_on, d_pre, win_rows);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-22s window=%8u rows blocks=%5d  %.3f ms  %.3f ns/gate\n", name, win_rows, blocks, best, best * 1e6 / n);
    }
    hipFree(d_win); hipFree(d_corr); hipFree(d_masks); hipFree(d_on); hipFree(d_pre);
}
int main() {
    const uint32_t n = 1u << 21;
    for (uint32_t win_rows : {65536u, 1u << 21}) {
        std::vector<Rec> recs(n);
        uint64_t s = 42;
        for (uint32_t i = 0; i < n; i++) recs[i] = Rec{(uint32_t)(sm(s) % win_rows), (uint32_t)(sm(s) % win_rows), i, 2 * i};
        Rec* d_recs; CK(hipMalloc(&d_recs, n * sizeof(Rec)));
        CK(hipMemcpy(d_recs, recs.data(), n * sizeof(Rec), hipMemcpyHostToDevice));
        run<0, 64, true>("NONE", d_recs, n, win_rows);
        run<1, 64, true>("SPLIT (today)", d_recs, n, win_rows);
        run<1, 64, false>("SPLIT reads only", d_recs, n, win_rows);
        run<3, 64, true>("SPLIT scalar corr", d_recs, n, win_rows);
        run<3, 64, false>("SPLIT scalar reads only", d_recs, n, win_rows);
        run<2, 72, true>("JOINED stride 288 B", d_recs, n, win_rows);
        run<2, 72, false>("JOINED 288 reads only", d_recs, n, win_rows);
        run<2, 128, false>("JOINED 512 reads only", d_recs, n, win_rows);
        run<2, 128, true>("JOINED stride 512 B", d_recs, n, win_rows);
        hipFree(d_recs);
    }
    return 0;
}
