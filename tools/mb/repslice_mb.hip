// Rep-sliced interpreter probe (VERDICT r1 item 3 / north_star "wire values staged in LDS"):
// one workgroup = ONE repetition (8 players = one byte per wire), the live layer of the layered 10^7-gate
// workload (65 536 wires) resident in LDS as two ping-pong byte arrays (2 x 64 KiB), gate records streamed
// (all 256 workgroups read the same records -> L2 / Infinity Cache), PRG masks and transcripts in REP-MAJOR
// byte layout ([rep][mask index], [rep][event]), streamed coalesced.  The corr bit of a wire is not stored:
// the prover knows the cleartext value v of every wire (the same for all repetitions), so
// c^ = v ^ parity(mask); v arrives as 2 bits per gate from a per-proof side array.
//   level: ANDs first (nA, multiple of 4), then XORs; dst slot = position in the level.
//   lane handles 4 consecutive gates per step: 8 random ds_read_u8, one ds_write_b32.
// Variants: 0 full; 1 sequential (conflict-free) LDS reads; 2 no mask/transcript streams (LDS + records only);
//           3 no gate records (a,b from a hash of the index): LDS random + streams
// Build: hipcc --offload-arch=gfx950 -O3 repslice_mb.hip -o repslice_mb.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr uint32_t W = 65536, L = 153, NA = W / 2;

__device__ __forceinline__ uint32_t par8(uint32_t x) { return __builtin_popcount(x) & 1u; }

template <int T, int VAR>
__global__ __launch_bounds__(T) void k_rep(const uint2* __restrict__ recs /*[L][W] (a,b)*/, const uint8_t* __restrict__ vbits /*[L][W/4]*/,
                                           const uint8_t* __restrict__ masks /*[R][L*NA*2]*/, uint8_t* __restrict__ on /*[R][L*NA]*/,
                                           uint8_t* __restrict__ pre /*[R][L*NA/8]*/, uint32_t* __restrict__ sink) {
    extern __shared__ uint8_t lds[];  // 2 x W
    const uint32_t rep = blockIdx.x, t = threadIdx.x;
    uint8_t* cur = lds;
    uint8_t* nxt = lds + W;
    for (uint32_t i = t; i < W / 4; i += T) ((uint32_t*)cur)[i] = i * 2654435761u + rep;
    __syncthreads();
    const uint8_t* mrep = masks + (size_t)rep * L * NA * 2;
    uint8_t* onrep = on + (size_t)rep * L * NA;
    uint8_t* prerep = pre + (size_t)rep * (L * NA / 8);
    uint32_t acc = 0;
    for (uint32_t l = 0; l < L; l++) {
        const uint2* rl = recs + (size_t)l * W;
        const uint8_t* vl = vbits + (size_t)l * (W / 4);
        // ANDs: chunks of 4
        for (uint32_t c = t; c < NA / 4; c += T) {
            uint2 r[4];
            if (VAR == 3) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t h = (4 * c + k + l) * 2654435761u;
                    r[k] = make_uint2(h >> 16, h & 0xFFFFu);
                }
            } else {
                const uint4* rp = (const uint4*)(rl + 4 * c);
                const uint4 x0 = rp[0], x1 = rp[1];
                r[0] = make_uint2(x0.x, x0.y), r[1] = make_uint2(x0.z, x0.w), r[2] = make_uint2(x1.x, x1.y), r[3] = make_uint2(x1.z, x1.w);
            }
            const uint32_t vb = vl[c];
            uint2 mk = make_uint2(0x12345678u + c, 0x9abcdef0u ^ c);
            if (VAR != 2) mk = *(const uint2*)(mrep + ((size_t)l * NA + 4 * c) * 2);
            uint32_t ma[4], mb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t ia = VAR == 1 ? (4 * c + k) : r[k].x, ib = VAR == 1 ? ((4 * c + k + 64) & (W - 1)) : r[k].y;
                ma[k] = cur[ia & (W - 1)];
                mb[k] = cur[ib & (W - 1)];
            }
            uint32_t s4 = 0, d4 = 0, new4 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t lab = (k < 2 ? mk.x >> (16 * k) : mk.y >> (16 * (k - 2))) & 0xFFu;
                const uint32_t lnew = ((k < 2 ? mk.x >> (16 * k) : mk.y >> (16 * (k - 2))) >> 8) & 0xFFu;
                const uint32_t ra = par8(ma[k]), rb = par8(mb[k]), rab = par8(lab);
                const uint32_t ca = ((vb >> (2 * k)) & 1u) ^ ra, cb = ((vb >> (2 * k + 1)) & 1u) ^ rb;
                const uint32_t delta = (ra & rb) ^ rab;
                const uint32_t s = (mb[k] & (0u - ca)) ^ (ma[k] & (0u - cb)) ^ lab ^ lnew;
                s4 |= (s & 0xFFu) << (8 * k);
                d4 |= delta << k;
                new4 |= lnew << (8 * k);
            }
            ((uint32_t*)nxt)[c] = new4;
            if (VAR != 2) {
                *(uint32_t*)(onrep + (size_t)l * NA + 4 * c) = s4;
                const uint32_t other = __shfl_xor(d4, 1);
                if (!(t & 1)) prerep[((size_t)l * NA + 4 * c) / 8] = (uint8_t)(d4 | (other << 4));
            } else {
                acc ^= s4 + d4;
            }
        }
        // XORs
        for (uint32_t c = NA / 4 + t; c < W / 4; c += T) {
            uint2 r[4];
            if (VAR == 3) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t h = (4 * c + k + l) * 2654435761u;
                    r[k] = make_uint2(h >> 16, h & 0xFFFFu);
                }
            } else {
                const uint4* rp = (const uint4*)(rl + 4 * c);
                const uint4 x0 = rp[0], x1 = rp[1];
                r[0] = make_uint2(x0.x, x0.y), r[1] = make_uint2(x0.z, x0.w), r[2] = make_uint2(x1.x, x1.y), r[3] = make_uint2(x1.z, x1.w);
            }
            uint32_t new4 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t ia = VAR == 1 ? (4 * c + k) : r[k].x, ib = VAR == 1 ? ((4 * c + k + 64) & (W - 1)) : r[k].y;
                new4 |= (uint32_t)(cur[ia & (W - 1)] ^ cur[ib & (W - 1)]) << (8 * k);
            }
            ((uint32_t*)nxt)[c] = new4;
        }
        __syncthreads();
        uint8_t* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    acc ^= cur[t];
    if (acc == 0x7fffffffu) sink[rep] = acc;
}

template <int T, int VAR>
static void run(const char* name, const uint2* recs, const uint8_t* vb, const uint8_t* masks, uint8_t* on, uint8_t* pre, uint32_t* sink,
                int reps) {
    (void)hipFuncSetAttribute((const void*)k_rep<T, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_rep<T, VAR>), dim3(reps), dim3(T), 2 * W, 0, recs, vb, masks, on, pre, sink);
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_rep<T, VAR>), dim3(reps), dim3(T), 2 * W, 0, recs, vb, masks, on, pre, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError();
    printf("%-46s T=%4d reps=%3d  %.3f ms  (%s)\n", name, T, reps, ms / 3, hipGetErrorString(e));
}

int main() {
    const size_t n_rec = (size_t)L * W;
    std::vector<uint2> h(n_rec);
    uint64_t s = 0x5EED000000000004ull;
    for (size_t i = 0; i < n_rec; i++) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        h[i] = make_uint2((uint32_t)z & 0xFFFF, (uint32_t)(z >> 32) & 0xFFFF);
    }
    uint2* recs;
    uint8_t *vb, *masks, *on, *pre;
    uint32_t* sink;
    const int R = 256;
    (void)hipMalloc(&recs, n_rec * 8);
    (void)hipMemcpy(recs, h.data(), n_rec * 8, hipMemcpyHostToDevice);
    (void)hipMalloc(&vb, n_rec / 4);
    (void)hipMemset(vb, 0x5a, n_rec / 4);
    (void)hipMalloc(&masks, (size_t)R * L * NA * 2);
    (void)hipMemset(masks, 0x3c, (size_t)R * L * NA * 2);
    (void)hipMalloc(&on, (size_t)R * L * NA);
    (void)hipMalloc(&pre, (size_t)R * L * NA / 8);
    (void)hipMalloc(&sink, 4096);
    run<1024, 0>("full", recs, vb, masks, on, pre, sink, R);
    run<512, 0>("full", recs, vb, masks, on, pre, sink, R);
    run<1024, 1>("sequential LDS reads", recs, vb, masks, on, pre, sink, R);
    run<1024, 2>("no mask / transcript streams", recs, vb, masks, on, pre, sink, R);
    run<1024, 3>("no gate records", recs, vb, masks, on, pre, sink, R);
    run<1024, 0>("full, 32 reps (one GPU of eight)", recs, vb, masks, on, pre, sink, 32);
    return 0;
}
