// Lane-distributed bitsliced AES-128-CTR mask generator for gfx950 ("col4"): a quad of lanes holds ONE bitsliced
// state -- lane c of the quad = state column c (4 bytes x 8 bit planes = 32 VGPRs, 32 slots per word) -- instead of a
// lane holding all 128 planes (k_aes_gf2_masks: 256 VGPRs, 2 wavefronts per SIMD, a CU per workgroup).
//   SubBytes      lane-local (the 74-op cover of aes_sbox.inc, four times)
//   MixColumns    lane-local (a column is a lane)
//   ShiftRows     folded into AddRoundKey: y = dpp_quad_perm_r(t) ^ srk, one v_xor_b32_dpp per word, round keys stored
//                 pre-shifted.  The state is kept SHIFTED (y_i = ShiftRows(x_i)); SubBytes commutes with it.
// Same lane-op count as the 128-plane form (~103 per byte and round against 101), a quarter of the registers.
// Output layout = the product's masks[(j*128 + b)*NQ + q], b = 8*byte + (7 - bit)  (gf2/domain.rs bit order).
//
// gate = at least 0.8x the rate of k_aes_gf2_masks (1.63 ms for 78 381 blocks x 64 quads) at a
// register count that lets it share a compute unit with the interpreter.
// Build: hipcc --offload-arch=gfx950 -O3 -I reverie_amd/csrc tools/mb/aes_col4_mb.hip -o tools/mb/aes_col4_mb.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef WAVES
#define WAVES 8  // wavefronts per workgroup
#endif
#ifndef MINB
#define MINB 1   // __launch_bounds__ second argument (workgroups per CU the register budget is computed for)
#endif

// ---------------- host reference ----------------
static uint8_t SB[256];
static void init_sbox() {
    uint8_t p = 1, q = 1;
    do {
        p = p ^ (uint8_t)(p << 1) ^ ((p & 0x80) ? 0x1B : 0);
        q ^= q << 1; q ^= q << 2; q ^= q << 4; q ^= (q & 0x80) ? 0x09 : 0;
        uint8_t x = q ^ (uint8_t)((q << 1) | (q >> 7)) ^ (uint8_t)((q << 2) | (q >> 6)) ^ (uint8_t)((q << 3) | (q >> 5)) ^ (uint8_t)((q << 4) | (q >> 4));
        SB[p] = x ^ 0x63;
    } while (p != 1);
    SB[0] = 0x63;
}
static uint8_t xt(uint8_t x) { return (uint8_t)((x << 1) ^ ((x & 0x80) ? 0x1b : 0)); }
static void key_expand(const uint8_t key[16], uint8_t rk[176]) {
    memcpy(rk, key, 16);
    uint8_t rcon = 1;
    for (int r = 1; r <= 10; r++) {
        const uint8_t* p = rk + 16 * (r - 1);
        uint8_t* q = rk + 16 * r;
        q[0] = p[0] ^ SB[p[13]] ^ rcon; q[1] = p[1] ^ SB[p[14]]; q[2] = p[2] ^ SB[p[15]]; q[3] = p[3] ^ SB[p[12]];
        rcon = xt(rcon);
        for (int i = 4; i < 16; i++) q[i] = p[i] ^ q[i - 4];
    }
}
static void encrypt(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int r = 1; r <= 10; r++) {
        for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) t[4 * c + row] = SB[s[4 * ((c + row) & 3) + row]];
        if (r < 10) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3], all = a0 ^ a1 ^ a2 ^ a3;
                s[4 * c + 0] = a0 ^ all ^ xt(a0 ^ a1); s[4 * c + 1] = a1 ^ all ^ xt(a1 ^ a2);
                s[4 * c + 2] = a2 ^ all ^ xt(a2 ^ a3); s[4 * c + 3] = a3 ^ all ^ xt(a3 ^ a0);
            }
        } else memcpy(s, t, 16);
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
    }
    memcpy(out, s, 16);
}

// ---------------- device ----------------
#define XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
__device__ __forceinline__ void sbox8(uint32_t& b7, uint32_t& b6, uint32_t& b5, uint32_t& b4, uint32_t& b3, uint32_t& b2, uint32_t& b1, uint32_t& b0) {
    const uint32_t U0 = b7, U1 = b6, U2 = b5, U3 = b4, U4 = b3, U5 = b2, U6 = b1, U7 = b0;
#include "aes_sbox.inc"
    b7 = S0; b6 = S1; b5 = S2; b4 = S3; b3 = S4; b2 = S5; b1 = S6; b0 = S7;
}

// y = quad_perm_R(t) ^ k: lane c of every quad reads lane (c + R) & 3 (folds into ONE v_xor_b32_dpp; the compiler knows the
// VALU-write -> DPP-read hazard, which inline assembly would hide from it)
template <int R>
__device__ __forceinline__ uint32_t shift_xor(uint32_t t, uint32_t k) {
    if (R == 0) return t ^ k;
    constexpr int ctrl = R == 1 ? 0x39 : (R == 2 ? 0x4E : 0x93);  // quad_perm:[R, R+1, R+2, R+3] mod 4
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, ctrl, 0xf, 0xf, true) ^ k;
}

// LDS round keys: uint4 at lds4[(area*8 + k)*64 + lane] = bit plane k of the lane's four bytes (rows 0..3)
// one middle round on the SHIFTED state: s = ShiftRows(MixColumns(SubBytes(s)) ^ rk), rk pre-shifted.  MixColumns runs plane
// by plane over the four rows at once, in place: out_r[k] = d_r[k-1] ^ all[k] ^ a_r[k] (^ d_r[7] for k = 1, 3, 4; d_r[-1] = d_r[7])
// with d_r = a_r ^ a_(r+1), all = a_0 ^ a_1 ^ a_2 ^ a_3 -- beside the state only d[7], d[k-1], d[k] and four key words are live.
__device__ __forceinline__ void round_col4(uint32_t* s, const uint4* rk4 /* + area*8*64 + lane */) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        sbox8(s[8 * r + 7], s[8 * r + 6], s[8 * r + 5], s[8 * r + 4], s[8 * r + 3], s[8 * r + 2], s[8 * r + 1], s[8 * r + 0]);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t d7[4], prev[4];
#pragma unroll
    for (int r = 0; r < 4; r++) prev[r] = d7[r] = s[8 * r + 7] ^ s[8 * ((r + 1) & 3) + 7];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint4 kv = rk4[k * 64];
        const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w};
        uint32_t cur[4];
#pragma unroll
        for (int r = 0; r < 4; r++) cur[r] = k == 7 ? d7[r] : (s[8 * r + k] ^ s[8 * ((r + 1) & 3) + k]);
        const uint32_t all = cur[0] ^ cur[2];
        uint32_t t[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            t[r] = XOR3(prev[r], all, s[8 * r + k]);
            if (k == 1 || k == 3 || k == 4) t[r] ^= d7[r];
        }
        s[k] = t[0] ^ kw[0];
        s[8 + k] = shift_xor<1>(t[1], kw[1]);
        s[16 + k] = shift_xor<2>(t[2], kw[2]);
        s[24 + k] = shift_xor<3>(t[3], kw[3]);
#pragma unroll
        for (int r = 0; r < 4; r++) prev[r] = cur[r];
    }
}

#ifndef WPE
#define WPE 2
#endif
template <bool STORE>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_aes_col4(const uint4* __restrict__ rk_col4 /*[n_qg][11][8][64] uint4*/, uint32_t NQ, uint64_t first_block,
                                                             uint64_t n_blocks, uint32_t blocks_per_wg, uint32_t* __restrict__ masks) {
    extern __shared__ uint4 lds4[];  // 11 * 8 * 64 (dynamic: the compiler must not cap the register budget by LDS occupancy)
    const uint32_t n_qg = NQ / 16;
    const uint32_t qg = blockIdx.x % n_qg;
    const uint64_t chunk = blockIdx.x / n_qg;
    {
        const uint4* src = rk_col4 + (size_t)qg * 11 * 8 * 64;
        for (uint32_t i = threadIdx.x; i < 11 * 8 * 64; i += blockDim.x) lds4[i] = src[i];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ql = lane >> 2, c = lane & 3;
    const uint32_t q = qg * 16 + ql;
    const uint4* rkl = lds4 + lane;
    const uint64_t j_lo = chunk * blocks_per_wg;
    const uint64_t j_hi = (j_lo + blocks_per_wg < n_blocks) ? j_lo + blocks_per_wg : n_blocks;
    // the counter meets state bytes 13..15 = (row 1..3, column 3); in the shifted state they sit in column 3 - row: lane c holds
    // byte 15 - c in row 3 - c (c < 3)
    const uint32_t m1 = c == 2 ? ~0u : 0u, m2 = c == 1 ? ~0u : 0u, m3 = c == 0 ? ~0u : 0u;
    for (uint64_t jl = j_lo + wave; jl < j_hi; jl += WAVES) {
        const uint32_t j = (uint32_t)(first_block + jl);
        uint32_t s[32];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 v = rkl[k * 64];
            s[k] = v.x; s[8 + k] = v.y; s[16 + k] = v.z; s[24 + k] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t cb = (uint32_t)__builtin_amdgcn_sbfe((int)j, 8 * c + k, 1);
            s[8 + k] = __builtin_amdgcn_bitop3_b32(s[8 + k], cb, m1, 0x78);   // a ^ (b & c)
            s[16 + k] = __builtin_amdgcn_bitop3_b32(s[16 + k], cb, m2, 0x78);
            s[24 + k] = __builtin_amdgcn_bitop3_b32(s[24 + k], cb, m3, 0x78);
        }
#pragma unroll 1
        for (int r = 1; r < 10; r++) round_col4(s, rkl + r * 8 * 64);
        // final round: SubBytes, (the state is already shifted), AddRoundKey unshifted
#pragma unroll
        for (int r = 0; r < 4; r++) sbox8(s[8 * r + 7], s[8 * r + 6], s[8 * r + 5], s[8 * r + 4], s[8 * r + 3], s[8 * r + 2], s[8 * r + 1], s[8 * r + 0]);
        uint32_t* out = masks + ((size_t)jl * 128 + 32 * c) * NQ + q;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 v = rkl[(10 * 8 + k) * 64];
            const uint32_t kw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t o = s[8 * r + k] ^ kw[r];
                if (STORE) out[(size_t)(8 * r + (7 - k)) * NQ] = o;
                else acc |= o;
            }
        }
        if (!STORE && acc == 0x12345678u) out[0] = acc;
    }
}

int main(int argc, char** argv) {
    init_sbox();
    const uint32_t NQ = 64;
    const uint64_t n_blocks = argc > 1 ? strtoull(argv[1], 0, 0) : 78381;
    const int wgs_target = argc > 2 ? atoi(argv[2]) : 256;
    const uint32_t n_qg = NQ / 16;
    // keys: slot = q*32 + s (s = 8*i4 + p at bit 31 - s)
    std::vector<uint8_t> keys((size_t)NQ * 32 * 16), rks((size_t)NQ * 32 * 176);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& b : keys) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; b = (uint8_t)(x >> 32); }
    for (size_t s = 0; s < (size_t)NQ * 32; s++) key_expand(&keys[16 * s], &rks[176 * s]);
    // col4 LDS image: [qg][area][g = bit plane][lane][e = row] of lane (ql, c); areas 0..9 shifted: key byte
    // (row, (c + row) & 3); area 10 unshifted
    std::vector<uint32_t> img((size_t)n_qg * 11 * 8 * 64 * 4);
    for (uint32_t qg = 0; qg < n_qg; qg++)
        for (int area = 0; area < 11; area++)
            for (int g = 0; g < 8; g++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 4; e++) {
                        const int bit = g, row = e, ql = lane >> 2, c = lane & 3;
                        const int col = area < 10 ? ((c + row) & 3) : c;
                        const int byte = 4 * col + row;
                        const uint32_t q = qg * 16 + ql;
                        uint32_t word = 0;
                        for (int s = 0; s < 32; s++) word |= (uint32_t)((rks[176 * ((size_t)q * 32 + s) + 16 * area + byte] >> bit) & 1) << (31 - s);
                        img[((((size_t)qg * 11 + area) * 8 + g) * 64 + lane) * 4 + e] = word;
                    }
    uint4* d_rk;
    CK(hipMalloc(&d_rk, img.size() * 4));
    CK(hipMemcpy(d_rk, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    uint32_t* d_masks;
    const size_t mask_words = (size_t)n_blocks * 128 * NQ;
    CK(hipMalloc(&d_masks, mask_words * 4));
    CK(hipMemset(d_masks, 0, mask_words * 4));
    uint64_t per = (n_blocks * n_qg + wgs_target - 1) / wgs_target;
    per = (per + WAVES - 1) / WAVES * WAVES;
    const uint64_t chunks = (n_blocks + per - 1) / per;
    const unsigned grid = (unsigned)(chunks * n_qg);
    const size_t LDS_BYTES = 11 * 8 * 64 * 16;
    CK(hipFuncSetAttribute((const void*)k_aes_col4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)k_aes_col4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)k_aes_col4<true>));
    printf("k_aes_col4: %d VGPRs, %zu B LDS, %zu B scratch; grid %u x %d threads, %llu blocks per workgroup\n", fa.numRegs, (size_t)fa.sharedSizeBytes,
           (size_t)fa.localSizeBytes, grid, WAVES * 64, (unsigned long long)per);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int store = 1; store >= 0; store--) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CK(hipEventRecord(e0));
            if (store) hipLaunchKernelGGL(k_aes_col4<true>, dim3(grid), dim3(WAVES * 64), LDS_BYTES, 0, d_rk, NQ, (uint64_t)0, n_blocks, (uint32_t)per, d_masks);
            else hipLaunchKernelGGL(k_aes_col4<false>, dim3(grid), dim3(WAVES * 64), LDS_BYTES, 0, d_rk, NQ, (uint64_t)0, n_blocks, (uint32_t)per, d_masks);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s: %.3f ms for %llu blocks x %u quads  (%.3g AES blocks/s; k_aes_gf2_masks: 1.63 ms at 78381)\n", store ? "with stores" : "no stores  ", best,
               (unsigned long long)n_blocks, NQ, (double)n_blocks * NQ * 32 / (best * 1e-3));
    }
    // check sampled words against the byte-wise cipher
    hipLaunchKernelGGL(k_aes_col4<true>, dim3(grid), dim3(WAVES * 64), LDS_BYTES, 0, d_rk, NQ, (uint64_t)0, n_blocks, (uint32_t)per, d_masks);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(mask_words);
    CK(hipMemcpy(h.data(), d_masks, mask_words * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    for (int t = 0; t < 4000; t++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t j = t < 8 ? (uint64_t)t : (t < 16 ? n_blocks - 1 - (t - 8) : x % n_blocks);
        const uint32_t q = (uint32_t)((x >> 40) % NQ), s = (uint32_t)((x >> 50) % 32);
        uint8_t in[16] = {0}, out[16];
        for (int i = 0; i < 8; i++) in[8 + i] = (uint8_t)(j >> (56 - 8 * i));
        encrypt(&rks[176 * ((size_t)q * 32 + s)], in, out);
        for (int b = 0; b < 128; b++) {
            const uint32_t want = (out[b >> 3] >> (7 - (b & 7))) & 1;
            const uint32_t got = (h[((size_t)j * 128 + b) * NQ + q] >> (31 - s)) & 1;
            bad += want != got;
            checked++;
        }
    }
    printf("check: %zu of %zu sampled keystream bits differ -> %s\n", bad, checked, bad ? "FAIL" : "OK");
    return bad ? 1 : 0;
}
