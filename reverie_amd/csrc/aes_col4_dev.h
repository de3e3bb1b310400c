// Device side of the lane-distributed bitsliced AES-128 (a quad of lanes per state, lane = column): shared by the GF(2) mask
// generator (aes_col4.hip) and the Z64 prover's fused level kernel (z64c4.hip).  See aes_col4.hip for the layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rv {

#define XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#ifndef C4_SBOX_INC
#define C4_SBOX_INC "aes_sbox.inc"
#endif

__device__ __forceinline__ void c4_sbox8(uint32_t& b7, uint32_t& b6, uint32_t& b5, uint32_t& b4, uint32_t& b3, uint32_t& b2, uint32_t& b1,
                                         uint32_t& b0) {
    const uint32_t U0 = b7, U1 = b6, U2 = b5, U3 = b4, U4 = b3, U5 = b2, U6 = b1, U7 = b0;
#include C4_SBOX_INC
    b7 = S0;
    b6 = S1;
    b5 = S2;
    b4 = S3;
    b3 = S4;
    b2 = S5;
    b1 = S6;
    b0 = S7;
}

// quad_perm_R(t) ^ k: lane c of every quad reads lane (c + R) & 3.  The builtin folds into one v_xor_b32_dpp and the compiler
// keeps the VALU-write -> DPP-read distance (inline assembly would hide the hazard from it).
template <int R>
__device__ __forceinline__ uint32_t c4_shift_xor(uint32_t t, uint32_t k) {
    if (R == 0) return t ^ k;
    constexpr int ctrl = R == 1 ? 0x39 : (R == 2 ? 0x4E : 0x93);  // quad_perm:[R, R+1, R+2, R+3] mod 4
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, ctrl, 0xf, 0xf, true) ^ k;
}

constexpr uint32_t C4_AREAS = 11;                      // srk0 .. srk9 (pre-shifted), rk10
constexpr uint32_t C4_IMG_U4 = C4_AREAS * 8 * 64;      // uint4 per group of 16 quad words: [area][plane][lane] -> rows 0..3
constexpr uint32_t C4_LDS_BYTES = C4_IMG_U4 * 16;      // 88 KiB

// one middle round on the shifted state: s = ShiftRows(MixColumns(SubBytes(s))) ^ srk.  MixColumns plane by plane:
// out_r[k] = d_r[k-1] ^ all[k] ^ a_r[k] (^ d_r[7] for k = 1, 3, 4; d_r[-1] = d_r[7]) with d_r = a_r ^ a_(r+1), all = a_0^a_1^a_2^a_3:
// beside the state only d[7], d[k-1], d[k] and the plane's four key words are live.  The d_r[7] of planes 1, 3, 4 rides in the
// PREVIOUS plane's d (planes 0, 2, 3 make theirs with a 3-input XOR: d' = d ^ d_r[7]; their `all` takes d_0[7] ^ d_2[7] back out
// in its own 3-input XOR) -- twelve XORs per round and lane less than adding it to the four outputs of three planes
__device__ __forceinline__ void c4_round(uint32_t* s, const uint4* rk4 /* lds + area*8*64 + lane */) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        c4_sbox8(s[8 * r + 7], s[8 * r + 6], s[8 * r + 5], s[8 * r + 4], s[8 * r + 3], s[8 * r + 2], s[8 * r + 1], s[8 * r + 0]);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t d7[4], prev[4];
#pragma unroll
    for (int r = 0; r < 4; r++) prev[r] = d7[r] = s[8 * r + 7] ^ s[8 * ((r + 1) & 3) + 7];
    const uint32_t dd = d7[0] ^ d7[2];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint4 kv = rk4[k * 64];
        const bool fold = k == 0 || k == 2 || k == 3;  // the next plane (1, 3, 4) wants d_r[7] in its d_r[k - 1]
        uint32_t cur[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            cur[r] = k == 7 ? d7[r] : fold ? XOR3(s[8 * r + k], s[8 * ((r + 1) & 3) + k], d7[r]) : (s[8 * r + k] ^ s[8 * ((r + 1) & 3) + k]);
        const uint32_t all = fold ? XOR3(cur[0], cur[2], dd) : cur[0] ^ cur[2];
        uint32_t t[4];
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = XOR3(prev[r], all, s[8 * r + k]);
        s[k] = t[0] ^ kv.x;
        s[8 + k] = c4_shift_xor<1>(t[1], kv.y);
        s[16 + k] = c4_shift_xor<2>(t[2], kv.z);
        s[24 + k] = c4_shift_xor<3>(t[3], kv.w);
#pragma unroll
        for (int r = 0; r < 4; r++) prev[r] = cur[r];
    }
}

// Rounds 0 and 1 of CTR block j (< 2^24) into the shifted state s.  Only state bytes 13..15 meet the counter; in the shifted state
// lane c (< 3) holds exactly one of them, byte 15 - c, in row r0 = 3 - c.  Everything else of round 1 is a constant of the key
// (K1, image area 1), and MixColumns is linear: the lane runs ONE S-box, v = S(rk0[15 - c] ^ counter byte c), and adds its
// column's share 2v / 3v / v / v (rows r0, r0 - 1, the other two) -- 197 instructions instead of a full round's 417.
__device__ __forceinline__ void c4_rounds_0_1(uint32_t j, uint32_t c, const uint4* rkl, uint32_t* s) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = rkl[k * 64].x ^ (uint32_t)__builtin_amdgcn_sbfe((int)j, 8 * c + k, 1);
    c4_sbox8(v[7], v[6], v[5], v[4], v[3], v[2], v[1], v[0]);
    // e[r] = all ones in the lanes whose counter byte sits in row r (r0 = 3 - c); made here, per block, from an opaque copy of c:
    // hoisted out of the block loop the four masks would sit in registers the rounds need (selects on c itself compile to branches)
    uint32_t co = c;
    asm volatile("" : "+v"(co));
    uint32_t e[4];
#pragma unroll
    for (int r = 0; r < 4; r++) e[r] = co == (uint32_t)(3 - r) ? ~0u : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] &= ~e[0];  // column 3 holds no counter byte: v = 0 there (e[0] marks c == 3)
    const uint32_t x[8] = {v[7], v[0] ^ v[7], v[1], v[2] ^ v[7], v[3] ^ v[7], v[4], v[5], v[6]};  // xtime(v)
    const uint4* k1 = rkl + 8 * 64;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint4 kv = k1[k * 64];
        // row r of the lane's column gets 2v = x where e[r], 3v = x ^ v where e[r + 1], v elsewhere: two 3-input LUTs per word
        uint32_t t[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t sel = __builtin_amdgcn_bitop3_b32(e[r], x[k], v[k], 0xca);        // e ? x : v
            t[r] = __builtin_amdgcn_bitop3_b32(sel, x[k], e[(r + 1) & 3], 0x78);             // sel ^ (x & e')
        }
        s[k] = t[0] ^ kv.x;
        s[8 + k] = c4_shift_xor<1>(t[1], kv.y);
        s[16 + k] = c4_shift_xor<2>(t[2], kv.z);
        s[24 + k] = c4_shift_xor<3>(t[3], kv.w);
    }
}

}  // namespace rv
