// AES-128-CTR share expansion for gfx950 (integer VALU only; no MFMA — this is XOR/AND work).
//
// Replaces (all under /root/reference/src/):
//   crypto/prg.rs:16-37            PRG::new / gen            (AES-128-CTR, BE counter, IV 0)
//   transcript/mod.rs:99-122       expand_seed               (rep seed -> 8 player keys)
//   generator/batch.rs:13-40       BatchGen::gen             (one 16-byte batch per player)
//   generator/share.rs:54-65       ShareGen::next refill
//   algebra/gf2/domain.rs:66-378   batches_to_shares + the AVX2 movemask transpose
//
// Design: the mask generator is BITSLICED with one lane = 32 AES blocks = the 4
// repetitions x 8 players of one quad word, all at the same CTR block index j.  Register
// b of the bitsliced state then holds keystream bit b for those 32 (rep, player) slots —
// which IS the packed share word for mask index 128*j + b (bit 31-(8*i4+p)).  The
// reference's 64x128 bit transpose (its top CPU cost) disappears: the
// bitsliced cipher emits the transposed layout natively, and a wavefront stores one
// 256-byte row per mask index, fully coalesced.
#include <stdlib.h>

#include <algorithm>

#include "internal.h"
#include "z64_dev.h"
#include "aes_col4_dev.h"
#include "launch.h"

namespace rv {

__device__ __constant__ uint8_t SBOX_TAB[256] = {
#include "aes_sbox_table.inc"
};

// ------------------------------------------------------------------------------------
// byte-wise AES for the tiny per-shard setup work (256 seeds, 2048 key schedules)
// ------------------------------------------------------------------------------------
__device__ inline uint8_t xtime8(uint8_t x) { return (uint8_t)((x << 1) ^ ((x & 0x80) ? 0x1b : 0)); }

__device__ void key_expand(const uint8_t key[16], uint8_t rk[176]) {
    for (int i = 0; i < 16; i++) rk[i] = key[i];
    uint8_t rcon = 1;
    for (int r = 1; r <= 10; r++) {
        const uint8_t* p = rk + 16 * (r - 1);
        uint8_t* q = rk + 16 * r;
        q[0] = p[0] ^ SBOX_TAB[p[13]] ^ rcon;
        q[1] = p[1] ^ SBOX_TAB[p[14]];
        q[2] = p[2] ^ SBOX_TAB[p[15]];
        q[3] = p[3] ^ SBOX_TAB[p[12]];
        rcon = xtime8(rcon);
        for (int i = 4; i < 16; i++) q[i] = p[i] ^ q[i - 4];
    }
}

__device__ void encrypt_bytes(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int r = 1; r <= 10; r++) {
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < 4; row++) t[4 * c + row] = SBOX_TAB[s[4 * ((c + row) & 3) + row]];
        if (r < 10) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                uint8_t all = a0 ^ a1 ^ a2 ^ a3;
                s[4 * c + 0] = a0 ^ all ^ xtime8(a0 ^ a1);
                s[4 * c + 1] = a1 ^ all ^ xtime8(a1 ^ a2);
                s[4 * c + 2] = a2 ^ all ^ xtime8(a2 ^ a3);
                s[4 * c + 3] = a3 ^ all ^ xtime8(a3 ^ a0);
            }
        } else {
            for (int i = 0; i < 16; i++) s[i] = t[i];
        }
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
    }
    for (int i = 0; i < 16; i++) out[i] = s[i];
}

__device__ inline void ctr_block(uint64_t j, uint8_t b[16]) {
    for (int i = 0; i < 8; i++) {
        b[i] = 0;
        b[8 + i] = (uint8_t)(j >> (56 - 8 * i));
    }
}

// expand_seed: keys[r][p] = AES_{seed[r]}(BE128(p))
struct B_k_expand_seeds {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ seeds, uint32_t n_reps, uint8_t* __restrict__ keys) const {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reps * 8) return;
    slot_key(seeds, t, keys);
    }
    // slot t = 8 * repetition + player
    static __device__ __forceinline__ void slot_key(const uint8_t* __restrict__ seeds, uint32_t t, uint8_t* __restrict__ keys) {
    uint32_t r = t >> 3, p = t & 7;
    uint8_t key[16], rk[176], in[16], out[16];
    for (int i = 0; i < 16; i++) key[i] = seeds[16 * r + i];
    key_expand(key, rk);
    ctr_block(p, in);
    encrypt_bytes(rk, in, out);
    for (int i = 0; i < 16; i++) keys[16 * t + i] = out[i];
}
};
__global__ void k_expand_seeds(const uint8_t* __restrict__ seeds, uint32_t n_reps, uint8_t* __restrict__ keys) {
    B_k_expand_seeds{}(seeds, n_reps, keys);
}

struct B_k_key_schedule {
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ keys, uint32_t n_slots, uint8_t* __restrict__ rkbytes) const {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slots) return;
    slot_schedule(keys, t, rkbytes + RK_BYTES * (size_t)t);
    }
    // slot t's RK_BYTES at dst (global or LDS)
    static __device__ __forceinline__ void slot_schedule(const uint8_t* __restrict__ keys, uint32_t t, uint8_t* dst) {
    uint8_t key[16], rk[176];
    for (int i = 0; i < 16; i++) key[i] = keys[16 * t + i];
    key_expand(key, rk);
    for (int i = 0; i < 176; i++) dst[i] = rk[i];
    // first-round constants (layout: internal.h).  Round 0 of a CTR block with j < 2^24: x = rk0 ^ (0,..,0,j2,j1,j0).
    uint8_t sb[16], k1[16];
    for (int i = 0; i < 13; i++) sb[i] = SBOX_TAB[rk[i]];
    sb[13] = sb[14] = sb[15] = 0;  // the varying bytes' S-box outputs are added by the mask kernel (MixColumns is linear)
    for (int c = 0; c < 4; c++) {
        const uint8_t a0 = sb[4 * ((c + 0) & 3) + 0], a1 = sb[4 * ((c + 1) & 3) + 1], a2 = sb[4 * ((c + 2) & 3) + 2],
                      a3 = sb[4 * ((c + 3) & 3) + 3];
        const uint8_t all = a0 ^ a1 ^ a2 ^ a3;
        k1[4 * c + 0] = a0 ^ all ^ xtime8(a0 ^ a1) ^ rk[16 + 4 * c + 0];
        k1[4 * c + 1] = a1 ^ all ^ xtime8(a1 ^ a2) ^ rk[16 + 4 * c + 1];
        k1[4 * c + 2] = a2 ^ all ^ xtime8(a2 ^ a3) ^ rk[16 + 4 * c + 2];
        k1[4 * c + 3] = a3 ^ all ^ xtime8(a3 ^ a0) ^ rk[16 + 4 * c + 3];
    }
    for (int i = 0; i < 16; i++) dst[176 + i] = 0;
    for (int r = 0; r < 4; r++) dst[176 + r] = SBOX_TAB[k1[12 + r]];
    for (int i = 13; i < 16; i++) dst[176 + i] = rk[i];
    for (int i = 0; i < 16; i++) dst[192 + i] = k1[i];
}
};
__global__ void k_key_schedule(const uint8_t* __restrict__ keys, uint32_t n_slots, uint8_t* __restrict__ rkbytes) {
    B_k_key_schedule{}(keys, n_slots, rkbytes);
}

// 32x32 bit-matrix transpose (Hacker's Delight 7-3), fully unrolled: registers only
__device__ __forceinline__ void transpose32(uint32_t* A) {
    uint32_t m = 0x0000FFFFu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = (A[k] ^ (A[k + j] >> j)) & m;
            A[k] ^= t;
            A[k + j] ^= (t << j);
        }
        m ^= (m << (j >> 1));
    }
}

// rk[(round*128 + 8*byte + bit)*NQ + q] = bit `bit` of round-key byte `byte` of the 32
// slots of quad q, slot (i4, p) at bit 31 - (8*i4 + p).  Slots are numbered rep*8 + p.
struct B_k_bitslice_rk {
    // thread = (quad q, group of 4 key bytes): 32 dword loads (one per slot), 32 words out.  (One thread per output
    // word re-read every byte eight times through 3.4 M byte loads per proof: 10 us per proof, 2.7 ms per batch of 256.)
    __device__ __forceinline__ void operator()(const uint8_t* __restrict__ rkbytes, uint32_t NQ, uint32_t* __restrict__ rk) const {
    static_assert(RK_BYTES % 4 == 0, "dword loads");
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (RK_BYTES / 4) * NQ) return;
    const uint32_t q = t % NQ, bg = t / NQ;
    uint32_t v[32];
#pragma unroll
    for (uint32_t s = 0; s < 32; s++) v[s] = *(const uint32_t*)(rkbytes + RK_BYTES * (size_t)(q * 32 + s) + 4 * bg);
    // w_k = bit k of the 32 slots' words, slot s at bit 31 - s: the 32x32 bit transpose (row i of the result holds bit 31 - i
    // of every input row, input row j at bit 31 - j) read backwards
    transpose32(v);
#pragma unroll
    for (uint32_t k = 0; k < 32; k++) rk[(size_t)(32 * bg + k) * NQ + q] = v[31 - k];  // k = 8 * (byte in the group) + bit
}
};
__global__ void k_bitslice_rk(const uint8_t* __restrict__ rkbytes, uint32_t NQ, uint32_t* __restrict__ rk) {
    B_k_bitslice_rk{}(rkbytes, NQ, rk);
}

// The whole key setup of a prover's shard in ONE launch, a workgroup per quad word (its 32 slots = 4 repetitions x 8 players):
// seed -> player key (B_k_expand_seeds) -> round keys and first-round constants (B_k_key_schedule) by 32 threads into LDS, the quad
// word's bit planes (B_k_bitslice_rk) out of LDS, and -- img != null -- its lanes of the lane-distributed generator's key image
// (aes_col4.hip: k_rk_col4) out of the planes.  Four dependent launches of 4 - 8 us each otherwise, at the head of every proof.
__global__ __launch_bounds__(256) void k_setup_keys(const uint8_t* __restrict__ seeds, uint32_t NQ, uint8_t* __restrict__ keys, uint8_t* __restrict__ rkbytes,
                                                    uint32_t* __restrict__ rk, uint32_t* __restrict__ img) {
    __shared__ __attribute__((aligned(16))) uint8_t s_rkb[32][RK_BYTES];
    __shared__ uint32_t s_pl[RK_BYTES * 8];  // plane 8 * byte + bit of this quad word
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    if (t < 32) {
        const uint32_t slot = q * 32 + t;
        B_k_expand_seeds::slot_key(seeds, slot, keys);
        B_k_key_schedule::slot_schedule(keys, slot, s_rkb[t]);
        uint8_t* dst = rkbytes + RK_BYTES * (size_t)slot;
        for (uint32_t i = 0; i < RK_BYTES; i += 4) *(uint32_t*)(dst + i) = *(const uint32_t*)(&s_rkb[t][i]);
    }
    __syncthreads();
    if (t < RK_BYTES / 4) {
        const uint32_t bg = t;
        uint32_t v[32];
#pragma unroll
        for (uint32_t s = 0; s < 32; s++) v[s] = *(const uint32_t*)(&s_rkb[s][4 * bg]);
        transpose32(v);
#pragma unroll
        for (uint32_t k = 0; k < 32; k++) {
            rk[(size_t)(32 * bg + k) * NQ + q] = v[31 - k];
            s_pl[32 * bg + k] = v[31 - k];
        }
    }
    if (!img) return;
    __syncthreads();
    const uint32_t qg = q >> 4, ql = q & 15;
    for (uint32_t e = t; e < C4_AREAS * 8 * 16; e += 256) {
        const uint32_t row = e & 3, c = (e >> 2) & 3, k = (e >> 4) & 7, area = e >> 7;
        const uint32_t lane = 4 * ql + c;
        const size_t at = ((((size_t)qg * C4_AREAS + area) * 8 + k) * 64 + lane) * 4 + row;
        // (the mapping of k_rk_col4)
        uint32_t ga = area, byte = 4 * ((c + row) & 3) + row;
        if (area == 10) byte = 4 * c + row;
        if (area == 1) ga = 12;
        uint32_t val;
        if (area == 0) {
            ga = 11;
            byte = 15 - c;
            val = (row != 0 || c == 3) ? 0u : s_pl[ga * 128 + 8 * byte + k];
        } else {
            val = s_pl[ga * 128 + 8 * byte + k];
        }
        img[at] = val;
    }
}
void launch_setup_keys(hipStream_t st, const uint8_t* d_seeds, uint32_t NQ, uint8_t* d_keys, uint8_t* d_rkbytes, uint32_t* d_rk, uint32_t* d_img) {
    hipLaunchKernelGGL(k_setup_keys, dim3(NQ), dim3(256), 0, st, d_seeds, NQ, d_keys, d_rkbytes, d_rk, d_img);
}

// test hook / Z64 path helper: plain CTR blocks, one thread per (key, block)
__global__ void k_aes_blocks(const uint8_t* __restrict__ rkbytes, uint32_t n_keys, uint64_t first, uint64_t n_blocks,
                             uint8_t* __restrict__ out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_keys * n_blocks) return;
    uint64_t k = t / n_blocks, b = t % n_blocks;
    uint8_t rk[176], in[16], o[16];
    for (int i = 0; i < 176; i++) rk[i] = rkbytes[RK_BYTES * k + i];
    ctr_block(first + b, in);
    encrypt_bytes(rk, in, o);
    for (int i = 0; i < 16; i++) out[16 * t + i] = o[i];
}

// ------------------------------------------------------------------------------------
// bitsliced AES-128: state s[8*i + k] = bit k (0 = LSB) of state byte i, 32 blocks/lane
// ------------------------------------------------------------------------------------
#define XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)

__device__ __forceinline__ void sbox8(uint32_t& b7, uint32_t& b6, uint32_t& b5, uint32_t& b4, uint32_t& b3, uint32_t& b2,
                                      uint32_t& b1, uint32_t& b0) {
    const uint32_t U0 = b7, U1 = b6, U2 = b5, U3 = b4, U4 = b3, U5 = b2, U6 = b1, U7 = b0;
#include "aes_sbox.inc"
    b7 = S0;
    b6 = S1;
    b5 = S2;
    b4 = S3;
    b3 = S4;
    b2 = S5;
    b1 = S6;
    b0 = S7;
}

// SubBytes + ShiftRows: t[new position] = S(s[old position])
__device__ __forceinline__ void sub_shift(const uint32_t* s, uint32_t* t) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int row = 0; row < 4; row++) {
            const int src = 8 * (4 * ((c + row) & 3) + row), dst = 8 * (4 * c + row);
#pragma unroll
            for (int k = 0; k < 8; k++) t[dst + k] = s[src + k];
            sbox8(t[dst + 7], t[dst + 6], t[dst + 5], t[dst + 4], t[dst + 3], t[dst + 2], t[dst + 1], t[dst + 0]);
        }
    }
}

// MixColumns + AddRoundKey: s = MC(t) ^ rk, round keys read from LDS (stride QW words)
template <int QW>
__device__ __forceinline__ void mix_ark(const uint32_t* t, uint32_t* s, const uint32_t* rk) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t* a = t + 32 * c;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t* a0 = a + 8 * r;
            const uint32_t* a1 = a + 8 * ((r + 1) & 3);
            const uint32_t* a2 = a + 8 * ((r + 2) & 3);
            const uint32_t* a3 = a + 8 * ((r + 3) & 3);
            uint32_t d[8];
#pragma unroll
            for (int k = 0; k < 8; k++) d[k] = a0[k] ^ a1[k];
            // xtime(d): y0=d7 y1=d0^d7 y2=d1 y3=d2^d7 y4=d3^d7 y5=d4 y6=d5 y7=d6
            uint32_t x[8] = {d[7], d[0] ^ d[7], d[1], d[2] ^ d[7], d[3] ^ d[7], d[4], d[5], d[6]};
            uint32_t* o = s + 32 * c + 8 * r;
#pragma unroll
            for (int k = 0; k < 8; k++) o[k] = x[k] ^ a1[k] ^ a2[k] ^ a3[k] ^ rk[(32 * c + 8 * r + k) * QW];
        }
    }
}

// One full middle round, column by column: n = MixColumns(ShiftRows(SubBytes(s))) ^ rk.
// Output column c only needs the four S-box outputs it consumes, so at most one column of
// temporaries (32 registers) is live beside the old and the new state: this is what lets the
// kernel fit 2 wavefronts per SIMD (<= 256 registers) without scratch spills.
// SECOND = round 2 of a block with j < 2^24: state column 3 is a per-key constant, so the four S-boxes it
// feeds are read from LDS (sk = S(round-1 output bytes 12..15), bit k of byte r at sk[(8*r + k)*QW]).
template <int QW, bool SECOND = false>
__device__ __forceinline__ void round_cols(const uint32_t* s, uint32_t* n, const uint32_t* rk, const uint32_t* sk = nullptr) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t col[32];
        // this column's 32 round-key words: issued first, consumed after the four S-boxes (~320 VALU ops),
        // so the LDS latency is covered (SQ_WAIT_ANY was 20 % of the wave cycles with the reads next to their use)
        uint32_t rkv[32];
#pragma unroll
        for (int k = 0; k < 32; k++) rkv[k] = rk[(32 * c + k) * QW];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int row = 0; row < 4; row++) {
            const int src = 8 * (4 * ((c + row) & 3) + row);
            if (SECOND && ((c + row) & 3) == 3) {
#pragma unroll
                for (int k = 0; k < 8; k++) col[8 * row + k] = sk[(8 * row + k) * QW];
                continue;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) col[8 * row + k] = s[src + k];
            sbox8(col[8 * row + 7], col[8 * row + 6], col[8 * row + 5], col[8 * row + 4], col[8 * row + 3], col[8 * row + 2],
                  col[8 * row + 1], col[8 * row + 0]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t* a0 = col + 8 * r;
            const uint32_t* a1 = col + 8 * ((r + 1) & 3);
            const uint32_t* a2 = col + 8 * ((r + 2) & 3);
            const uint32_t* a3 = col + 8 * ((r + 3) & 3);
            uint32_t d[8];
#pragma unroll
            for (int k = 0; k < 8; k++) d[k] = a0[k] ^ a1[k];
            // out = xtime(d) ^ a1 ^ a2 ^ a3 ^ rk with xtime(d) = {d7, d0^d7, d1, d2^d7, d3^d7, d4, d5, d6},
            // written as 3-input XORs (v_bitop3_b32 0x96): 27 ops per byte instead of 43
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t rkw = rkv[8 * r + k];
                const uint32_t lo = d[(k + 7) & 7];  // d[k-1], d[7] for k = 0
                uint32_t v;
                if (k == 1 || k == 3 || k == 4)
                    v = XOR3(XOR3(lo, d[7], a1[k]), a2[k], a3[k]) ^ rkw;
                else
                    v = XOR3(XOR3(lo, a1[k], a2[k]), a3[k], rkw);
                n[32 * c + 8 * r + k] = v;
            }
        }
        // keep the scheduler from hoisting the next columns' LDS reads / S-boxes up here (that is
        // what blew the register budget: 128 round-key words in flight at once)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Rounds 0..9 of CTR block j (< 2^24) into s; t is the ping-pong buffer.  rkl: this lane's LDS round keys with the
// rk0 / rk1 areas replaced by the first-round constants (internal.h: RK_BYTES): only state bytes 13..15 meet the
// counter, so round 1 is 3 S-boxes plus the linear spread of those three bytes over K1 instead of 16 S-boxes and a
// full MixColumns, and round 2 reads 4 of its 16 S-box outputs from LDS.  Rounds 3..9 run two per loop trip so no
// register shuffling is needed at the back edge.
// REP (rep-major mask generator): the lane's 32 slots are 4 CONSECUTIVE counter blocks x 8 players of one repetition
// (counter 4j + c in byte c from the MSB), so the two lowest counter bits are constants of the slot position and the
// rest is j shifted up by two; otherwise all 32 slots share the counter j.
template <bool REP>
__device__ __forceinline__ uint32_t ctr_bit_word(uint64_t j, int bit) {
    if (!REP) return (uint32_t)0 - (uint32_t)((j >> bit) & 1);
    if (bit == 0) return 0x00FF00FFu;  // c & 1: bytes 1 and 3 (from the MSB)
    if (bit == 1) return 0x0000FFFFu;  // c & 2: bytes 2 and 3
    return (uint32_t)0 - (uint32_t)((j >> (bit - 2)) & 1);
}

template <int QW, bool REP = false>
__device__ __forceinline__ void rounds_0_to_9(uint64_t j, uint32_t* s, uint32_t* t, const uint32_t* rkl) {
    const uint32_t* a0 = rkl;             // area 0: SK (bytes 0..3), rk0[13..15] (bytes 13..15)
    const uint32_t* k1 = rkl + 128 * QW;  // area 1: K1
#pragma unroll
    for (int i = 0; i < 128; i++) s[i] = k1[i * QW];
#pragma unroll
    for (int b = 13; b < 16; b++) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t cb = ctr_bit_word<REP>(j, 8 * (15 - b) + k);
            v[k] = a0[(8 * b + k) * QW] ^ cb;
        }
        sbox8(v[7], v[6], v[5], v[4], v[3], v[2], v[1], v[0]);
        // ShiftRows sends byte 12 + r (row r of column 3) to column (3 - r) & 3; MixColumns then adds
        // 2v to row r, 3v to row r - 1, v to the other two rows of that column
        const int r = b - 12, c = (3 - r) & 3;
        const uint32_t x[8] = {v[7], v[0] ^ v[7], v[1], v[2] ^ v[7], v[3] ^ v[7], v[4], v[5], v[6]};  // xtime(v)
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t& o = s[32 * c + 8 * i + k];
                if (i == r)
                    o ^= x[k];
                else if (((i + 1) & 3) == r)
                    o = XOR3(o, x[k], v[k]);
                else
                    o ^= v[k];
            }
        }
    }
    round_cols<QW, true>(s, t, rkl + 2 * 128 * QW, a0);
#pragma unroll 1
    for (int r = 3; r < 9; r += 2) {
        round_cols<QW>(t, s, rkl + r * 128 * QW);
        round_cols<QW>(s, t, rkl + (r + 1) * 128 * QW);
    }
    round_cols<QW>(t, s, rkl + 9 * 128 * QW);
}

// this workgroup's round keys into LDS: lds[(area*128 + idx)*QW + ql] for quads qg*QW .. qg*QW+QW-1.  LDS areas
// 0 and 1 hold the first-round constants (global areas 11 and 12), not rk0 / rk1 (see rounds_0_to_9)
template <int QW>
__device__ __forceinline__ void stage_round_keys(const uint32_t* __restrict__ rk, uint32_t NQ, uint32_t qg, uint32_t* lds_rk) {
    // several loads in flight per thread before the first LDS store: one load, one store at a time is 44 dependent L2
    // round trips for a 512-thread workgroup at QW = 16
    constexpr uint32_t N = 11 * 128 * QW, B = 11;
    for (uint32_t i0 = threadIdx.x; i0 < N; i0 += B * blockDim.x) {
        uint32_t v[B];
#pragma unroll
        for (uint32_t k = 0; k < B; k++) {
            const uint32_t i = i0 + k * blockDim.x;
            const uint32_t area = i / (128 * QW), w = i % (128 * QW);
            const uint32_t ga = area == 0 ? 11u : (area == 1 ? 12u : area);
            v[k] = i < N ? rk[(size_t)(ga * 128 + w / QW) * NQ + qg * QW + (w % QW)] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < B; k++) {
            const uint32_t i = i0 + k * blockDim.x;
            if (i < N) lds_rk[i] = v[k];
        }
    }
    __syncthreads();
}

// Mask generator.  A workgroup owns QW consecutive quads (QW*32 AES keys) and keeps their
// 11 bitsliced round keys in LDS (QW*5.5 KiB; 88 KiB at QW=16) for its whole lifetime; its
// 4 wavefronts then stream CTR blocks: lane = (block sub-index, quad), 64/QW blocks per
// wavefront per iteration.  Round keys never come from L2 inside the round loop (a first
// version that read them from global memory was 15x slower: every wavefront of the chip
// requested the same 256-byte row at the same time and serialised on one L2 channel).
template <int QW>
struct B_k_aes_gf2_masks {
    __device__ __forceinline__ void operator()(const uint32_t* __restrict__ rk, const uint32_t* __restrict__ keep, uint32_t NQ, uint64_t first_block, uint64_t n_blocks, uint32_t blocks_per_wg, uint32_t* __restrict__ masks) const {
    __shared__ uint32_t lds_rk[11 * 128 * QW];
    constexpr uint32_t JW = 64 / QW;  // CTR blocks per wavefront per iteration
    const uint32_t n_qg = NQ / QW;
    const uint32_t qg = blockIdx.x % n_qg;
    const uint64_t chunk = blockIdx.x / n_qg;
    // stage this workgroup's round keys: rk[(round*128+idx)*NQ + qg*QW + ql]
    stage_round_keys<QW>(rk, NQ, qg, lds_rk);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ql = lane % QW, jsub = lane / QW;
    const uint32_t q = qg * QW + ql;
    const uint32_t kp = keep ? keep[q] : 0xFFFFFFFFu;
    const uint32_t* rkl = lds_rk + ql;
    const uint64_t j_lo = chunk * blocks_per_wg;
    const uint64_t j_hi = (j_lo + blocks_per_wg < n_blocks) ? j_lo + blocks_per_wg : n_blocks;
    for (uint64_t jb = j_lo + (uint64_t)wave * JW; jb < j_hi; jb += 8 * JW) {
        const uint64_t jl = jb + jsub;
        if (jl >= j_hi) continue;
        const uint64_t j = first_block + jl;
        uint32_t s[128], t[128];
        rounds_0_to_9<QW>(j, s, t, rkl);
        sub_shift(s, t);
        const uint32_t* rk10 = rkl + 10 * 128 * QW;
        uint32_t* out = masks + (size_t)jl * 128 * NQ + q;
        // keystream bit order is MSB-first inside each byte (gf2/domain.rs: share 8i+j <- bit 7-j of byte i)
#pragma unroll
        for (int i = 0; i < 16; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t v = __builtin_amdgcn_bitop3_b32(t[8 * i + k], rk10[(8 * i + k) * QW], kp, 0x28);  // (t ^ rk) & kp
                out[(size_t)(8 * i + (7 - k)) * NQ] = v;
            }
        }
    }
}
};
template <int QW>
__global__ __launch_bounds__(512, 2) void k_aes_gf2_masks(const uint32_t* __restrict__ rk, const uint32_t* __restrict__ keep, uint32_t NQ, uint64_t first_block, uint64_t n_blocks, uint32_t blocks_per_wg, uint32_t* __restrict__ masks) {
    B_k_aes_gf2_masks<QW>{}(rk, keep, NQ, first_block, n_blocks, blocks_per_wg, masks);
}



// ---- launchers ----
void launch_expand_seeds(hipStream_t st, const uint8_t* d_seeds, uint32_t n_reps, uint8_t* d_keys) {
    uint32_t n = n_reps * 8;
    launch<B_k_expand_seeds, 64>(k_expand_seeds, st, dim3((n + 63) / 64), dim3(64), d_seeds, n_reps, d_keys);
}
void launch_key_schedule(hipStream_t st, const uint8_t* d_keys, uint32_t n_slots, uint8_t* d_rkbytes) {
    launch<B_k_key_schedule, 64>(k_key_schedule, st, dim3((n_slots + 63) / 64), dim3(64), d_keys, n_slots, d_rkbytes);
}
void launch_bitslice_rk(hipStream_t st, const uint8_t* d_rkbytes, uint32_t NQ, uint32_t* d_rk) {
    uint32_t n = (RK_BYTES / 4) * NQ;
    launch<B_k_bitslice_rk, 256>(k_bitslice_rk, st, dim3((n + 255) / 256), dim3(256), d_rkbytes, NQ, d_rk);
}

// Measured (round 2, config 5: 2.05e9 cipher blocks per proof, 26.8 ms): with the stores left out the kernel takes 24.5 ms,
// i.e. it is VALU-bound on the cipher (20.5 ms at the GF(2) generator's rate) plus the four 32x32 bit transposes of a
// lane (~2 000 of ~14 700 instructions); the 8-byte-per-lane stores cost 2.3 ms.  Passing the runs through an LDS tile
// so that a half-wavefront stores 512 contiguous bytes was built and measured: 38 ms -- the cipher leaves no register
// for the tile bookkeeping and the spills sit in its last rounds -- and dropped.
// Z64 mask generator (replaces BatchZ64::random + DomainZ64::batches_to_shares,
// src/algebra/z64/batch.rs:26-29, z64/domain.rs:64-83): the same bitsliced cipher, then the
// 128 bit-planes of a lane are transposed back to two little-endian u64 per (rep, player)
// slot.  masks64[(2j+h)*S + slot] with S = NQ*32 slots, slot = rep*8 + player.
template <int QW>
__global__ __launch_bounds__(512, 2) void k_aes_z64_masks(const uint32_t* __restrict__ rk, const uint32_t* __restrict__ keep,
                                                       uint32_t NQ, uint64_t first_block, uint64_t n_blocks, uint32_t blocks_per_wg,
                                                       uint64_t* __restrict__ masks64) {
    __shared__ uint32_t lds_rk[11 * 128 * QW];
    constexpr uint32_t JW = 64 / QW;
    const uint32_t n_qg = NQ / QW;
    const uint32_t qg = blockIdx.x % n_qg;
    const uint64_t chunk = blockIdx.x / n_qg;
    stage_round_keys<QW>(rk, NQ, qg, lds_rk);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ql = lane % QW, jsub = lane / QW;
    const uint32_t q = qg * QW + ql;
    const uint32_t kp = keep ? keep[q] : 0xFFFFFFFFu;
    const uint32_t* rkl = lds_rk + ql;
    const uint64_t S = (uint64_t)NQ * 32;
    const uint64_t j_lo = chunk * blocks_per_wg;
    const uint64_t j_hi = (j_lo + blocks_per_wg < n_blocks) ? j_lo + blocks_per_wg : n_blocks;
    for (uint64_t jb = j_lo + (uint64_t)wave * JW; jb < j_hi; jb += 8 * JW) {
        const uint64_t j = jb + jsub;  // block inside this launch: its output slot; CTR index first_block + j
        if (j >= j_hi) continue;
        uint32_t s[128], t[128];
        rounds_0_to_9<QW>(first_block + j, s, t, rkl);
        sub_shift(s, t);
        const uint32_t* rk10 = rkl + 10 * 128 * QW;
#pragma unroll
        for (int i = 0; i < 128; i++) t[i] ^= rk10[i * QW];
        // plane 8*i + k = bit k of keystream byte i; u64 h, bit b  <->  plane 64*h + b
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t lo[32], hi[32];
#pragma unroll
            for (int k = 0; k < 32; k++) {
                lo[k] = t[64 * h + 31 - k];
                hi[k] = t[64 * h + 32 + 31 - k];
            }
            transpose32(lo);
            transpose32(hi);
            uint64_t* out = masks64 + (2 * j + h) * S + (uint64_t)q * 32;
#pragma unroll
            for (int sl = 0; sl < 32; sl++) {
                const uint32_t on = (uint32_t)0 - ((kp >> (31 - sl)) & 1u);  // omitted player's stream stays zero
                out[sl] = ((uint64_t)(hi[sl] & on) << 32) | (lo[sl] & on);
            }
        }
    }
}

template <int QW>
static void launch_masks_qw(hipStream_t st, const uint32_t* d_rk, const uint32_t* d_keep, uint32_t NQ, uint64_t first_block,
                            uint64_t n_blocks, uint32_t* d_masks) {
    const uint32_t n_qg = NQ / QW;
    constexpr uint32_t JW = 64 / QW;
    // Target workgroup count: a workgroup takes a whole CU (88 KiB LDS, 512 x 256 registers), so ONE workgroup per CU
    // = one perfectly balanced generation (measured 1.61 ms against 1.64 with two generations of half the size and
    // 1.73 with eight); each workgroup's share is a multiple of one full iteration (8 waves x JW blocks).
    static const uint64_t target_wgs = [] {
        if (const char* e = getenv("RV_AES_WGS")) return (uint64_t)std::max(atoi(e), 1);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return (uint64_t)cus;
    }();
    // recorded for a batch of proofs (rv_prove_batch): the batch supplies the parallelism, so a proof's share of the
    // chip is its 1/batch of the workgroups, at least one per quad group -- each workgroup fills 88 KiB of LDS with
    // round keys before its first block
    const uint64_t wgs = g_recorder ? std::max<uint64_t>(n_qg, target_wgs / std::max(g_recorder->batch, 1u))
                                    : target_wgs;
    uint64_t per = (n_blocks * n_qg + wgs - 1) / wgs;
    per = ((per + 8 * JW - 1) / (8 * JW)) * (8 * JW);
    const uint64_t chunks = (n_blocks + per - 1) / per;
    launch<B_k_aes_gf2_masks<QW>, 512>(k_aes_gf2_masks<QW>, st, dim3((unsigned)(chunks * n_qg)), dim3(512), d_rk, d_keep, NQ, first_block,
                       n_blocks, (uint32_t)per, d_masks);
}

void launch_aes_gf2_masks(hipStream_t st, const uint32_t* d_rk, const uint32_t* d_keep, uint32_t NQ, uint64_t first_block,
                          uint64_t n_blocks, uint32_t* d_masks) {
    if (!n_blocks) return;
    if (NQ % 16 == 0)
        launch_masks_qw<16>(st, d_rk, d_keep, NQ, first_block, n_blocks, d_masks);
    else if (NQ % 8 == 0)
        launch_masks_qw<8>(st, d_rk, d_keep, NQ, first_block, n_blocks, d_masks);
    else
        launch_masks_qw<2>(st, d_rk, d_keep, NQ, first_block, n_blocks, d_masks);
}
template <int QW>
static void launch_z64_qw(hipStream_t st, const uint32_t* d_rk, const uint32_t* d_keep, uint32_t NQ, uint64_t first_block, uint64_t n_blocks,
                          uint64_t* d_masks64) {
    const uint32_t n_qg = NQ / QW;
    constexpr uint32_t JW = 64 / QW;
    uint64_t per = (n_blocks * n_qg + 511) / 512;
    per = ((per + 8 * JW - 1) / (8 * JW)) * (8 * JW);
    const uint64_t chunks = (n_blocks + per - 1) / per;
    hipLaunchKernelGGL(k_aes_z64_masks<QW>, dim3((unsigned)(chunks * n_qg)), dim3(512), 0, st, d_rk, d_keep, NQ, first_block, n_blocks,
                       (uint32_t)per, d_masks64);
}

void launch_aes_z64_masks(hipStream_t st, const uint32_t* d_rk, const uint32_t* d_keep, uint32_t NQ, uint64_t n_blocks,
                          uint64_t* d_masks64, uint64_t first_block) {
    if (!n_blocks) return;
    if (NQ % 16 == 0)
        launch_z64_qw<16>(st, d_rk, d_keep, NQ, first_block, n_blocks, d_masks64);
    else if (NQ % 8 == 0)
        launch_z64_qw<8>(st, d_rk, d_keep, NQ, first_block, n_blocks, d_masks64);
    else
        launch_z64_qw<2>(st, d_rk, d_keep, NQ, first_block, n_blocks, d_masks64);
}

// ------------------------------------------------------------------------------------
// The Z64 prover's level with the cipher inside (internal.h: Z64FParams).  Replaces, for one dependency level,
// generator/share.rs:54-65 + z64/domain.rs:64-83 (the two fresh masks of every Mul) AND interpreter/single.rs:25-157
// instantiated at Z64 (op_mul, the linear ops, Input, AssertZero) with transcript/prover.rs:181-232's records.
//
// Lane = (gate sub-index jsub, quad word ql of the workgroup's 16): after the cipher and four 32x32 bit transposes it holds
// lambda_ab and lambda_new of ITS Mul gate for the 4 repetitions x 8 players of quad q -- 32 slots that are 256 contiguous
// bytes of every share row, so operand rows are read and result rows written in 16-byte pieces, and the sum over a
// repetition's players is eight lane-local adds.  Linear gates and the few Input / AssertZero / Const gates of the level use
// the same mapping (memory only) and are dealt to the wavefronts between their cipher batches.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void z_ld16(const uint64_t* p, uint64_t& a, uint64_t& b) {
    const ulonglong2 v = *(const ulonglong2*)p;
    a = v.x;
    b = v.y;
}
__device__ __forceinline__ void z_st16(uint64_t* p, uint64_t a, uint64_t b) { *(ulonglong2*)p = make_ulonglong2(a, b); }
__device__ __forceinline__ const uint64_t* z_row(const Z64FParams& p, uint32_t ref, uint64_t S) {
    return (ref & G64_MASK_ROW) ? p.masks + (size_t)(ref & ~G64_MASK_ROW) * S : p.wmask + (size_t)ref * S;
}
// one repetition's eight transcript words (64 bytes; the stream is only 8-byte aligned)
__device__ __forceinline__ void z_store_on(uint64_t* op, const uint64_t* w) {
    if (((uintptr_t)op & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) z_st16(op + 2 * i, w[2 * i], w[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) op[i] = w[i];
    }
}

// Row layout of this path ("swizzled"): inside a workgroup's block of QW = 16 quad words (512 u64), the 16-byte piece i (slots 2i,
// 2i + 1 of a quad word's 32) of quad ql sits at u64 offset i * 2 QW + ql * 2 -- piece-major, so that the 16 lanes of a gate
// read 256 contiguous bytes per load instruction while each lane still collects the 32 slots of ITS quad word (the bitsliced
// cipher fixes which lane holds which slot).  With a lane's pieces in the natural order (256 contiguous bytes per lane, 256
// bytes apart between lanes) every load instruction touched 64 lines and the eight wavefronts of a compute unit evicted each
// other's lines from L1 between the instructions that shared them: 72 ms per proof instead of the 52 of the two-kernel path.
// Every row this path reads it also wrote (wmask rows, the Mul gates' lambda_new rows) -- except the Input gates' mask rows,
// which come from k_aes_z64_masks in the natural order and are rewritten in place by the Input gate itself (z64f_oth).
// (QW = quad words per workgroup block: 16, or 8 for shards whose rows are not a multiple of 16 quad words -- 32 repetitions)
template <int QW>
__device__ __forceinline__ uint32_t z_piece(uint32_t i) { return i * (2 * QW); }

template <int QW, bool VERIFY>
__device__ __forceinline__ void z64f_mul(const Gate64* __restrict__ gates, uint32_t gi, bool valid, const Z64FParams& p, const uint32_t* rkl, uint32_t q,
                                         uint32_t zo, bool writer, uint32_t kp) {
    const uint64_t S = (uint64_t)p.NQ * 32;
    const uint32_t m = valid ? gates[gi].m : 0u;
    uint32_t lo0[32], hi0[32], lo1[32], hi1[32];
    {
        uint32_t s[128], t[128];
        rounds_0_to_9<QW>(p.first_block + (m >> 1), s, t, rkl);
        sub_shift(s, t);
        const uint32_t* rk10 = rkl + 10 * 128 * QW;
        // plane 8*i + k = bit k of keystream byte i; u64 h, bit b  <->  plane 64*h + b (k_aes_z64_masks)
#pragma unroll
        for (int k = 0; k < 32; k++) {
            lo0[k] = t[31 - k] ^ rk10[(31 - k) * QW];
            hi0[k] = t[32 + 31 - k] ^ rk10[(32 + 31 - k) * QW];
            lo1[k] = t[64 + 31 - k] ^ rk10[(64 + 31 - k) * QW];
            hi1[k] = t[96 + 31 - k] ^ rk10[(96 + 31 - k) * QW];
        }
    }
    transpose32(lo0);
    transpose32(hi0);
    transpose32(lo1);
    transpose32(hi1);
    if (VERIFY) {  // the omitted player's stream of an opened repetition stays zero (k_aes_z64_masks: keep)
#pragma unroll
        for (int sl = 0; sl < 32; sl++) {
            const uint32_t on = (uint32_t)0 - ((kp >> (31 - sl)) & 1u);
            lo0[sl] &= on;
            hi0[sl] &= on;
            lo1[sl] &= on;
            hi1[sl] &= on;
        }
    }
    // the cipher and the transposes end HERE: without these pins the compiler sinks the last round and the transposes into the
    // `valid` branch below and schedules them among the row loads (160 registers spilled, the spill reloads then wait for
    // every outstanding store)
#pragma unroll
    for (int k = 0; k < 32; k++) asm volatile("" : "+v"(lo0[k]), "+v"(hi0[k]), "+v"(lo1[k]), "+v"(hi1[k])::"memory");
    // ... and every address below is computed from here on: hoisted above the cipher as loop invariants (the transcript bases of
    // the lane's four repetitions) they were spilled, and each reload sat between two repetitions' stores, waiting for them
    asm volatile("" : "+v"(q), "+v"(zo), "+v"(gi));
    if (!valid) return;
    const uint32_t qc = q & 3u, qmh = (qc & 2u) ? ~0u : 0u, qml = (qc & 1u) ? ~0u : 0u;
    const uint32_t row = gi - (uint32_t)__builtin_amdgcn_readfirstlane((int)gi);  // (= lane / 16 when QW == 16; computed HERE, not carried through the cipher)
    const Gate64 g = gates[gi];
    const uint32_t ep0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.ep);
    const bool pre_run = QW == 16 && __builtin_amdgcn_ballot_w64(g.ep == ep0 + row) == ~0ull;  // (a wavefront with a gate missing: false)
    const uint64_t* ap = z_row(p, g.am, S) + zo;
    const uint64_t* bp = z_row(p, g.bm, S) + zo;
    uint64_t* lnp = p.masks + (size_t)(g.m + 1) * S + zo;
    const uint32_t R = p.NQ * 4;
    const uint64_t va = VERIFY ? 0 : p.v[g.a], vb = VERIFY ? 0 : p.v[g.b];
    // the verifier (verifier/online.rs:122-183, verifier/preprocess.rs:46-79 at Z64): public corrections per repetition, which
    // repetitions are opened and which player they hide
    uint64_t cxs[4] = {0, 0, 0, 0}, cys[4] = {0, 0, 0, 0};
    uint32_t om4 = 0x08080808u;
    uint64_t dcs[4];
    // Software-pipelined over the lane's four repetitions: the operand pieces of repetition k + 2 are requested BEFORE the
    // stores of repetition k are issued.  CDNA has one in-order counter for vector loads and stores, so waiting for a load
    // also waits for every store issued before it -- with load / compute / store / load ... in program order each repetition
    // paid a load round trip AND a store round trip, and the eight wavefronts of a compute unit kept it at a third of the
    // row traffic the two-kernel path reaches (measured without the cipher: 80 ms per proof against 25).
    uint64_t lx[2][8], ly[2][8];
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            z_ld16(ap + z_piece<QW>(4 * k + i), lx[k][2 * i], lx[k][2 * i + 1]);
            z_ld16(bp + z_piece<QW>(4 * k + i), ly[k][2 * i], ly[k][2 * i + 1]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // lambda_new goes out at once (it is only stored), what stays is lambda_ab - lambda_new per slot and the four sums of lambda_ab:
    // 72 registers instead of 128 beside the operand pieces
    uint64_t d[32], cs[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint64_t lab = ((uint64_t)hi0[8 * k + i] << 32) | lo0[8 * k + i];
            const uint64_t lnw = ((uint64_t)hi1[8 * k + i] << 32) | lo1[8 * k + i];
            c += lab;
            d[8 * k + i] = lab - lnw;
        }
        cs[k] = c;
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
        z_st16(lnp + z_piece<QW>(i), ((uint64_t)hi1[2 * i] << 32) | lo1[2 * i], ((uint64_t)hi1[2 * i + 1] << 32) | lo1[2 * i + 1]);
    if (VERIFY) {  // (requested here, once lambda_new has left and freed its registers)
        z_ld16(p.wcorr + (size_t)g.a * R + 4 * q, cxs[0], cxs[1]);
        z_ld16(p.wcorr + (size_t)g.a * R + 4 * q + 2, cxs[2], cxs[3]);
        z_ld16(p.wcorr + (size_t)g.b * R + 4 * q, cys[0], cys[1]);
        z_ld16(p.wcorr + (size_t)g.b * R + 4 * q + 2, cys[2], cys[3]);
        om4 = *(const uint32_t*)(p.omit + 4 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int bf = k & 1;
        uint64_t w[8];
        uint64_t a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            a += lx[bf][i];
            b += ly[bf][i];
        }
        // corr = value - reconstruct(mask) (prover.rs:181-199 with the cleartext value known)
        const uint64_t cx = VERIFY ? cxs[k] : va - a, cy = VERIFY ? cys[k] : vb - b;
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = ly[bf][i] * cx + lx[bf][i] * cy + d[8 * k + i];
        uint64_t delta = a * b - cs[k];
        if (VERIFY) {
            const uint32_t om = (om4 >> (8 * k)) & 0xFFu;
            uint64_t rec = 0;
            if (om < 8) {  // opened: the supplied correction, and the hidden player's broadcast share from the proof
                const uint32_t rep_ = 4 * q + k;
                delta = p.sup_corr[(size_t)g.xc * p.sup_r + rep_];
                const uint64_t sup = p.sup_rec[(size_t)g.x * p.sup_r + rep_];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if ((uint32_t)i == om) w[i] += sup;
                    rec += w[i];
                }
            }
            dcs[k] = rec + delta + cx * cy;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k + 2 < 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                z_ld16(ap + z_piece<QW>(4 * (k + 2) + i), lx[bf][2 * i], lx[bf][2 * i + 1]);
                z_ld16(bp + z_piece<QW>(4 * (k + 2) + i), ly[bf][2 * i], ly[bf][2 * i + 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t rep = 4 * q + k;
        // (whole 64-byte segments per four lanes = four neighbouring quad words' repetition k: z64_dev.h)
        z4_store_on_quad(p.on + (size_t)(4 * (q & ~3u) + k) * p.on_words + g.eo, 4 * p.on_words, w, qc, qmh, qml);
        // the preprocessing transcript: one word per gate and repetition.  The wavefront's four gates (its four rows of 16 lanes) are
        // consecutive Mul gates of the level; where their words are consecutive in the streams too (pre_run: always, unless gates of
        // other levels sit between them in program order) the row of lanes k collects the four gates' words of repetition 4q + k and
        // stores 32 contiguous bytes, instead of 64 lanes storing 8 bytes each into 64 x 4 places
        if (pre_run) {
            uint64_t v4[4];
            z4_row_gather(delta, v4);
            if (row == (uint32_t)k) {
                uint64_t* pp = p.pre + (size_t)rep * p.pre_words + ep0;
                z4_st16_stream(pp, v4[0], v4[1]);
                z4_st16_stream(pp + 2, v4[2], v4[3]);
            }
        } else {
            p.pre[(size_t)rep * p.pre_words + g.ep] = delta;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (VERIFY) {
        z_st16(p.wcorr + (size_t)g.dst * R + 4 * q, dcs[0], dcs[1]);
        z_st16(p.wcorr + (size_t)g.dst * R + 4 * q + 2, dcs[2], dcs[3]);
    } else if (writer) {
        p.v[g.dst] = va * vb;
    }
}

// Add / Sub / AddConst / SubConst / MulConst: the mask row (z64/share.rs:110-136 player by player; elementwise, so the
// row layout does not matter) and the value
template <int QW, bool VERIFY>
__device__ __forceinline__ void z64f_lin(const Gate64& g, const Z64FParams& p, uint32_t q, uint32_t zo, bool writer) {
    const uint64_t S = (uint64_t)p.NQ * 32;
    const uint64_t* ap = z_row(p, g.am, S) + zo;
    uint64_t* dp = p.wmask + (size_t)g.dst * S + zo;
    if (VERIFY) {  // the public corrections of the lane's four repetitions (z64/recon.rs arithmetic)
        const uint32_t R = p.NQ * 4;
        uint64_t a4[4], b4[4] = {0, 0, 0, 0};
        z_ld16(p.wcorr + (size_t)g.a * R + 4 * q, a4[0], a4[1]);
        z_ld16(p.wcorr + (size_t)g.a * R + 4 * q + 2, a4[2], a4[3]);
        if (g.op == G64_ADD || g.op == G64_SUB) {
            z_ld16(p.wcorr + (size_t)g.b * R + 4 * q, b4[0], b4[1]);
            z_ld16(p.wcorr + (size_t)g.b * R + 4 * q + 2, b4[2], b4[3]);
        }
        uint64_t r4[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            r4[k] = g.op == G64_ADD ? a4[k] + b4[k] : g.op == G64_SUB ? a4[k] - b4[k] : g.op == G64_ADDC ? a4[k] + g.imm : g.op == G64_SUBC ? a4[k] - g.imm : a4[k] * g.imm;
        z_st16(p.wcorr + (size_t)g.dst * R + 4 * q, r4[0], r4[1]);
        z_st16(p.wcorr + (size_t)g.dst * R + 4 * q + 2, r4[2], r4[3]);
        writer = false;
    }
    const uint64_t va = VERIFY ? 0 : p.v[g.a];
    if (g.op == G64_ADD || g.op == G64_SUB) {
        const uint64_t* bp = z_row(p, g.bm, S) + zo;
        const uint64_t vb = VERIFY ? 0 : p.v[g.b];
        const bool sub = g.op == G64_SUB;
        uint64_t x[32], y[32];  // (every load before the first store: see z64f_mul)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            z_ld16(ap + z_piece<QW>(i), x[2 * i], x[2 * i + 1]);
            z_ld16(bp + z_piece<QW>(i), y[2 * i], y[2 * i + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; i++)
            z_st16(dp + z_piece<QW>(i), sub ? x[2 * i] - y[2 * i] : x[2 * i] + y[2 * i], sub ? x[2 * i + 1] - y[2 * i + 1] : x[2 * i + 1] + y[2 * i + 1]);
        if (writer) p.v[g.dst] = sub ? va - vb : va + vb;
    } else {
        const uint64_t f = g.op == G64_MULC ? g.imm : 1;
        uint64_t x[32];
#pragma unroll
        for (int i = 0; i < 16; i++) z_ld16(ap + z_piece<QW>(i), x[2 * i], x[2 * i + 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; i++) z_st16(dp + z_piece<QW>(i), x[2 * i] * f, x[2 * i + 1] * f);
        if (writer) p.v[g.dst] = g.op == G64_MULC ? va * g.imm : (g.op == G64_ADDC ? va + g.imm : va - g.imm);
    }
}

// Input (masked input = witness - reconstruct(fresh mask) into the online transcript; its mask row, written by
// k_aes_z64_masks in the natural order, is rewritten in this path's layout), AssertZero (the wire's mask shares into the
// online transcript; the VALUE must be zero, prover.rs:221-228), Const
template <int QW, bool VERIFY>
__device__ __forceinline__ void z64f_oth(const Gate64& g, const Z64FParams& p, uint32_t q, uint32_t zo, bool writer) {
    const uint64_t S = (uint64_t)p.NQ * 32;
    const uint32_t R = p.NQ * 4;
    const uint32_t om4 = VERIFY ? *(const uint32_t*)(p.omit + 4 * q) : 0x08080808u;
    if (VERIFY) writer = false;
    if (g.op == G64_INPUT) {
        uint64_t* row = p.masks + (size_t)g.m * S;
        const uint64_t* lp = row + (size_t)q * 32;
        const uint64_t wv = VERIFY ? 0 : p.wit[g.x];
        uint64_t l[32];
#pragma unroll
        for (int i = 0; i < 16; i++) z_ld16(lp + 2 * i, l[2 * i], l[2 * i + 1]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint64_t a = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) a += l[8 * k + i];
            if (VERIFY) {  // the masked input comes with the proof (opened repetitions; the others: junk zero, preprocess.rs:46-79)
                const uint32_t om = (om4 >> (8 * k)) & 0xFFu;
                const uint64_t corr = om < 8 ? p.sup_in[(size_t)g.x * p.sup_r + 4 * q + k] : 0;
                p.wcorr[(size_t)g.dst * R + 4 * q + k] = corr;
                p.on[(size_t)(4 * q + k) * p.on_words + g.eo] = corr;
            } else {
                p.on[(size_t)(4 * q + k) * p.on_words + g.eo] = wv - a;
            }
        }
        // every lane of the wavefront has its 256 bytes before any lane overwrites them (the 16 lanes of a gate exchange places
        // inside one 4 KiB block, and they sit in one wavefront)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; i++) z_st16(row + zo + z_piece<QW>(i), l[2 * i], l[2 * i + 1]);
        if (writer) p.v[g.dst] = wv;
    } else if (g.op == G64_ASSERT) {
        const uint64_t* ap = z_row(p, g.am, S) + zo;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint64_t l[8];
#pragma unroll
            for (int i = 0; i < 4; i++) z_ld16(ap + z_piece<QW>(4 * k + i), l[2 * i], l[2 * i + 1]);
            if (VERIFY) {
                const uint32_t om = (om4 >> (8 * k)) & 0xFFu;
                if (om < 8) {
                    const uint64_t sup = p.sup_rec[(size_t)g.x * p.sup_r + 4 * q + k];
                    uint64_t v = p.wcorr[(size_t)g.a * R + 4 * q + k];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if ((uint32_t)i == om) l[i] += sup;
                        v += l[i];
                    }
                    // online.rs:175-177: okay &= recon.is_zero() -- the reference never reads it; RV_VERIFY_STRICT does
                    if (v != 0) atomicOr(p.err, RV_DEV_ZERO_CHECK);
                }
            }
            z_store_on(p.on + (size_t)(4 * q + k) * p.on_words + g.eo, l);
        }
        if (writer && p.v[g.a] != 0) atomicOr(p.err, RV_E_WITNESS_INVALID);
    } else if (g.op == G64_CONST) {
        uint64_t* dp = p.wmask + (size_t)g.dst * S + zo;
#pragma unroll
        for (int i = 0; i < 16; i++) z_st16(dp + z_piece<QW>(i), 0, 0);
        if (VERIFY) {
            z_st16(p.wcorr + (size_t)g.dst * R + 4 * q, g.imm, g.imm);
            z_st16(p.wcorr + (size_t)g.dst * R + 4 * q + 2, g.imm, g.imm);
        }
        if (writer) p.v[g.dst] = g.imm;
    }
}

template <int QW, bool VERIFY>
__global__ __launch_bounds__(512, 2) void k_z64_fused(const Gate64* __restrict__ gates, Z64FLevel lv, uint32_t mul_per, uint32_t lin_per,
                                                    uint32_t oth_per, uint32_t lin_bias, Z64FParams p) {
    __shared__ uint32_t lds_rk[11 * 128 * QW];
    constexpr uint32_t JW = 64 / QW, STEP = 8 * JW;  // gates per wavefront and per workgroup iteration
    const uint32_t n_qg = p.qgn;
    const uint32_t qg = p.qg0 + blockIdx.x % n_qg, chunk = blockIdx.x / n_qg;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ql = lane % QW, jsub = lane / QW;
    const uint32_t q = qg * QW + ql;
    const uint32_t zo = qg * (QW * 32) + ql * 2;  // this lane's piece 0 inside a row (u64 units)
    const bool writer = q == 0;
    const uint32_t kp = (VERIFY && p.keep) ? p.keep[q] : 0xFFFFFFFFu;
    const uint32_t m_lo = min(lv.mul0 + chunk * mul_per, lv.mul1), m_hi = min(m_lo + mul_per, lv.mul1);
    const uint32_t l_lo = min(lv.mul1 + chunk * lin_per, lv.lin1), l_hi = min(l_lo + lin_per, lv.lin1);
    const uint32_t o_lo = min(lv.lin1 + chunk * oth_per, lv.oth1), o_hi = min(o_lo + oth_per, lv.oth1);
    if (m_hi > m_lo) stage_round_keys<QW>(p.rk, p.NQ, qg, lds_rk);  // (uniform over the workgroup)
    const uint32_t* rkl = lds_rk + ql;
    // Work of the workgroup's eight wavefronts in wavefront steps of JW gates.  Mul steps are dealt round-robin, so TM % 8 wavefronts
    // run one step more than the others -- and a level of the benchmark circuit has 8 192 +- 64 Mul gates, i.e. on half the levels
    // a few wavefronts of the chip ran a FIFTH cipher batch while every other one idled for it (5 x 65 us instead of 4).  The
    // linear steps (memory only, ~1/5 of a Mul step: lin_bias) make up for it: the wavefronts with the extra Mul step take that many
    // fewer of them.
    const uint32_t n_m = m_hi - m_lo, n_l = l_hi - l_lo;
    const uint32_t TM = (n_m + JW - 1) / JW, TL = (n_l + JW - 1) / JW;
    const uint32_t x = TM % 8;  // wavefronts 0 .. x - 1 run TM / 8 + 1 Mul steps
    const uint32_t MI = TM / 8 + (wave < x ? 1u : 0u);
    uint32_t ql_ = (TL + lin_bias * x + 7) / 8;                    // light wavefronts' linear steps; heavy ones: lin_bias fewer
    if (ql_ < lin_bias) ql_ = (TL + (8 - x) - 1) / (8 - x);        // (not enough linear work to even it out: the light ones take all)
    const uint32_t qh_ = ql_ > lin_bias ? ql_ - lin_bias : 0u;
    const uint32_t LI = wave < x ? qh_ : ql_;
    const uint32_t l0s = wave < x ? wave * qh_ : x * qh_ + (wave - x) * ql_;  // this wavefront's first linear step
    uint32_t ld = 0;
    for (uint32_t it = 0; it < MI; it++) {
        const uint32_t gi = m_lo + (wave + 8 * it) * JW + jsub;
        z64f_mul<QW, VERIFY>(gates, gi, gi < m_hi, p, rkl, q, zo, writer, kp);
        const uint32_t lend = (uint32_t)(((uint64_t)(it + 1) * LI) / MI);
        for (; ld < lend; ld++) {
            const uint32_t gl = l_lo + (l0s + ld) * JW + jsub;
            if (gl < l_hi) z64f_lin<QW, VERIFY>(gates[gl], p, q, zo, writer);
        }
    }
    for (; ld < LI; ld++) {
        const uint32_t gl = l_lo + (l0s + ld) * JW + jsub;
        if (gl < l_hi) z64f_lin<QW, VERIFY>(gates[gl], p, q, zo, writer);
    }
    for (uint32_t go = o_lo + wave * JW + jsub; go < o_hi; go += STEP) z64f_oth<QW, VERIFY>(gates[go], p, q, zo, writer);
}

// quad words per workgroup block: 16 where the rows are whole blocks of 16 (shards of 64 repetitions and more), else 8 (32-repetition
// shards, NQ = 8: VERDICT r4 #3), else 0 = this path does not take the shard
uint32_t z64_fused_qw(uint32_t NQ) { return NQ >= 16 && NQ % 16 == 0 ? 16u : (NQ >= 8 && NQ % 8 == 0 ? 8u : 0u); }
bool z64_fused_supports(uint32_t NQ) { return z64_fused_qw(NQ) != 0; }

template <int QW>
static void launch_z64_fused_qw(hipStream_t st, const Gate64* d_gates, const Z64FLevel& lv, const Z64FParams& p) {
    constexpr uint32_t JW = 64 / QW, STEP = 8 * JW;
    static const uint32_t cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return (uint32_t)n;
    }();
    // a Mul step (cipher batch + rows) in linear steps (rows only): what the wavefronts with one Mul step more get fewer of
    constexpr uint32_t lin_bias = 3;  // (measured 0 / 3 / 5 / 8: 38.6 / 38.3 / 38.9 / 38.8 ms)
    const uint32_t n_qg = p.qgn;  // (quad groups of this launch)
    const uint32_t n_mul = lv.mul1 - lv.mul0, n_lin = lv.lin1 - lv.mul1, n_oth = lv.oth1 - lv.lin1;
    if (!(n_mul + n_lin + n_oth) || !n_qg) return;
    auto up = [](uint64_t x, uint64_t m) { return (x + m - 1) / m * m; };
    // one generation of workgroups, one per compute unit (a workgroup owns it: 88 KiB of round keys, all registers), each with
    // an equal share of the level in whole wavefront steps; a level with little cipher work still gets enough workgroups for its
    // row traffic, one with very little of anything only as many as have a workgroup step to do
    const uint64_t per_qg = std::max<uint32_t>(cus / n_qg, 1);
    uint64_t chunks = n_mul ? std::min<uint64_t>(per_qg, ((uint64_t)n_mul + STEP - 1) / STEP) : 1;
    const uint64_t mem_chunks = std::min<uint64_t>(((uint64_t)n_lin + n_oth + 2 * STEP - 1) / (2 * STEP), per_qg * (n_mul ? 1u : 4u));
    chunks = std::max<uint64_t>(std::max(chunks, mem_chunks), 1);
    const uint32_t mul_per = (uint32_t)up((n_mul + chunks - 1) / chunks, JW), lin_per = (uint32_t)up((n_lin + chunks - 1) / chunks, JW),
                   oth_per = (uint32_t)up((n_oth + chunks - 1) / chunks, STEP);
    if (p.omit)
        hipLaunchKernelGGL((k_z64_fused<QW, true>), dim3((unsigned)(chunks * n_qg)), dim3(512), 0, st, d_gates, lv, mul_per, lin_per, oth_per, lin_bias, p);
    else
        hipLaunchKernelGGL((k_z64_fused<QW, false>), dim3((unsigned)(chunks * n_qg)), dim3(512), 0, st, d_gates, lv, mul_per, lin_per, oth_per, lin_bias, p);
}
void launch_z64_fused(hipStream_t st, const Gate64* d_gates, const Z64FLevel& lv, const Z64FParams& p) {
    if (z64_fused_qw(p.NQ) == 16)
        launch_z64_fused_qw<16>(st, d_gates, lv, p);
    else
        launch_z64_fused_qw<8>(st, d_gates, lv, p);
}

void launch_aes_blocks(hipStream_t st, const uint8_t* d_rkbytes, uint32_t n_keys, uint64_t first_block, uint64_t n_blocks,
                       uint8_t* d_out) {
    uint64_t n = (uint64_t)n_keys * n_blocks;
    if (!n) return;
    hipLaunchKernelGGL(k_aes_blocks, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_rkbytes, n_keys, first_block,
                       n_blocks, d_out);
}

}  // namespace rv
