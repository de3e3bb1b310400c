// BLAKE3 compression function shared by device kernels and the host-side Fiat-Shamir code.
// Replaces the `blake3` crate calls at /root/reference/src/crypto/hash.rs:14-57 and
// src/crypto/ro.rs:8-20 (plain unkeyed hash + XOF).  Own implementation of the published
// BLAKE3 specification; checked against the oracle and golden vectors in tests/.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RV_HD __host__ __device__ __forceinline__
#else
#define RV_HD inline
#endif

namespace b3 {

enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

#define B3_IV0 0x6A09E667u
#define B3_IV1 0xBB67AE85u
#define B3_IV2 0x3C6EF372u
#define B3_IV3 0xA54FF53Au
#define B3_IV4 0x510E527Fu
#define B3_IV5 0x9B05688Cu
#define B3_IV6 0x1F83D9ABu
#define B3_IV7 0x5BE0CD19u

RV_HD uint32_t rotr(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(x, x, n);
#else
    return (x >> n) | (x << (32 - n));
#endif
}

#define B3_G(a, b, c, d, mx, my) \
    a = a + b + (mx);            \
    d = rotr(d ^ a, 16);         \
    c = c + d;                   \
    b = rotr(b ^ c, 12);         \
    a = a + b + (my);            \
    d = rotr(d ^ a, 8);          \
    c = c + d;                   \
    b = rotr(b ^ c, 7);

#define B3_ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
    B3_G(v0, v4, v8, v12, m0, m1)                                                      \
    B3_G(v1, v5, v9, v13, m2, m3)                                                      \
    B3_G(v2, v6, v10, v14, m4, m5)                                                     \
    B3_G(v3, v7, v11, v15, m6, m7)                                                     \
    B3_G(v0, v5, v10, v15, m8, m9)                                                     \
    B3_G(v1, v6, v11, v12, m10, m11)                                                   \
    B3_G(v2, v7, v8, v13, m12, m13)                                                    \
    B3_G(v3, v4, v9, v14, m14, m15)

// Full compression: cv[8] (in), m[16], counter t, block length, flags -> out[16].
// out[0..8) is the new chaining value; out[8..16) is only needed for XOF output.
template <bool FULL>
RV_HD void compress(const uint32_t cv[8], const uint32_t m[16], uint64_t t, uint32_t blen, uint32_t flags,
                    uint32_t* out) {
    uint32_t v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
    uint32_t v8 = B3_IV0, v9 = B3_IV1, v10 = B3_IV2, v11 = B3_IV3;
    uint32_t v12 = (uint32_t)t, v13 = (uint32_t)(t >> 32), v14 = blen, v15 = flags;
    // message schedule: permutation [2,6,3,10,7,0,4,13,1,11,12,5,9,14,15,8] applied per round
    B3_ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15])
    B3_ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8])
    B3_ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1])
    B3_ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6])
    B3_ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4])
    B3_ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7])
    B3_ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13])
    out[0] = v0 ^ v8;
    out[1] = v1 ^ v9;
    out[2] = v2 ^ v10;
    out[3] = v3 ^ v11;
    out[4] = v4 ^ v12;
    out[5] = v5 ^ v13;
    out[6] = v6 ^ v14;
    out[7] = v7 ^ v15;
    if (FULL) {
        out[8] = v8 ^ cv[0];
        out[9] = v9 ^ cv[1];
        out[10] = v10 ^ cv[2];
        out[11] = v11 ^ cv[3];
        out[12] = v12 ^ cv[4];
        out[13] = v13 ^ cv[5];
        out[14] = v14 ^ cv[6];
        out[15] = v15 ^ cv[7];
    }
}

// N independent compressions in lockstep (same counter / length / flags, different cv and message):
// the G function is one 12-deep dependency chain, so a single compression only offers 4-way ILP per
// half round; interleaving N of them keeps the VALU issuing (the transcript kernels were stalled on
// instruction issue for half of their cycles with one compression at a time).
#define B3_GN(a, b, c, d, mx, my)                 \
    _Pragma("unroll") for (int i_ = 0; i_ < N; i_++) { \
        B3_G(v[i_][a], v[i_][b], v[i_][c], v[i_][d], m[i_][mx], m[i_][my])                        \
    }
#define B3_ROUNDN(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B3_GN(0, 4, 8, 12, s0, s1)                                                          \
    B3_GN(1, 5, 9, 13, s2, s3)                                                          \
    B3_GN(2, 6, 10, 14, s4, s5)                                                         \
    B3_GN(3, 7, 11, 15, s6, s7)                                                         \
    B3_GN(0, 5, 10, 15, s8, s9)                                                         \
    B3_GN(1, 6, 11, 12, s10, s11)                                                       \
    B3_GN(2, 7, 8, 13, s12, s13)                                                        \
    B3_GN(3, 4, 9, 14, s14, s15)

template <int N>
RV_HD void compress_n(uint32_t cv[N][8], const uint32_t m[N][16], uint64_t t, uint32_t blen, uint32_t flags) {
    uint32_t v[N][16];
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[i][k] = cv[i][k];
        v[i][8] = B3_IV0;
        v[i][9] = B3_IV1;
        v[i][10] = B3_IV2;
        v[i][11] = B3_IV3;
        v[i][12] = (uint32_t)t;
        v[i][13] = (uint32_t)(t >> 32);
        v[i][14] = blen;
        v[i][15] = flags;
    }
    B3_ROUNDN(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B3_ROUNDN(2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
    B3_ROUNDN(3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
    B3_ROUNDN(10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
    B3_ROUNDN(12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
    B3_ROUNDN(9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
    B3_ROUNDN(11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int k = 0; k < 8; k++) cv[i][k] = v[i][k] ^ v[i][k + 8];
}

RV_HD void iv(uint32_t cv[8]) {
    cv[0] = B3_IV0;
    cv[1] = B3_IV1;
    cv[2] = B3_IV2;
    cv[3] = B3_IV3;
    cv[4] = B3_IV4;
    cv[5] = B3_IV5;
    cv[6] = B3_IV6;
    cv[7] = B3_IV7;
}

// parent node: cv' = compress(IV, left||right, 0, 64, PARENT|extra)
RV_HD void parent(const uint32_t l[8], const uint32_t r[8], uint32_t extra_flags, uint32_t out[8]) {
    uint32_t m[16], c[8];
    for (int i = 0; i < 8; i++) {
        m[i] = l[i];
        m[8 + i] = r[i];
    }
    iv(c);
    compress<false>(c, m, 0, 64, PARENT | extra_flags, out);
}

// hash of exactly 64 bytes given as 16 LE words (one block, one chunk, root)
RV_HD void hash64(const uint32_t m[16], uint32_t out[8]) {
    uint32_t c[8];
    iv(c);
    compress<false>(c, m, 0, 64, CHUNK_START | CHUNK_END | ROOT, out);
}

#if defined(__HIPCC__)
// One compression on the FOUR lanes of a quad (lane & 3 = column c of the 4 x 4 state): a lone chain of compressions -- the 16 blocks of a
// chunk, the levels of a small tree -- is bound by the ~690 instructions of a compression issued one after the other, whatever the
// number of idle lanes beside it; a column per lane is ~200 per lane (the G function once per half round instead of four times,
// three quad-permute moves to turn columns into diagonals and three back).  The message is read from memory the four lanes share
// (LDS): word idx of block `msg` at msg[idx]; which words a lane needs depends on its column, so its 28 word indices are made once
// (quad_schedule) and reused for every block.  cva / cvb: chaining-value words c and 4 + c, in and out.  FULL: hi_a / hi_b receive
// output words 8 + c and 12 + c (XOF).
template <int CTRL>
__device__ __forceinline__ uint32_t quad_from(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
struct QuadSchedule {
    uint32_t idx[28];  // [round][column step x, y, diagonal step x, y]
};
__device__ __forceinline__ QuadSchedule quad_schedule(uint32_t c) {
    // the message schedule of compress(): row r = the 16 word indices of round r
    constexpr uint8_t S[7][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
                                  {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1}, {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
                                  {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4}, {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
                                  {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};
    QuadSchedule q;
#pragma unroll
    for (int r = 0; r < 7; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // column step: words 2c, 2c + 1 of the row; diagonal step: words 8 + 2c, 8 + 2c + 1
            const int base = (j >> 1) * 8 + (j & 1);
            const uint32_t k = (uint32_t)S[r][base] | ((uint32_t)S[r][base + 2] << 4) | ((uint32_t)S[r][base + 4] << 8) | ((uint32_t)S[r][base + 6] << 12);
            q.idx[4 * r + j] = (k >> (4 * c)) & 15u;
        }
    return q;
}
template <bool FULL>
__device__ __forceinline__ void compress_q(uint32_t& cva, uint32_t& cvb, const uint32_t* msg, const QuadSchedule& q, uint32_t c, uint64_t t, uint32_t blen,
                                           uint32_t flags, uint32_t* hi_a = nullptr, uint32_t* hi_b = nullptr) {
    uint32_t a = cva, b = cvb;
    uint32_t cc = c == 0 ? B3_IV0 : c == 1 ? B3_IV1 : c == 2 ? B3_IV2 : B3_IV3;
    uint32_t d = c == 0 ? (uint32_t)t : c == 1 ? (uint32_t)(t >> 32) : c == 2 ? blen : flags;
    uint32_t m[28];
#pragma unroll
    for (int k = 0; k < 28; k++) m[k] = msg[q.idx[k]];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        B3_G(a, b, cc, d, m[4 * r], m[4 * r + 1])
        // diagonals: lane c takes b of column c + 1, c of column c + 2, d of column c + 3
        b = quad_from<0x39>(b), cc = quad_from<0x4E>(cc), d = quad_from<0x93>(d);
        B3_G(a, b, cc, d, m[4 * r + 2], m[4 * r + 3])
        b = quad_from<0x93>(b), cc = quad_from<0x4E>(cc), d = quad_from<0x39>(d);
    }
    if (FULL) {
        *hi_a = cc ^ cva;
        *hi_b = d ^ cvb;
    }
    cva = a ^ cc;
    cvb = b ^ d;
}
#endif

}  // namespace b3

// ---- host-only incremental hasher (small inputs: commitment of 256 digests, random oracle) ----
#include <string.h>
#include <vector>
namespace b3 {

struct Hasher {
    std::vector<uint8_t> buf;  // the inputs on this path are <= 8 KiB: buffer everything
    void update(const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        buf.insert(buf.end(), b, b + n);
    }
    // root output node (cv, block words, counter base, blen, flags) for hash / XOF
    struct Out {
        uint32_t cv[8];
        uint32_t m[16];
        uint32_t blen;
        uint32_t flags;
    };
    static void words(const uint8_t* p, size_t n, uint32_t m[16]) {
        uint8_t blk[64] = {0};
        memcpy(blk, p, n);
        for (int i = 0; i < 16; i++)
            m[i] = (uint32_t)blk[4 * i] | ((uint32_t)blk[4 * i + 1] << 8) | ((uint32_t)blk[4 * i + 2] << 16) |
                   ((uint32_t)blk[4 * i + 3] << 24);
    }
    // chaining value (or pending root output) of chunk `c`
    void chunk(size_t c, size_t n_chunks, uint32_t cv_out[8], Out* root) const {
        size_t off = c * 1024;
        size_t len = buf.size() - off < 1024 ? buf.size() - off : 1024;
        size_t nblk = len == 0 ? 1 : (len + 63) / 64;
        uint32_t cv[8];
        iv(cv);
        for (size_t b = 0; b < nblk; b++) {
            size_t bl = (b + 1 < nblk) ? 64 : len - 64 * b;
            uint32_t m[16];
            words(buf.data() + off + 64 * b, bl, m);
            uint32_t fl = (b == 0 ? CHUNK_START : 0) | (b + 1 == nblk ? CHUNK_END : 0);
            if (b + 1 == nblk && n_chunks == 1 && root) {
                memcpy(root->cv, cv, 32);
                memcpy(root->m, m, 64);
                root->blen = (uint32_t)bl;
                root->flags = fl;
                return;
            }
            uint32_t o[8];
            compress<false>(cv, m, c, (uint32_t)bl, fl, o);
            memcpy(cv, o, 32);
        }
        memcpy(cv_out, cv, 32);
    }
    Out root() const {
        Out r;
        size_t n = buf.empty() ? 1 : (buf.size() + 1023) / 1024;
        if (n == 1) {
            uint32_t dummy[8];
            chunk(0, 1, dummy, &r);
            return r;
        }
        std::vector<uint32_t> cvs(n * 8);
        for (size_t c = 0; c < n; c++) chunk(c, n, &cvs[8 * c], nullptr);
        // pairwise reduction with odd-node promotion == BLAKE3's left-complete tree
        while (n > 2) {
            size_t nn = (n + 1) / 2;
            for (size_t i = 0; i < n / 2; i++) {
                uint32_t o[8];
                parent(&cvs[16 * i], &cvs[16 * i + 8], 0, o);
                memcpy(&cvs[8 * i], o, 32);
            }
            if (n & 1) memmove(&cvs[8 * (n / 2)], &cvs[8 * (n - 1)], 32);
            n = nn;
        }
        iv(r.cv);
        memcpy(r.m, &cvs[0], 64);
        r.blen = 64;
        r.flags = PARENT;
        return r;
    }
    void finalize(uint8_t out[32]) const { xof(0, out, 32); }
    void xof(uint64_t seek, uint8_t* out, size_t len) const {
        Out r = root();
        uint64_t blk = seek / 64;
        size_t off = (size_t)(seek % 64);
        while (len) {
            uint32_t o[16];
            compress<true>(r.cv, r.m, blk, r.blen, r.flags | ROOT, o);
            uint8_t bytes[64];
            for (int i = 0; i < 16; i++) {
                bytes[4 * i] = (uint8_t)o[i];
                bytes[4 * i + 1] = (uint8_t)(o[i] >> 8);
                bytes[4 * i + 2] = (uint8_t)(o[i] >> 16);
                bytes[4 * i + 3] = (uint8_t)(o[i] >> 24);
            }
            size_t take = 64 - off < len ? 64 - off : len;
            memcpy(out, bytes + off, take);
            out += take;
            len -= take;
            off = 0;
            blk++;
        }
    }
};

}  // namespace b3
