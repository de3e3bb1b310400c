/* TEST INFRASTRUCTURE — CPU oracle, see rv_blake3.h / oracle/README.md. */
#include "rv_blake3.h"
#include <string.h>

enum { F_CHUNK_START = 1, F_CHUNK_END = 2, F_PARENT = 4, F_ROOT = 8 };

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                               0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

static const uint8_t SCHED[7][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
    {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1},
    {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
    {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4},
    {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
    {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13},
};

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

#define G(a, b, c, d, mx, my)        \
    do {                             \
        a = a + b + (mx);            \
        d = rotr(d ^ a, 16);         \
        c = c + d;                   \
        b = rotr(b ^ c, 12);         \
        a = a + b + (my);            \
        d = rotr(d ^ a, 8);          \
        c = c + d;                   \
        b = rotr(b ^ c, 7);          \
    } while (0)

/* full 16-word output of the compression function */
static void compress(const uint32_t cv[8], const uint32_t m[16], uint64_t t, uint32_t blen,
                     uint32_t flags, uint32_t out[16]) {
    uint32_t v[16];
    for (int i = 0; i < 8; i++) v[i] = cv[i];
    v[8] = IV[0];
    v[9] = IV[1];
    v[10] = IV[2];
    v[11] = IV[3];
    v[12] = (uint32_t)t;
    v[13] = (uint32_t)(t >> 32);
    v[14] = blen;
    v[15] = flags;
    for (int r = 0; r < 7; r++) {
        const uint8_t *s = SCHED[r];
        G(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
        G(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
        G(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
        G(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
        G(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
        G(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
        G(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
        G(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) {
        out[i] = v[i] ^ v[i + 8];
        out[i + 8] = v[i + 8] ^ cv[i];
    }
}

static void load_words(const uint8_t b[64], uint32_t m[16]) {
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) |
               ((uint32_t)b[4 * i + 3] << 24);
}

/* an "output" = the pending final compression of a node, so ROOT can be added late */
typedef struct {
    uint32_t cv[8];
    uint32_t m[16];
    uint64_t t;
    uint32_t blen;
    uint32_t flags;
} output_t;

static void output_cv(const output_t *o, uint32_t cv[8]) {
    uint32_t full[16];
    compress(o->cv, o->m, o->t, o->blen, o->flags, full);
    memcpy(cv, full, 32);
}

static void chunk_init(rvo_b3_chunk *c, uint64_t counter) {
    memcpy(c->cv, IV, 32);
    c->chunk_counter = counter;
    memset(c->buf, 0, 64);
    c->buf_len = 0;
    c->blocks_compressed = 0;
}

static size_t chunk_len(const rvo_b3_chunk *c) { return 64 * (size_t)c->blocks_compressed + c->buf_len; }

static uint32_t chunk_start_flag(const rvo_b3_chunk *c) { return c->blocks_compressed == 0 ? F_CHUNK_START : 0; }

static void chunk_update(rvo_b3_chunk *c, const uint8_t *in, size_t len) {
    while (len > 0) {
        if (c->buf_len == 64) {
            uint32_t m[16], full[16];
            load_words(c->buf, m);
            compress(c->cv, m, c->chunk_counter, 64, chunk_start_flag(c), full);
            memcpy(c->cv, full, 32);
            c->blocks_compressed++;
            c->buf_len = 0;
            memset(c->buf, 0, 64);
        }
        size_t want = 64 - c->buf_len;
        size_t take = len < want ? len : want;
        memcpy(c->buf + c->buf_len, in, take);
        c->buf_len += (uint8_t)take;
        in += take;
        len -= take;
    }
}

static void chunk_output(const rvo_b3_chunk *c, output_t *o) {
    memcpy(o->cv, c->cv, 32);
    load_words(c->buf, o->m);
    o->t = c->chunk_counter;
    o->blen = c->buf_len;
    o->flags = chunk_start_flag(c) | F_CHUNK_END;
}

static void parent_output(const uint32_t l[8], const uint32_t r[8], output_t *o) {
    memcpy(o->cv, IV, 32);
    memcpy(o->m, l, 32);
    memcpy(o->m + 8, r, 32);
    o->t = 0;
    o->blen = 64;
    o->flags = F_PARENT;
}

void rvo_blake3_init(rvo_blake3 *h) {
    chunk_init(&h->chunk, 0);
    h->stack_len = 0;
}

static void add_chunk_cv(rvo_blake3 *h, uint32_t cv[8], uint64_t total_chunks) {
    /* merge completed subtrees: one merge per trailing zero bit of total_chunks */
    while ((total_chunks & 1) == 0) {
        output_t o;
        parent_output(h->stack[h->stack_len - 1], cv, &o);
        output_cv(&o, cv);
        h->stack_len--;
        total_chunks >>= 1;
    }
    memcpy(h->stack[h->stack_len++], cv, 32);
}

#if defined(__AVX2__)
/* Eight whole chunks at once, one chunk per 32-bit lane -- what the `blake3` crate's SIMD back ends do with the 64 KiB
 * slabs Reverie's BufferedHasher hands them (crypto/hash.rs:36-51).  in: 8 x 1024 bytes, chunk counters t0 .. t0+7;
 * out[i] = chaining value of chunk i.  Message words are gathered by an 8x8 transpose per half block. */
#include <immintrin.h>
static inline __m256i rot16(__m256i x) {
    return _mm256_shuffle_epi8(x, _mm256_setr_epi8(2, 3, 0, 1, 6, 7, 4, 5, 10, 11, 8, 9, 14, 15, 12, 13, 2, 3, 0, 1, 6, 7, 4, 5, 10, 11, 8, 9, 14,
                                                   15, 12, 13));
}
static inline __m256i rot8(__m256i x) {
    return _mm256_shuffle_epi8(x, _mm256_setr_epi8(1, 2, 3, 0, 5, 6, 7, 4, 9, 10, 11, 8, 13, 14, 15, 12, 1, 2, 3, 0, 5, 6, 7, 4, 9, 10, 11, 8, 13,
                                                   14, 15, 12));
}
static inline __m256i rot12(__m256i x) { return _mm256_or_si256(_mm256_srli_epi32(x, 12), _mm256_slli_epi32(x, 20)); }
static inline __m256i rot7(__m256i x) { return _mm256_or_si256(_mm256_srli_epi32(x, 7), _mm256_slli_epi32(x, 25)); }
#define G8(a, b, c, d, mx, my)                                    \
    do {                                                          \
        a = _mm256_add_epi32(_mm256_add_epi32(a, b), (mx));       \
        d = rot16(_mm256_xor_si256(d, a));                        \
        c = _mm256_add_epi32(c, d);                               \
        b = rot12(_mm256_xor_si256(b, c));                        \
        a = _mm256_add_epi32(_mm256_add_epi32(a, b), (my));       \
        d = rot8(_mm256_xor_si256(d, a));                         \
        c = _mm256_add_epi32(c, d);                               \
        b = rot7(_mm256_xor_si256(b, c));                         \
    } while (0)
static inline void transpose8(__m256i r[8]) {
    const __m256i a0 = _mm256_unpacklo_epi32(r[0], r[1]), a1 = _mm256_unpackhi_epi32(r[0], r[1]);
    const __m256i a2 = _mm256_unpacklo_epi32(r[2], r[3]), a3 = _mm256_unpackhi_epi32(r[2], r[3]);
    const __m256i a4 = _mm256_unpacklo_epi32(r[4], r[5]), a5 = _mm256_unpackhi_epi32(r[4], r[5]);
    const __m256i a6 = _mm256_unpacklo_epi32(r[6], r[7]), a7 = _mm256_unpackhi_epi32(r[6], r[7]);
    const __m256i b0 = _mm256_unpacklo_epi64(a0, a2), b1 = _mm256_unpackhi_epi64(a0, a2);
    const __m256i b2 = _mm256_unpacklo_epi64(a1, a3), b3 = _mm256_unpackhi_epi64(a1, a3);
    const __m256i b4 = _mm256_unpacklo_epi64(a4, a6), b5 = _mm256_unpackhi_epi64(a4, a6);
    const __m256i b6 = _mm256_unpacklo_epi64(a5, a7), b7 = _mm256_unpackhi_epi64(a5, a7);
    r[0] = _mm256_permute2x128_si256(b0, b4, 0x20);
    r[1] = _mm256_permute2x128_si256(b1, b5, 0x20);
    r[2] = _mm256_permute2x128_si256(b2, b6, 0x20);
    r[3] = _mm256_permute2x128_si256(b3, b7, 0x20);
    r[4] = _mm256_permute2x128_si256(b0, b4, 0x31);
    r[5] = _mm256_permute2x128_si256(b1, b5, 0x31);
    r[6] = _mm256_permute2x128_si256(b2, b6, 0x31);
    r[7] = _mm256_permute2x128_si256(b3, b7, 0x31);
}
static void hash8_chunks_avx2(const uint8_t *in, uint64_t t0, uint32_t out[8][8]) {
    __m256i cv[8];
    for (int i = 0; i < 8; i++) cv[i] = _mm256_set1_epi32((int)IV[i]);
    const __m256i tlo = _mm256_setr_epi32((int)(uint32_t)(t0 + 0), (int)(uint32_t)(t0 + 1), (int)(uint32_t)(t0 + 2), (int)(uint32_t)(t0 + 3),
                                          (int)(uint32_t)(t0 + 4), (int)(uint32_t)(t0 + 5), (int)(uint32_t)(t0 + 6), (int)(uint32_t)(t0 + 7));
    const __m256i thi = _mm256_setr_epi32((int)(uint32_t)((t0 + 0) >> 32), (int)(uint32_t)((t0 + 1) >> 32), (int)(uint32_t)((t0 + 2) >> 32),
                                          (int)(uint32_t)((t0 + 3) >> 32), (int)(uint32_t)((t0 + 4) >> 32), (int)(uint32_t)((t0 + 5) >> 32),
                                          (int)(uint32_t)((t0 + 6) >> 32), (int)(uint32_t)((t0 + 7) >> 32));
    for (int b = 0; b < 16; b++) {
        __m256i m[16];
        for (int half = 0; half < 2; half++) {
            __m256i r[8];
            for (int i = 0; i < 8; i++) r[i] = _mm256_loadu_si256((const __m256i *)(in + 1024 * i + 64 * b + 32 * half));
            transpose8(r);
            for (int i = 0; i < 8; i++) m[8 * half + i] = r[i];
        }
        __m256i v[16];
        for (int i = 0; i < 8; i++) v[i] = cv[i];
        v[8] = _mm256_set1_epi32((int)IV[0]);
        v[9] = _mm256_set1_epi32((int)IV[1]);
        v[10] = _mm256_set1_epi32((int)IV[2]);
        v[11] = _mm256_set1_epi32((int)IV[3]);
        v[12] = tlo;
        v[13] = thi;
        v[14] = _mm256_set1_epi32(64);
        v[15] = _mm256_set1_epi32((int)((b == 0 ? F_CHUNK_START : 0) | (b == 15 ? F_CHUNK_END : 0)));
        for (int r = 0; r < 7; r++) {
            const uint8_t *s = SCHED[r];
            G8(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
            G8(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
            G8(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
            G8(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
            G8(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
            G8(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
            G8(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
            G8(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) cv[i] = _mm256_xor_si256(v[i], v[i + 8]);
    }
    transpose8(cv); /* cv[i] lane w = word w of chunk i */
    for (int i = 0; i < 8; i++) _mm256_storeu_si256((__m256i *)out[i], cv[i]);
}
#endif

void rvo_blake3_update(rvo_blake3 *h, const void *data, size_t len) {
    const uint8_t *in = (const uint8_t *)data;
#if defined(__AVX2__)
    /* whole chunks eight at a time while MORE input follows them (a chunk may only be closed once it is known not to
     * be the last one: the root needs its own flag) */
    if (len > 0 && chunk_len(&h->chunk) == 1024) { /* a full chunk was waiting to learn that more input follows */
        output_t o;
        uint32_t cv[8];
        chunk_output(&h->chunk, &o);
        output_cv(&o, cv);
        const uint64_t total = h->chunk.chunk_counter + 1;
        add_chunk_cv(h, cv, total);
        chunk_init(&h->chunk, total);
    }
    while (chunk_len(&h->chunk) == 0 && len > 8 * 1024) {
        uint32_t cvs[8][8];
        const uint64_t t0 = h->chunk.chunk_counter;
        hash8_chunks_avx2(in, t0, cvs);
        for (int i = 0; i < 8; i++) add_chunk_cv(h, cvs[i], t0 + (uint64_t)i + 1);
        chunk_init(&h->chunk, t0 + 8);
        in += 8 * 1024;
        len -= 8 * 1024;
    }
#endif
    while (len > 0) {
        if (chunk_len(&h->chunk) == 1024) {
            output_t o;
            uint32_t cv[8];
            chunk_output(&h->chunk, &o);
            output_cv(&o, cv);
            uint64_t total = h->chunk.chunk_counter + 1;
            add_chunk_cv(h, cv, total);
            chunk_init(&h->chunk, total);
        }
        size_t want = 1024 - chunk_len(&h->chunk);
        size_t take = len < want ? len : want;
        chunk_update(&h->chunk, in, take);
        in += take;
        len -= take;
    }
}

static void root_output(const rvo_blake3 *h, output_t *o) {
    chunk_output(&h->chunk, o);
    int remaining = h->stack_len;
    while (remaining > 0) {
        uint32_t cv[8];
        output_cv(o, cv);
        remaining--;
        parent_output(h->stack[remaining], cv, o);
    }
}

void rvo_blake3_finalize_xof(const rvo_blake3 *h, uint64_t seek, uint8_t *out, size_t len) {
    output_t o;
    root_output(h, &o);
    uint64_t blk = seek / 64;
    size_t off = (size_t)(seek % 64);
    while (len > 0) {
        uint32_t full[16];
        uint8_t bytes[64];
        compress(o.cv, o.m, blk, o.blen, o.flags | F_ROOT, full);
        for (int i = 0; i < 16; i++) {
            bytes[4 * i] = (uint8_t)full[i];
            bytes[4 * i + 1] = (uint8_t)(full[i] >> 8);
            bytes[4 * i + 2] = (uint8_t)(full[i] >> 16);
            bytes[4 * i + 3] = (uint8_t)(full[i] >> 24);
        }
        size_t take = 64 - off;
        if (take > len) take = len;
        memcpy(out, bytes + off, take);
        out += take;
        len -= take;
        off = 0;
        blk++;
    }
}

void rvo_blake3_finalize(const rvo_blake3 *h, uint8_t out[32]) { rvo_blake3_finalize_xof(h, 0, out, 32); }

void rvo_blake3_hash(const void *data, size_t len, uint8_t out[32]) {
    rvo_blake3 h;
    rvo_blake3_init(&h);
    rvo_blake3_update(&h, data, len);
    rvo_blake3_finalize(&h, out);
}
