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

void rvo_blake3_update(rvo_blake3 *h, const void *data, size_t len) {
    const uint8_t *in = (const uint8_t *)data;
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
