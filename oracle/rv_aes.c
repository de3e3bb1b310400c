/* TEST INFRASTRUCTURE — CPU oracle, see rv_aes.h / oracle/README.md. */
#include "rv_aes.h"
#include <string.h>

#if defined(__AES__) && defined(__SSE2__)
#include <wmmintrin.h>
#include <emmintrin.h>
#define RVO_AESNI 1
#endif

/* ---- S-box computed from its definition (inverse in GF(2^8) + affine map) ---- */
static uint8_t SBOX[256];
static int sbox_ready = 0;

static uint8_t gf_mul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    while (b) {
        if (b & 1) r ^= a;
        a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0));
        b >>= 1;
    }
    return r;
}

static void sbox_init(void) {
    if (sbox_ready) return;
    for (int x = 0; x < 256; x++) {
        /* inverse by exponentiation x^254 */
        uint8_t inv = 0;
        if (x) {
            uint8_t acc = 1, base = (uint8_t)x;
            int e = 254;
            while (e) {
                if (e & 1) acc = gf_mul(acc, base);
                base = gf_mul(base, base);
                e >>= 1;
            }
            inv = acc;
        }
        uint8_t s = inv;
        uint8_t rot = inv;
        for (int k = 0; k < 4; k++) {
            rot = (uint8_t)((rot << 1) | (rot >> 7));
            s ^= rot;
        }
        SBOX[x] = (uint8_t)(s ^ 0x63);
    }
    sbox_ready = 1;
}

void rvo_aes128_init(rvo_aes128 *ctx, const uint8_t key[16]) {
    sbox_init();
    memcpy(ctx->rk[0], key, 16);
    uint8_t rcon = 1;
    for (int r = 1; r <= 10; r++) {
        const uint8_t *p = ctx->rk[r - 1];
        uint8_t *q = ctx->rk[r];
        uint8_t t[4] = {SBOX[p[13]], SBOX[p[14]], SBOX[p[15]], SBOX[p[12]]};
        t[0] ^= rcon;
        rcon = (uint8_t)((rcon << 1) ^ ((rcon & 0x80) ? 0x1b : 0));
        for (int i = 0; i < 4; i++) q[i] = p[i] ^ t[i];
        for (int i = 4; i < 16; i++) q[i] = p[i] ^ q[i - 4];
    }
}

static void encrypt_portable(const rvo_aes128 *ctx, const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ ctx->rk[0][i];
    for (int r = 1; r <= 10; r++) {
        /* SubBytes + ShiftRows: state is column-major, s[4*c + row] */
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < 4; row++)
                t[4 * c + row] = SBOX[s[4 * ((c + row) & 3) + row]];
        if (r < 10) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                uint8_t all = a0 ^ a1 ^ a2 ^ a3;
                s[4 * c + 0] = a0 ^ all ^ gf_mul(a0 ^ a1, 2);
                s[4 * c + 1] = a1 ^ all ^ gf_mul(a1 ^ a2, 2);
                s[4 * c + 2] = a2 ^ all ^ gf_mul(a2 ^ a3, 2);
                s[4 * c + 3] = a3 ^ all ^ gf_mul(a3 ^ a0, 2);
            }
        } else {
            memcpy(s, t, 16);
        }
        for (int i = 0; i < 16; i++) s[i] ^= ctx->rk[r][i];
    }
    memcpy(out, s, 16);
}

void rvo_aes128_encrypt(const rvo_aes128 *ctx, const uint8_t in[16], uint8_t out[16]) {
#ifdef RVO_AESNI
    __m128i b = _mm_loadu_si128((const __m128i *)in);
    b = _mm_xor_si128(b, _mm_loadu_si128((const __m128i *)ctx->rk[0]));
    for (int r = 1; r < 10; r++)
        b = _mm_aesenc_si128(b, _mm_loadu_si128((const __m128i *)ctx->rk[r]));
    b = _mm_aesenclast_si128(b, _mm_loadu_si128((const __m128i *)ctx->rk[10]));
    _mm_storeu_si128((__m128i *)out, b);
#else
    encrypt_portable(ctx, in, out);
#endif
}

/* exposed so the tests can cross-check the AES-NI path against the portable one */
void rvo_aes128_encrypt_portable(const rvo_aes128 *ctx, const uint8_t in[16], uint8_t out[16]) {
    encrypt_portable(ctx, in, out);
}

static void ctr_block(uint64_t hi, uint64_t lo, uint8_t out[16]) {
    for (int i = 0; i < 8; i++) {
        out[i] = (uint8_t)(hi >> (56 - 8 * i));
        out[8 + i] = (uint8_t)(lo >> (56 - 8 * i));
    }
}

void rvo_prg_init(rvo_prg *prg, const uint8_t key[16]) {
    rvo_aes128_init(&prg->aes, key);
    prg->ctr_hi = 0;
    prg->ctr_lo = 0;
}

void rvo_prg_gen(rvo_prg *prg, uint8_t *dst, size_t len) {
    uint8_t blk[16];
    for (size_t off = 0; off + 16 <= len; off += 16) {
        ctr_block(prg->ctr_hi, prg->ctr_lo, blk);
        rvo_aes128_encrypt(&prg->aes, blk, dst + off);
        if (++prg->ctr_lo == 0) prg->ctr_hi++;
    }
}

void rvo_prg_block(const uint8_t key[16], uint64_t blk, uint8_t out[16]) {
    rvo_aes128 a;
    uint8_t in[16];
    rvo_aes128_init(&a, key);
    ctr_block(0, blk, in);
    rvo_aes128_encrypt(&a, in, out);
}
