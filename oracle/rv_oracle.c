/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See rv_oracle.h for scope, provenance and the
 * pinning statement.  Structure mirrors the reference (one packed group = 8 repetitions
 * x 8 players walked through the whole gate stream), written in plain C. */
#include "rv_oracle.h"
#include "rv_aes.h"
#include "rv_blake3.h"

#include <pthread.h>
#if defined(__AVX2__)
#include <immintrin.h>
#endif
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

/* ------------------------------------------------------------------ small vectors */
typedef struct {
    u8 *p;
    size_t len, cap;
} bvec;

static int bvec_reserve(bvec *v, size_t extra) {
    if (v->len + extra <= v->cap) return 0;
    size_t nc = v->cap ? v->cap * 2 : 64;
    while (nc < v->len + extra) nc *= 2;
    u8 *np = (u8 *)realloc(v->p, nc);
    if (!np) return -1;
    v->p = np;
    v->cap = nc;
    return 0;
}
static int bvec_push(bvec *v, const void *src, size_t n) {
    if (bvec_reserve(v, n)) return -1;
    memcpy(v->p + v->len, src, n);
    v->len += n;
    return 0;
}
static int bvec_push_u8(bvec *v, u8 b) { return bvec_push(v, &b, 1); }
static int bvec_push_u64le(bvec *v, u64 x) {
    u8 b[8];
    for (int i = 0; i < 8; i++) b[i] = (u8)(x >> (8 * i));
    return bvec_push(v, b, 8);
}
static void bvec_free(bvec *v) {
    free(v->p);
    v->p = NULL;
    v->len = v->cap = 0;
}

/* ------------------------------------------------------------------ BufferedHasher
 * crypto/hash.rs:17-58.  The reference stages 64 KiB before each blake3 update, which lets the blake3 crate hash many
 * chunks per call with its SIMD back ends; the staging size does not change the digest.  Here the stage is 65 KiB: a
 * flush then hands rvo_blake3_update 64 KiB that are KNOWN not to be the end of the stream (one more KiB follows), so
 * all 64 chunks take the eight-chunks-at-a-time AVX2 path (rv_blake3.c). */
#define STAGE (65 * 1024)
typedef struct {
    rvo_blake3 h;
    u8 buf[STAGE];
    size_t n;
} bhasher;

static void bh_init(bhasher *b) {
    rvo_blake3_init(&b->h);
    b->n = 0;
}
static inline void bh_push(bhasher *b, u8 v) {
    b->buf[b->n++] = v;
    if (b->n == STAGE) {
        rvo_blake3_update(&b->h, b->buf, STAGE);
        b->n = 0;
    }
}
static inline void bh_update(bhasher *b, const u8 *p, size_t n) {
    for (size_t i = 0; i < n; i++) bh_push(b, p[i]);
}
static void bh_finalize(const bhasher *b, u8 out[32]) { /* non-consuming, hash.rs:53-57 */
    rvo_blake3 c = b->h;
    rvo_blake3_update(&c, b->buf, b->n);
    rvo_blake3_finalize(&c, out);
}

/* ------------------------------------------------------------------ expand_seed */
void rvo_expand_seed(const uint8_t seed[16], uint8_t keys[8][16]) { /* transcript/mod.rs:99-106 */
    rvo_prg prg;
    rvo_prg_init(&prg, seed);
    for (int p = 0; p < 8; p++) rvo_prg_gen(&prg, keys[p], 16);
}

/* ------------------------------------------------------------------ GF2 algebra */
uint64_t rvo_gf2_reconstruct(uint64_t t) { /* gf2/domain.rs:47-63 */
    t ^= t >> 4;
    t ^= t >> 2;
    t ^= t >> 1;
    t &= 0x0101010101010101ull;
    t |= t << 1;
    t |= t << 2;
    t |= t << 4;
    return t;
}
#define RECON2 rvo_gf2_reconstruct

/* byte_to_shares (gf2/domain.rs:293-378): 64 bytes (k = rep*8+player) -> 8 shares;
 * share j takes bit (7-j) of every byte, byte k lands at bit (63-k). */
static void byte_to_shares(u64 dst[8], const u8 src[64]) {
#if defined(__AVX2__)
    /* the reference's own method (gf2/domain.rs:293-378): movemask of the byte MSBs, then shift
     * every byte left by one (add to itself) — 8 rounds.  _mm256_set_epi8 puts src[0] in the top
     * lane, so movemask bit 31 <- src[0]; here the bytes are loaded in memory order and reversed. */
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6,
                                         5, 4, 3, 2, 1, 0);
    __m256i fst = _mm256_loadu_si256((const __m256i *)src);
    __m256i snd = _mm256_loadu_si256((const __m256i *)(src + 32));
    /* byte k must end at movemask bit (31 - k): reverse within 128-bit lanes, then swap the lanes */
    fst = _mm256_permute2x128_si256(_mm256_shuffle_epi8(fst, rev), _mm256_shuffle_epi8(fst, rev), 1);
    snd = _mm256_permute2x128_si256(_mm256_shuffle_epi8(snd, rev), _mm256_shuffle_epi8(snd, rev), 1);
    for (int j = 0; j < 8; j++) {
        const u64 top = (u32)_mm256_movemask_epi8(fst), bot = (u32)_mm256_movemask_epi8(snd);
        fst = _mm256_add_epi8(fst, fst);
        snd = _mm256_add_epi8(snd, snd);
        dst[j] = (top << 32) | bot;
    }
#else
    for (int j = 0; j < 8; j++) {
        u64 w = 0;
        for (int k = 0; k < 64; k++) w |= (u64)((src[k] >> (7 - j)) & 1) << (63 - k);
        dst[j] = w;
    }
#endif
}

typedef struct {
    int omit[8];
    rvo_prg prgs[8][8];
    u8 batches[8][8][16];
    u64 shares[RVO_BATCH];
    int next_idx;
} sgen2;

static void sgen2_init(sgen2 *g, const u8 keys[8][8][16], const int omit[8]) { /* share.rs:16-52 */
    for (int r = 0; r < 8; r++) {
        g->omit[r] = omit[r];
        for (int p = 0; p < 8; p++) rvo_prg_init(&g->prgs[r][p], keys[r][p]);
    }
    memset(g->batches, 0, sizeof g->batches);
    g->next_idx = RVO_BATCH;
}

static u64 sgen2_next(sgen2 *g) { /* share.rs:54-65, batch.rs:30-40, gf2/domain.rs:85-173 */
    if (g->next_idx >= RVO_BATCH) {
        for (int r = 0; r < 8; r++)
            for (int p = 0; p < 8; p++)
                if (p != g->omit[r]) rvo_prg_gen(&g->prgs[r][p], g->batches[r][p], 16);
        for (int i = 0; i < 16; i++) {
            u8 col[64];
            for (int r = 0; r < 8; r++)
                for (int p = 0; p < 8; p++) col[r * 8 + p] = g->batches[r][p][i];
            byte_to_shares(&g->shares[8 * i], col);
        }
        g->next_idx = 0;
    }
    return g->shares[g->next_idx++];
}

/* ------------------------------------------------------------------ Z64 algebra */
typedef struct {
    u64 v[8][8]; /* [rep][player], z64/share.rs:12-14 */
} sh64;
typedef struct {
    u64 v[8]; /* [rep], z64/recon.rs:14-16 */
} rc64;

static rc64 recon64(const sh64 *s) { /* z64/domain.rs:53-61 */
    rc64 r;
    for (int i = 0; i < 8; i++) {
        u64 acc = 0;
        for (int j = 0; j < 8; j++) acc += s->v[i][j];
        r.v[i] = acc;
    }
    return r;
}

typedef struct {
    int omit[8];
    rvo_prg prgs[8][8];
    u64 (*batches)[8][RVO_BATCH]; /* [8][8][128] */
    sh64 *shares;                 /* [128] */
    int next_idx;
} sgen64;

static int sgen64_init(sgen64 *g, const u8 keys[8][8][16], const int omit[8]) {
    for (int r = 0; r < 8; r++) {
        g->omit[r] = omit[r];
        for (int p = 0; p < 8; p++) rvo_prg_init(&g->prgs[r][p], keys[r][p]);
    }
    g->batches = calloc(8, sizeof *g->batches);
    g->shares = calloc(RVO_BATCH, sizeof *g->shares);
    g->next_idx = RVO_BATCH;
    return (g->batches && g->shares) ? 0 : -1;
}
static void sgen64_free(sgen64 *g) {
    free(g->batches);
    free(g->shares);
}

static void sgen64_next(sgen64 *g, sh64 *out) { /* z64/batch.rs:26-29, z64/domain.rs:64-83 */
    if (g->next_idx >= RVO_BATCH) {
        u8 raw[RVO_BATCH * 8];
        for (int r = 0; r < 8; r++)
            for (int p = 0; p < 8; p++)
                if (p != g->omit[r]) {
                    rvo_prg_gen(&g->prgs[r][p], raw, sizeof raw);
                    for (int i = 0; i < RVO_BATCH; i++) {
                        u64 x = 0;
                        for (int b = 0; b < 8; b++) x |= (u64)raw[8 * i + b] << (8 * b);
                        g->batches[r][p][i] = x;
                    }
                }
        for (int i = 0; i < RVO_BATCH; i++)
            for (int r = 0; r < 8; r++)
                for (int p = 0; p < 8; p++) g->shares[i].v[r][p] = g->batches[r][p][i];
        g->next_idx = 0;
    }
    *out = g->shares[g->next_idx++];
}

/* ------------------------------------------------------------------ transcripts */
enum { MODE_PROVER = 0, MODE_VONLINE = 1, MODE_VPRE = 2 };

typedef struct {
    u64 *p;
    size_t len, cap;
} u64vec;
static int u64vec_push(u64vec *v, u64 x) {
    if (v->len == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 256;
        u64 *np = realloc(v->p, nc * sizeof(u64));
        if (!np) return -1;
        v->p = np;
        v->cap = nc;
    }
    v->p[v->len++] = x;
    return 0;
}

typedef struct {
    int mode;
    int err;
    sgen2 gen;
    bhasher on[8], pre[8];
    /* prover (transcript/prover.rs:13-32) */
    const u8 *wit;
    size_t wit_n, wit_pos;
    u64vec recs, corrs, inputs;
    /* verifier online (verifier/online.rs:12-22): supplied values, consumed in order */
    u64vec s_recs, s_corrs, s_inputs;
    size_t p_recs, p_corrs, p_inputs;
    int okay;
    /* verifier preprocessing (verifier/preprocess.rs:10-14) */
    u8 comm_online[8][32];
} tr2;

typedef struct {
    sh64 *p;
    size_t len, cap;
} shvec;
typedef struct {
    rc64 *p;
    size_t len, cap;
} rcvec;
static int shvec_push(shvec *v, const sh64 *x) {
    if (v->len == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 64;
        sh64 *np = realloc(v->p, nc * sizeof(sh64));
        if (!np) return -1;
        v->p = np;
        v->cap = nc;
    }
    v->p[v->len++] = *x;
    return 0;
}
static int rcvec_push(rcvec *v, const rc64 *x) {
    if (v->len == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 64;
        rc64 *np = realloc(v->p, nc * sizeof(rc64));
        if (!np) return -1;
        v->p = np;
        v->cap = nc;
    }
    v->p[v->len++] = *x;
    return 0;
}

typedef struct {
    int mode;
    int err;
    sgen64 gen;
    bhasher on[8], pre[8];
    const u64 *wit;
    size_t wit_n, wit_pos;
    shvec recs;
    rcvec corrs, inputs;
    shvec s_recs;
    rcvec s_corrs, s_inputs;
    size_t p_recs, p_corrs, p_inputs;
    int okay;
    u8 comm_online[8][32];
} tr64;

/* Hashable impls: gf2/share.rs:211-218, gf2/recon.rs:314-321 (big-endian byte i = rep i) */
static inline void hash_word2(bhasher hs[8], u64 w) {
    for (int i = 0; i < 8; i++) bh_push(&hs[i], (u8)(w >> (56 - 8 * i)));
}
/* z64/share.rs:100-108, z64/recon.rs:131-137 */
static inline void hash_sh64(bhasher hs[8], const sh64 *s) {
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            u8 b[8];
            for (int k = 0; k < 8; k++) b[k] = (u8)(s->v[i][j] >> (8 * k));
            bh_update(&hs[i], b, 8);
        }
}
static inline void hash_rc64(bhasher hs[8], const rc64 *r) {
    for (int i = 0; i < 8; i++) {
        u8 b[8];
        for (int k = 0; k < 8; k++) b[k] = (u8)(r->v[i] >> (8 * k));
        bh_update(&hs[i], b, 8);
    }
}

/* ---- GF2 transcript ops (Transcript<D> trait, transcript/mod.rs:15-97) ---- */
typedef struct {
    u64 mask, corr;
} wire2;

static wire2 tr2_input(tr2 *t) {
    wire2 w;
    if (t->mode == MODE_PROVER) { /* prover.rs:181-199 */
        w.mask = sgen2_next(&t->gen);
        u64 lambda = RECON2(w.mask);
        if (t->wit_pos >= t->wit_n) {
            t->err = RVO_E_WITNESS_SHORT;
            w.corr = 0;
            return w;
        }
        u64 in = t->wit[t->wit_pos++] ? ~0ull : 0ull;
        w.corr = in ^ lambda;
        hash_word2(t->on, w.corr);
        if (u64vec_push(&t->inputs, w.corr)) t->err = RVO_E_NOMEM;
    } else if (t->mode == MODE_VONLINE) { /* online.rs:123-130: corr first, then mask */
        w.corr = t->p_inputs < t->s_inputs.len ? t->s_inputs.p[t->p_inputs++] : 0;
        hash_word2(t->on, w.corr);
        w.mask = sgen2_next(&t->gen);
    } else { /* preprocess.rs:47-53 */
        w.mask = sgen2_next(&t->gen);
        w.corr = 0;
    }
    return w;
}

static u64 tr2_reconstruct(tr2 *t, u64 mask) {
    if (t->mode == MODE_PROVER) { /* prover.rs:209-213 */
        hash_word2(t->on, mask);
        if (u64vec_push(&t->recs, mask)) t->err = RVO_E_NOMEM;
        return RECON2(mask);
    } else if (t->mode == MODE_VONLINE) { /* online.rs:140-166 */
        u64 msg = t->p_recs < t->s_recs.len ? t->s_recs.p[t->p_recs++] : 0;
        mask ^= msg;
        hash_word2(t->on, mask);
        return RECON2(mask);
    }
    return 0; /* preprocess.rs:63-65 */
}

static u64 tr2_correction(tr2 *t, u64 corr) {
    if (t->mode == MODE_PROVER) { /* prover.rs:215-219 */
        hash_word2(t->pre, corr);
        if (u64vec_push(&t->corrs, corr)) t->err = RVO_E_NOMEM;
        return corr;
    } else if (t->mode == MODE_VONLINE) { /* online.rs:168-173 */
        u64 c = t->p_corrs < t->s_corrs.len ? t->s_corrs.p[t->p_corrs++] : 0;
        hash_word2(t->pre, c);
        return c;
    }
    hash_word2(t->pre, corr); /* preprocess.rs:67-70 */
    return corr;
}

static void tr2_zero_check(tr2 *t, u64 recon) {
    if (t->mode == MODE_PROVER) { /* prover.rs:221-228: panics */
        if (recon != 0 && !t->err) t->err = RVO_E_WITNESS_INVALID;
    } else if (t->mode == MODE_VONLINE) {
        t->okay &= (recon == 0); /* online.rs:175-177 (never read, F9) */
    }
}

static wire2 op_mul2(tr2 *t, wire2 w1, wire2 w2) { /* interpreter/single.rs:25-69 */
    u64 mask_ab = sgen2_next(&t->gen);
    u64 mask_new = sgen2_next(&t->gen);
    u64 a = RECON2(w1.mask), b = RECON2(w2.mask), c = RECON2(mask_ab);
    u64 delta = tr2_correction(t, (a & b) ^ c);
    u64 s = (w2.mask & w1.corr) ^ (w1.mask & w2.corr) ^ mask_ab ^ mask_new;
    u64 recon = tr2_reconstruct(t, s) ^ delta;
    wire2 r = {mask_new, recon ^ (w1.corr & w2.corr)};
    return r;
}

/* ---- Z64 transcript ops ---- */
typedef struct {
    sh64 mask;
    rc64 corr;
} wire64;

static wire64 tr64_input(tr64 *t) {
    wire64 w;
    memset(&w, 0, sizeof w);
    if (t->mode == MODE_PROVER) {
        sgen64_next(&t->gen, &w.mask);
        rc64 lambda = recon64(&w.mask);
        if (t->wit_pos >= t->wit_n) {
            t->err = RVO_E_WITNESS_SHORT;
            return w;
        }
        u64 in = t->wit[t->wit_pos++];
        for (int i = 0; i < 8; i++) w.corr.v[i] = in - lambda.v[i];
        hash_rc64(t->on, &w.corr);
        if (rcvec_push(&t->inputs, &w.corr)) t->err = RVO_E_NOMEM;
    } else if (t->mode == MODE_VONLINE) {
        if (t->p_inputs < t->s_inputs.len) w.corr = t->s_inputs.p[t->p_inputs++];
        hash_rc64(t->on, &w.corr);
        sgen64_next(&t->gen, &w.mask);
    } else {
        sgen64_next(&t->gen, &w.mask);
    }
    return w;
}

static rc64 tr64_reconstruct(tr64 *t, sh64 *mask) {
    rc64 zero;
    memset(&zero, 0, sizeof zero);
    if (t->mode == MODE_PROVER) {
        hash_sh64(t->on, mask);
        if (shvec_push(&t->recs, mask)) t->err = RVO_E_NOMEM;
        return recon64(mask);
    } else if (t->mode == MODE_VONLINE) {
        if (t->p_recs < t->s_recs.len) {
            const sh64 *msg = &t->s_recs.p[t->p_recs++];
            for (int i = 0; i < 8; i++)
                for (int j = 0; j < 8; j++) mask->v[i][j] += msg->v[i][j];
        }
        hash_sh64(t->on, mask);
        return recon64(mask);
    }
    return zero;
}

static rc64 tr64_correction(tr64 *t, rc64 corr) {
    if (t->mode == MODE_PROVER) {
        hash_rc64(t->pre, &corr);
        if (rcvec_push(&t->corrs, &corr)) t->err = RVO_E_NOMEM;
        return corr;
    } else if (t->mode == MODE_VONLINE) {
        rc64 c;
        memset(&c, 0, sizeof c);
        if (t->p_corrs < t->s_corrs.len) c = t->s_corrs.p[t->p_corrs++];
        hash_rc64(t->pre, &c);
        return c;
    }
    hash_rc64(t->pre, &corr);
    return corr;
}

static void tr64_zero_check(tr64 *t, const rc64 *r) {
    int z = 1;
    for (int i = 0; i < 8; i++) z &= (r->v[i] == 0);
    if (t->mode == MODE_PROVER) {
        if (!z && !t->err) t->err = RVO_E_WITNESS_INVALID;
    } else if (t->mode == MODE_VONLINE) {
        t->okay &= z;
    }
}

static void op_mul64(tr64 *t, const wire64 *w1, const wire64 *w2, wire64 *out) {
    sh64 mask_ab, mask_new, s;
    sgen64_next(&t->gen, &mask_ab);
    sgen64_next(&t->gen, &mask_new);
    rc64 a = recon64(&w1->mask), b = recon64(&w2->mask), c = recon64(&mask_ab), d;
    for (int i = 0; i < 8; i++) d.v[i] = a.v[i] * b.v[i] - c.v[i];
    rc64 delta = tr64_correction(t, d);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++)
            s.v[i][j] = w2->mask.v[i][j] * w1->corr.v[i] + w1->mask.v[i][j] * w2->corr.v[i] +
                        mask_ab.v[i][j] - mask_new.v[i][j];
    rc64 rec = tr64_reconstruct(t, &s);
    out->mask = mask_new;
    for (int i = 0; i < 8; i++) out->corr.v[i] = rec.v[i] + delta.v[i] + w1->corr.v[i] * w2->corr.v[i];
}

/* ------------------------------------------------------------------ combined instance */
typedef struct {
    tr2 t2;
    tr64 t64;
    wire2 *w2;
    size_t n2;
    wire64 *w64;
    size_t n64;
    int err;
} group;

static int group_init_common(group *g, size_t z64_wires, size_t gf2_wires) {
    g->err = 0;
    g->n2 = gf2_wires;
    g->n64 = z64_wires;
    g->w2 = calloc(gf2_wires ? gf2_wires : 1, sizeof(wire2));
    g->w64 = calloc(z64_wires ? z64_wires : 1, sizeof(wire64));
    for (int i = 0; i < 8; i++) {
        bh_init(&g->t2.on[i]);
        bh_init(&g->t2.pre[i]);
        bh_init(&g->t64.on[i]);
        bh_init(&g->t64.pre[i]);
    }
    return (g->w2 && g->w64) ? 0 : -1;
}

static void group_free(group *g) {
    free(g->w2);
    free(g->w64);
    free(g->t2.recs.p);
    free(g->t2.corrs.p);
    free(g->t2.inputs.p);
    free(g->t2.s_recs.p);
    free(g->t2.s_corrs.p);
    free(g->t2.s_inputs.p);
    free(g->t64.recs.p);
    free(g->t64.corrs.p);
    free(g->t64.inputs.p);
    free(g->t64.s_recs.p);
    free(g->t64.s_corrs.p);
    free(g->t64.s_inputs.p);
    sgen64_free(&g->t64.gen);
}

/* prover group: ProverTranscript::new for both domains with the SAME seeds
 * (proof/mod.rs:131-146), share_gen_from_rep_seeds (transcript/mod.rs:108-122) */
static int group_init_prover(group *g, const u8 seeds8[8][16], const u8 *wit_gf2, size_t n_gf2,
                             const u64 *wit_z64, size_t n_z64, size_t z64_wires, size_t gf2_wires) {
    memset(g, 0, sizeof *g);
    u8 keys[8][8][16];
    int omit[8];
    for (int r = 0; r < 8; r++) {
        rvo_expand_seed(seeds8[r], keys[r]);
        omit[r] = 8;
    }
    g->t2.mode = g->t64.mode = MODE_PROVER;
    sgen2_init(&g->t2.gen, keys, omit);
    if (sgen64_init(&g->t64.gen, keys, omit)) return -1;
    g->t2.wit = wit_gf2;
    g->t2.wit_n = n_gf2;
    g->t64.wit = wit_z64;
    g->t64.wit_n = n_z64;
    return group_init_common(g, z64_wires, gf2_wires);
}

#define CHECK2(idx)                         \
    do {                                    \
        if ((size_t)(idx) >= g->n2) {       \
            g->err = RVO_E_WIRE_OOB;        \
            return;                         \
        }                                   \
    } while (0)
#define CHECK64(idx)                        \
    do {                                    \
        if ((size_t)(idx) >= g->n64) {      \
            g->err = RVO_E_WIRE_OOB;        \
            return;                         \
        }                                   \
    } while (0)

static void step_gf2(group *g, const rvo_op *op) { /* single.rs:106-157 */
    tr2 *t = &g->t2;
    u64 c = (op->imm & 1) ? ~0ull : 0ull; /* gf2/recon.rs:274-287 */
    switch (op->opcode) {
    case RVO_OP_INPUT:
        CHECK2(op->dst);
        g->w2[op->dst] = tr2_input(t);
        break;
    case RVO_OP_ADD:
    case RVO_OP_SUB: /* gf2/share.rs:220-238: both XOR */
        CHECK2(op->dst);
        CHECK2(op->a);
        CHECK2(op->b);
        {
            wire2 r = {g->w2[op->a].mask ^ g->w2[op->b].mask, g->w2[op->a].corr ^ g->w2[op->b].corr};
            g->w2[op->dst] = r;
        }
        break;
    case RVO_OP_MUL:
        CHECK2(op->dst);
        CHECK2(op->a);
        CHECK2(op->b);
        g->w2[op->dst] = op_mul2(t, g->w2[op->a], g->w2[op->b]);
        break;
    case RVO_OP_ADDCONST:
    case RVO_OP_SUBCONST:
        CHECK2(op->dst);
        CHECK2(op->a);
        {
            wire2 r = {g->w2[op->a].mask, g->w2[op->a].corr ^ c};
            g->w2[op->dst] = r;
        }
        break;
    case RVO_OP_MULCONST:
        CHECK2(op->dst);
        CHECK2(op->a);
        {
            wire2 r = {g->w2[op->a].mask & c, g->w2[op->a].corr & c};
            g->w2[op->dst] = r;
        }
        break;
    case RVO_OP_ASSERTZERO:
        CHECK2(op->a);
        {
            wire2 w = g->w2[op->a];
            u64 m = tr2_reconstruct(t, w.mask);
            tr2_zero_check(t, w.corr ^ m);
        }
        break;
    case RVO_OP_RANDOM:
        CHECK2(op->dst);
        {
            wire2 r = {sgen2_next(&t->gen), 0};
            g->w2[op->dst] = r;
        }
        break;
    case RVO_OP_CONST:
        CHECK2(op->dst);
        {
            wire2 r = {0, c};
            g->w2[op->dst] = r;
        }
        break;
    default:
        g->err = RVO_E_BAD_OP;
    }
}

static void step_z64(group *g, const rvo_op *op) {
    tr64 *t = &g->t64;
    u64 c = op->imm;
    switch (op->opcode) {
    case RVO_OP_INPUT:
        CHECK64(op->dst);
        g->w64[op->dst] = tr64_input(t);
        break;
    case RVO_OP_ADD:
    case RVO_OP_SUB:
        CHECK64(op->dst);
        CHECK64(op->a);
        CHECK64(op->b);
        {
            wire64 r;
            const wire64 *x = &g->w64[op->a], *y = &g->w64[op->b];
            int sub = op->opcode == RVO_OP_SUB;
            for (int i = 0; i < 8; i++) {
                r.corr.v[i] = sub ? x->corr.v[i] - y->corr.v[i] : x->corr.v[i] + y->corr.v[i];
                for (int j = 0; j < 8; j++)
                    r.mask.v[i][j] = sub ? x->mask.v[i][j] - y->mask.v[i][j] : x->mask.v[i][j] + y->mask.v[i][j];
            }
            g->w64[op->dst] = r;
        }
        break;
    case RVO_OP_MUL:
        CHECK64(op->dst);
        CHECK64(op->a);
        CHECK64(op->b);
        {
            wire64 r;
            op_mul64(t, &g->w64[op->a], &g->w64[op->b], &r);
            g->w64[op->dst] = r;
        }
        break;
    case RVO_OP_ADDCONST:
    case RVO_OP_SUBCONST:
        CHECK64(op->dst);
        CHECK64(op->a);
        {
            wire64 r = g->w64[op->a];
            for (int i = 0; i < 8; i++) r.corr.v[i] = op->opcode == RVO_OP_ADDCONST ? r.corr.v[i] + c : r.corr.v[i] - c;
            g->w64[op->dst] = r;
        }
        break;
    case RVO_OP_MULCONST:
        CHECK64(op->dst);
        CHECK64(op->a);
        {
            wire64 r = g->w64[op->a];
            for (int i = 0; i < 8; i++) {
                r.corr.v[i] *= c;
                for (int j = 0; j < 8; j++) r.mask.v[i][j] *= c;
            }
            g->w64[op->dst] = r;
        }
        break;
    case RVO_OP_ASSERTZERO:
        CHECK64(op->a);
        {
            wire64 w = g->w64[op->a];
            rc64 m = tr64_reconstruct(t, &w.mask);
            for (int i = 0; i < 8; i++) m.v[i] += w.corr.v[i];
            tr64_zero_check(t, &m);
        }
        break;
    case RVO_OP_RANDOM:
        CHECK64(op->dst);
        {
            wire64 r;
            memset(&r, 0, sizeof r);
            sgen64_next(&t->gen, &r.mask);
            g->w64[op->dst] = r;
        }
        break;
    case RVO_OP_CONST:
        CHECK64(op->dst);
        {
            wire64 r;
            memset(&r, 0, sizeof r);
            for (int i = 0; i < 8; i++) r.corr.v[i] = c;
            g->w64[op->dst] = r;
        }
        break;
    default:
        g->err = RVO_E_BAD_OP;
    }
}

/* recon_gf2_to_z64 (combine.rs:19-36): bit k of the result = wire k's value.
 * `recorded` selects transcript.reconstruct (hashed+recorded) vs plain reconstruct. */
static rc64 bits_to_z64(tr2 *t, const wire2 bits[64], int recorded) {
    rc64 z;
    memset(&z, 0, sizeof z);
    for (int k = 0; k < 64; k++) {
        u64 r = recorded ? tr2_reconstruct(t, bits[k].mask) : RECON2(bits[k].mask);
        u64 v = (r ^ bits[k].corr) & 0x0101010101010101ull; /* gf2/recon.rs:289-295 */
        for (int j = 0; j < 8; j++) {
            z.v[j] <<= 1;
            z.v[j] |= (v >> (56 - 8 * j)) & 1;
        }
    }
    for (int j = 0; j < 8; j++) { /* reverse_bits */
        u64 x = z.v[j], y = 0;
        for (int b = 0; b < 64; b++) y |= ((x >> b) & 1) << (63 - b);
        z.v[j] = y;
    }
    return z;
}

static wire2 xor2(wire2 a, wire2 b) {
    wire2 r = {a.mask ^ b.mask, a.corr ^ b.corr};
    return r;
}

static void step_b2a(group *g, const rvo_op *op) { /* combine.rs:132-219 */
    size_t dst = op->dst, src = op->a;
    if (dst >= g->n64) {
        g->err = RVO_E_WIRE_OOB;
        return;
    }
    if (src + 64 > g->n2 || src + 64 < src) {
        g->err = RVO_E_WIRE_OOB;
        return;
    }
    tr2 *t2 = &g->t2;
    tr64 *t64 = &g->t64;
    wire2 a[64], res[64];
    for (int k = 0; k < 64; k++) {
        a[k].mask = sgen2_next(&t2->gen);
        a[k].corr = 0;
    }
    rc64 zval = bits_to_z64(t2, a, 0);
    wire64 zw;
    sgen64_next(&t64->gen, &zw.mask);
    {
        rc64 mr = recon64(&zw.mask), d;
        for (int i = 0; i < 8; i++) d.v[i] = zval.v[i] - mr.v[i];
        zw.corr = tr64_correction(t64, d);
    }
    /* add_64 (combine.rs:39-93) */
    const wire2 *b = &g->w2[src];
    wire2 carry = op_mul2(t2, a[0], b[0]);
    res[0] = xor2(a[0], b[0]);
    for (int i = 1; i < 63; i++) {
        wire2 ac = xor2(a[i], carry);
        wire2 bc = xor2(b[i], carry);
        wire2 acbc = op_mul2(t2, ac, bc);
        res[i] = xor2(ac, b[i]);
        carry = xor2(acbc, carry);
    }
    res[63] = xor2(carry, xor2(a[63], b[63]));
    rc64 zrec = bits_to_z64(t2, res, 1);
    wire64 out;
    for (int i = 0; i < 8; i++) {
        out.corr.v[i] = zrec.v[i] - zw.corr.v[i];
        for (int j = 0; j < 8; j++) out.mask.v[i][j] = 0 - zw.mask.v[i][j];
    }
    g->w64[dst] = out;
}

static void group_step(group *g, const rvo_op *op) { /* combine.rs:120-131 */
    switch (op->domain) {
    case RVO_DOM_GF2:
        step_gf2(g, op);
        break;
    case RVO_DOM_Z64:
        step_z64(g, op);
        break;
    case RVO_DOM_B2A:
        step_b2a(g, op);
        break;
    case RVO_DOM_SIZEHINT: { /* a = z64 count, b = gf2 count */
        if (g->n64 < op->a) {
            wire64 *n = realloc(g->w64, (size_t)op->a * sizeof(wire64));
            if (!n) {
                g->err = RVO_E_NOMEM;
                return;
            }
            memset(n + g->n64, 0, ((size_t)op->a - g->n64) * sizeof(wire64));
            g->w64 = n;
            g->n64 = op->a;
        }
        if (g->n2 < op->b) {
            wire2 *n = realloc(g->w2, (size_t)op->b * sizeof(wire2));
            if (!n) {
                g->err = RVO_E_NOMEM;
                return;
            }
            memset(n + g->n2, 0, ((size_t)op->b - g->n2) * sizeof(wire2));
            g->w2 = n;
            g->n2 = op->b;
        }
        break;
    }
    default:
        g->err = RVO_E_BAD_OP;
    }
}

static int group_run(group *g, const rvo_op *ops, size_t n_ops) {
    for (size_t i = 0; i < n_ops; i++) {
        group_step(g, &ops[i]);
        if (g->err) return g->err;
        if (g->t2.err) return g->t2.err;
        if (g->t64.err) return g->t64.err;
    }
    return 0;
}

/* Transcript::hash (transcript/mod.rs:77-96) + CombineInstance::hash (combine.rs:104-118) */
static void group_hashes(const group *g, u8 h[8][32], u8 streams[8][4][32]) {
    for (int i = 0; i < 8; i++) {
        u8 st[4][32], join[64], h2[32], h64[32];
        bh_finalize(&g->t2.pre[i], st[0]);
        if (g->t2.mode == MODE_VPRE)
            memcpy(st[1], g->t2.comm_online[i], 32);
        else
            bh_finalize(&g->t2.on[i], st[1]);
        bh_finalize(&g->t64.pre[i], st[2]);
        if (g->t64.mode == MODE_VPRE)
            memcpy(st[3], g->t64.comm_online[i], 32);
        else
            bh_finalize(&g->t64.on[i], st[3]);
        memcpy(join, st[0], 32);
        memcpy(join + 32, st[1], 32);
        rvo_blake3_hash(join, 64, h2);
        memcpy(join, st[2], 32);
        memcpy(join + 32, st[3], 32);
        rvo_blake3_hash(join, 64, h64);
        memcpy(join, h2, 32);
        memcpy(join + 32, h64, 32);
        rvo_blake3_hash(join, 64, h[i]);
        if (streams) memcpy(streams[i], st, sizeof st);
    }
}

/* ------------------------------------------------------------------ challenge */
void rvo_challenge(const uint8_t comm[32], uint8_t omit[256]) { /* proof/mod.rs:68-83, ro.rs:8-20 */
    static const char CTX[] = "random-oracle challenge";
    rvo_blake3 h;
    rvo_blake3_init(&h);
    rvo_blake3_update(&h, CTX, sizeof CTX - 1);
    u8 z = 0;
    rvo_blake3_update(&h, &z, 1);
    rvo_blake3_update(&h, comm, 32);
    memset(omit, 8, 256);
    int count = 0;
    u64 pos = 0;
    while (count < RVO_ONLINE_REPS) {
        u8 buf[32];
        rvo_blake3_finalize_xof(&h, pos, buf, 32);
        pos += 32;
        /* u128 LE mod 256 = low byte; mod 8 = low 3 bits of the low byte */
        unsigned rep = buf[0];
        unsigned om = buf[16] & 7;
        if (omit[rep] == 8) count++;
        omit[rep] = (u8)om;
    }
}

/* ------------------------------------------------------------------ Pack / PackSelected */
/* gf2/recon.rs:127-148 */
static u8 recon_pack_byte(unsigned shift, const u64 src[8]) {
    u8 res = (u8)((src[0] >> shift) & 2);
    for (int k = 1; k < 8; k++) {
        res |= (u8)((src[k] >> shift) & 1);
        if (k < 7) res <<= 1;
    }
    return res;
}

/* Pack for ReconGF2 (gf2/recon.rs:189-239) into bvec per rep */
static int gf2_recon_pack(bvec dst[8], const u64 *src, size_t n, const u8 selected[8]) {
    int any = 0;
    for (int i = 0; i < 8; i++) any |= selected[i];
    if (!any) return 0;
    size_t full = n / 8;
    for (size_t c = 0; c <= full; c++) {
        u64 arr[8] = {0};
        size_t cnt = c < full ? 8 : n - 8 * full;
        for (size_t k = 0; k < cnt; k++) arr[k] = src[8 * c + k];
        for (int i = 0; i < 8; i++)
            if (selected[i])
                if (bvec_push_u8(&dst[i], recon_pack_byte(64 - (unsigned)(i + 1) * 8, arr))) return -1;
    }
    return 0;
}

/* gf2/share.rs:66-85 */
static u8 share_pack_byte(unsigned shift, const u64 src[8]) {
    u64 res = 0;
    for (int k = 0; k < 8; k++) {
        res |= (src[k] >> shift) & 1;
        if (k < 7) res <<= 1;
    }
    return (u8)res;
}

/* PackSelected for ShareGF2 (gf2/share.rs:87-149) */
static int gf2_share_pack_selected(bvec dst[8], const u64 *src, size_t n, const int selected[8]) {
    int any = 0;
    for (int i = 0; i < 8; i++) any |= selected[i] < 8;
    if (!any) return 0;
    size_t full = n / 8;
    for (size_t c = 0; c <= full; c++) {
        u64 arr[8] = {0};
        size_t cnt = c < full ? 8 : n - 8 * full;
        for (size_t k = 0; k < cnt; k++) arr[k] = src[8 * c + k];
        for (int i = 0; i < 8; i++)
            if (selected[i] < 8) {
                unsigned shift = (unsigned)((7 - i) * 8 + (7 - selected[i]));
                if (bvec_push_u8(&dst[i], share_pack_byte(shift, arr))) return -1;
            }
    }
    return 0;
}

/* ReconGF2::unpack (gf2/recon.rs:151-167,241-259): len bytes per rep -> 8*len recons */
static int gf2_recon_unpack(u64vec *dst, const u8 *const src[8], size_t len) {
    for (size_t i = 0; i < len; i++)
        for (int bit = 0; bit < 8; bit++) {
            u64 w = 0;
            for (int r = 0; r < 8; r++)
                if ((src[r][i] >> (7 - bit)) & 1) w |= 0xFFull << (56 - 8 * r);
            if (u64vec_push(dst, w)) return -1;
        }
    return 0;
}

/* ShareGF2::unpack_selected (gf2/share.rs:151-208) */
static int gf2_share_unpack_selected(u64vec *dst, const u8 *const src[8], size_t len, const int selected[8]) {
    for (size_t i = 0; i < len; i++) {
        u8 tmp[64] = {0};
        u64 out[8];
        for (int j = 0; j < 8; j++) tmp[selected[j] + 8 * j] = src[j][i];
        byte_to_shares(out, tmp);
        for (int k = 0; k < 8; k++)
            if (u64vec_push(dst, out[k])) return -1;
    }
    return 0;
}

void rvo_gf2_recon_pack(const uint64_t *src, size_t n, const uint8_t selected[8], uint8_t *dst, size_t cap,
                        size_t lens[8]) {
    bvec v[8];
    memset(v, 0, sizeof v);
    gf2_recon_pack(v, src, n, selected);
    for (int i = 0; i < 8; i++) {
        lens[i] = v[i].len;
        memcpy(dst + (size_t)i * cap, v[i].p, v[i].len < cap ? v[i].len : cap);
        bvec_free(&v[i]);
    }
}
size_t rvo_gf2_recon_unpack(const uint8_t *src, size_t len, uint64_t *dst) {
    const u8 *p[8];
    for (int i = 0; i < 8; i++) p[i] = src + (size_t)i * len;
    u64vec v = {0};
    gf2_recon_unpack(&v, p, len);
    memcpy(dst, v.p, v.len * 8);
    size_t n = v.len;
    free(v.p);
    return n;
}
void rvo_gf2_share_pack_selected(const uint64_t *src, size_t n, const uint32_t selected[8], uint8_t *dst,
                                 size_t cap, size_t lens[8]) {
    bvec v[8];
    int sel[8];
    memset(v, 0, sizeof v);
    for (int i = 0; i < 8; i++) sel[i] = (int)selected[i];
    gf2_share_pack_selected(v, src, n, sel);
    for (int i = 0; i < 8; i++) {
        lens[i] = v[i].len;
        memcpy(dst + (size_t)i * cap, v[i].p, v[i].len < cap ? v[i].len : cap);
        bvec_free(&v[i]);
    }
}
size_t rvo_gf2_share_unpack_selected(const uint8_t *src, size_t len, const uint32_t selected[8], uint64_t *dst) {
    const u8 *p[8];
    int sel[8];
    for (int i = 0; i < 8; i++) {
        p[i] = src + (size_t)i * len;
        sel[i] = (int)selected[i];
    }
    u64vec v = {0};
    gf2_share_unpack_selected(&v, p, len, sel);
    memcpy(dst, v.p, v.len * 8);
    size_t n = v.len;
    free(v.p);
    return n;
}

/* ------------------------------------------------------------------ test hooks */
void rvo_sharegen_gf2(const uint8_t *keys, const uint32_t omit[8], size_t n, uint64_t *out) {
    sgen2 *g = malloc(sizeof *g);
    int om[8];
    for (int i = 0; i < 8; i++) om[i] = (int)omit[i];
    sgen2_init(g, (const u8(*)[8][16])keys, om);
    for (size_t i = 0; i < n; i++) out[i] = sgen2_next(g);
    free(g);
}
void rvo_sharegen_z64(const uint8_t *keys, const uint32_t omit[8], size_t n, uint64_t *out) {
    sgen64 g;
    int om[8];
    for (int i = 0; i < 8; i++) om[i] = (int)omit[i];
    sgen64_init(&g, (const u8(*)[8][16])keys, om);
    for (size_t i = 0; i < n; i++) sgen64_next(&g, (sh64 *)(out + 64 * i));
    sgen64_free(&g);
}

int rvo_group_wire_values(const rvo_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2,
                          const uint64_t *wit_z64, size_t n_z64, size_t z64_wires, size_t gf2_wires,
                          const uint8_t *seeds8, uint32_t gf2_wire, uint64_t *gf2_out, uint32_t z64_wire,
                          uint64_t *z64_out) {
    group *g = malloc(sizeof *g);
    if (!g) return RVO_E_NOMEM;
    if (group_init_prover(g, (const u8(*)[16])seeds8, wit_gf2, n_gf2, wit_z64, n_z64, z64_wires, gf2_wires)) {
        group_free(g);
        free(g);
        return RVO_E_NOMEM;
    }
    int rc = group_run(g, ops, n_ops);
    if (!rc) {
        if (gf2_out) {
            if (gf2_wire >= g->n2)
                rc = RVO_E_WIRE_OOB;
            else
                *gf2_out = RECON2(g->w2[gf2_wire].mask) ^ g->w2[gf2_wire].corr; /* interpreter/mod.rs:15-19 */
        }
        if (z64_out && !rc) {
            if (z64_wire >= g->n64)
                rc = RVO_E_WIRE_OOB;
            else {
                rc64 r = recon64(&g->w64[z64_wire].mask);
                for (int i = 0; i < 8; i++) z64_out[i] = r.v[i] + g->w64[z64_wire].corr.v[i];
            }
        }
    }
    group_free(g);
    free(g);
    return rc;
}

/* ------------------------------------------------------------------ thread pool over groups */
typedef struct job_s {
    void (*fn)(struct job_s *, int idx);
    int n;
    int next;
    pthread_mutex_t mu;
} job;

static void *worker(void *arg) {
    job *j = (job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int i = j->next < j->n ? j->next++ : -1;
        pthread_mutex_unlock(&j->mu);
        if (i < 0) break;
        j->fn(j, i);
    }
    return NULL;
}

static void run_parallel(job *j, int threads) {
    if (threads < 1) threads = 1;
    if (threads > j->n) threads = j->n;
    j->next = 0;
    pthread_mutex_init(&j->mu, NULL);
    if (threads == 1) {
        worker(j);
    } else {
        pthread_t th[64];
        if (threads > 64) threads = 64;
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, j);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    pthread_mutex_destroy(&j->mu);
}

/* ------------------------------------------------------------------ prover */
typedef struct {
    job j;
    const rvo_op *ops;
    size_t n_ops;
    const u8 *wit_gf2;
    size_t n_gf2;
    const u64 *wit_z64;
    size_t n_z64;
    size_t z64_wires, gf2_wires;
    const u8 *seeds;
    group *groups[RVO_GROUPS];
    int rc[RVO_GROUPS];
    u8 h[RVO_GROUPS][8][32];
    u8 streams[RVO_GROUPS][8][4][32];
} prove_job;

static void prove_group(job *jb, int q) { /* proof/mod.rs:127-157 */
    prove_job *pj = (prove_job *)jb;
    group *g = malloc(sizeof *g);
    pj->groups[q] = g;
    if (!g) {
        pj->rc[q] = RVO_E_NOMEM;
        return;
    }
    if (group_init_prover(g, (const u8(*)[16])(pj->seeds + (size_t)q * 8 * 16), pj->wit_gf2, pj->n_gf2, pj->wit_z64,
                          pj->n_z64, pj->z64_wires, pj->gf2_wires)) {
        pj->rc[q] = RVO_E_NOMEM;
        return;
    }
    pj->rc[q] = group_run(g, pj->ops, pj->n_ops);
    if (!pj->rc[q]) group_hashes(g, pj->h[q], pj->streams[q]);
    /* wires are no longer needed (the reference drops the Instance, keeps the transcript) */
    free(g->w2);
    g->w2 = NULL;
    free(g->w64);
    g->w64 = NULL;
}

static prove_job *prove_run(const rvo_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2,
                            const uint64_t *wit_z64, size_t n_z64, size_t z64_wires, size_t gf2_wires,
                            const uint8_t *seeds, int threads, int *rc) {
    prove_job *pj = calloc(1, sizeof *pj);
    if (!pj) {
        *rc = RVO_E_NOMEM;
        return NULL;
    }
    pj->j.fn = prove_group;
    pj->j.n = RVO_GROUPS;
    pj->ops = ops;
    pj->n_ops = n_ops;
    pj->wit_gf2 = wit_gf2;
    pj->n_gf2 = n_gf2;
    pj->wit_z64 = wit_z64;
    pj->n_z64 = n_z64;
    pj->z64_wires = z64_wires;
    pj->gf2_wires = gf2_wires;
    pj->seeds = seeds;
    run_parallel(&pj->j, threads);
    *rc = 0;
    for (int q = 0; q < RVO_GROUPS; q++)
        if (pj->rc[q] && !*rc) *rc = pj->rc[q];
    return pj;
}

static void prove_job_free(prove_job *pj) {
    if (!pj) return;
    for (int q = 0; q < RVO_GROUPS; q++)
        if (pj->groups[q]) {
            group_free(pj->groups[q]);
            free(pj->groups[q]);
        }
    free(pj);
}

int rvo_commit(const rvo_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64,
               size_t n_z64, size_t z64_wires, size_t gf2_wires, const uint8_t *seeds, int threads, uint8_t *h,
               uint8_t *streams, uint8_t *comm) {
    int rc;
    prove_job *pj = prove_run(ops, n_ops, wit_gf2, n_gf2, wit_z64, n_z64, z64_wires, gf2_wires, seeds, threads, &rc);
    if (!rc) {
        if (h) memcpy(h, pj->h, sizeof pj->h);
        if (streams) memcpy(streams, pj->streams, sizeof pj->streams);
        if (comm) rvo_blake3_hash(pj->h, sizeof pj->h, comm);
    }
    prove_job_free(pj);
    return rc;
}

/* ProverTranscript::extract (prover.rs:57-175) for one group and one domain, appended
 * in bincode form (Appendix A.6) to `online` / `pre`. */
static int emit_open_online(bvec *out, int omit, const u8 seed[16], const bvec *recs, const bvec *corrs,
                            const bvec *inputs) {
    u8 keys[8][16];
    rvo_expand_seed(seed, keys);
    memset(keys[omit], 0, 16); /* prover.rs:125-127 */
    if (bvec_push_u8(out, (u8)omit)) return -1;
    if (bvec_push(out, keys, 128)) return -1;
    const bvec *vs[3] = {recs, corrs, inputs}; /* field order proof/mod.rs:41-47 */
    for (int k = 0; k < 3; k++) {
        if (bvec_push_u64le(out, vs[k]->len)) return -1;
        if (vs[k]->len && bvec_push(out, vs[k]->p, vs[k]->len)) return -1;
    }
    return 0;
}

static int extract_group(const group *g, const u8 seeds8[8][16], const u8 omit8[8], const u8 streams[8][4][32],
                         bvec *on2, bvec *pre2, bvec *on64, bvec *pre64) {
    int sel_player[8];
    u8 sel[8];
    for (int i = 0; i < 8; i++) {
        sel_player[i] = omit8[i];
        sel[i] = omit8[i] < 8;
    }
    bvec r2[8], c2[8], i2[8], r64[8], c64[8], i64[8];
    memset(r2, 0, sizeof r2);
    memset(c2, 0, sizeof c2);
    memset(i2, 0, sizeof i2);
    memset(r64, 0, sizeof r64);
    memset(c64, 0, sizeof c64);
    memset(i64, 0, sizeof i64);
    int rc = 0;
    rc |= gf2_share_pack_selected(r2, g->t2.recs.p, g->t2.recs.len, sel_player);
    rc |= gf2_recon_pack(c2, g->t2.corrs.p, g->t2.corrs.len, sel);
    rc |= gf2_recon_pack(i2, g->t2.inputs.p, g->t2.inputs.len, sel);
    /* z64/share.rs:36-49, z64/recon.rs:45-66 */
    for (size_t e = 0; e < g->t64.recs.len && !rc; e++)
        for (int i = 0; i < 8; i++)
            if (sel[i]) rc |= bvec_push_u64le(&r64[i], g->t64.recs.p[e].v[i][sel_player[i]]);
    for (size_t e = 0; e < g->t64.corrs.len && !rc; e++)
        for (int i = 0; i < 8; i++)
            if (sel[i]) rc |= bvec_push_u64le(&c64[i], g->t64.corrs.p[e].v[i]);
    for (size_t e = 0; e < g->t64.inputs.len && !rc; e++)
        for (int i = 0; i < 8; i++)
            if (sel[i]) rc |= bvec_push_u64le(&i64[i], g->t64.inputs.p[e].v[i]);
    for (int i = 0; i < 8 && !rc; i++) {
        if (sel[i]) {
            rc |= emit_open_online(on2, omit8[i], seeds8[i], &r2[i], &c2[i], &i2[i]);
            rc |= emit_open_online(on64, omit8[i], seeds8[i], &r64[i], &c64[i], &i64[i]);
        } else { /* prover.rs:167-170; field order proof/mod.rs:50-53 */
            rc |= bvec_push(pre2, seeds8[i], 16);
            rc |= bvec_push(pre2, streams[i][1], 32);
            rc |= bvec_push(pre64, seeds8[i], 16);
            rc |= bvec_push(pre64, streams[i][3], 32);
        }
    }
    for (int i = 0; i < 8; i++) {
        bvec_free(&r2[i]);
        bvec_free(&c2[i]);
        bvec_free(&i2[i]);
        bvec_free(&r64[i]);
        bvec_free(&c64[i]);
        bvec_free(&i64[i]);
    }
    return rc ? RVO_E_NOMEM : 0;
}

int rvo_prove(const rvo_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64,
              size_t n_z64, size_t z64_wires, size_t gf2_wires, const uint8_t *seeds, int threads, uint8_t **proof,
              size_t *proof_len) {
    int rc;
    *proof = NULL;
    *proof_len = 0;
    prove_job *pj = prove_run(ops, n_ops, wit_gf2, n_gf2, wit_z64, n_z64, z64_wires, gf2_wires, seeds, threads, &rc);
    if (rc) {
        prove_job_free(pj);
        return rc;
    }
    u8 comm[32], omit[256];
    rvo_blake3_hash(pj->h, sizeof pj->h, comm); /* combine_hashes, proof/mod.rs:160-168 */
    rvo_challenge(comm, omit);
    bvec on2 = {0}, pre2 = {0}, on64 = {0}, pre64 = {0}, out = {0};
    for (int q = 0; q < RVO_GROUPS && !rc; q++)
        rc = extract_group(pj->groups[q], (const u8(*)[16])(seeds + (size_t)q * 128), omit + 8 * q, pj->streams[q], &on2,
                           &pre2, &on64, &pre64);
    if (!rc) { /* bincode: Proof { comm, gf2, z64 }, ProofSingle { online, preprocessing } */
        int e = 0;
        e |= bvec_push(&out, comm, 32);
        e |= bvec_push_u64le(&out, RVO_ONLINE_REPS);
        e |= bvec_push(&out, on2.p, on2.len);
        e |= bvec_push_u64le(&out, RVO_PRE_REPS);
        e |= bvec_push(&out, pre2.p, pre2.len);
        e |= bvec_push_u64le(&out, RVO_ONLINE_REPS);
        e |= bvec_push(&out, on64.p, on64.len);
        e |= bvec_push_u64le(&out, RVO_PRE_REPS);
        e |= bvec_push(&out, pre64.p, pre64.len);
        if (e) rc = RVO_E_NOMEM;
    }
    bvec_free(&on2);
    bvec_free(&pre2);
    bvec_free(&on64);
    bvec_free(&pre64);
    prove_job_free(pj);
    if (rc) {
        bvec_free(&out);
        return rc;
    }
    *proof = out.p;
    *proof_len = out.len;
    return 0;
}

void rvo_free(void *p) { free(p); }

/* ------------------------------------------------------------------ proof parsing */
typedef struct {
    u8 omit;
    const u8 *seeds; /* 8 x 16 */
    const u8 *recons, *corrs, *inputs;
    size_t n_recons, n_corrs, n_inputs;
} open_online;
typedef struct {
    const u8 *seed;
    const u8 *comm_online;
} open_pre;
typedef struct {
    open_online *online;
    size_t n_online;
    open_pre *pre;
    size_t n_pre;
} proof_single;

typedef struct {
    const u8 *p;
    size_t len, pos;
    int bad;
} reader;

static const u8 *rd_bytes(reader *r, size_t n) {
    if (r->bad || n > r->len - r->pos) {
        r->bad = 1;
        return NULL;
    }
    const u8 *q = r->p + r->pos;
    r->pos += n;
    return q;
}
static u64 rd_u64(reader *r) {
    const u8 *q = rd_bytes(r, 8);
    u64 x = 0;
    if (q)
        for (int i = 0; i < 8; i++) x |= (u64)q[i] << (8 * i);
    return x;
}

static int parse_single(reader *r, proof_single *ps) {
    memset(ps, 0, sizeof *ps);
    u64 n = rd_u64(r);
    if (r->bad || n > (r->len - r->pos) / 153 + 1) return -1;
    ps->n_online = (size_t)n;
    ps->online = calloc(n ? n : 1, sizeof(open_online));
    if (!ps->online) return -1;
    for (size_t i = 0; i < ps->n_online; i++) {
        open_online *o = &ps->online[i];
        const u8 *b = rd_bytes(r, 1);
        if (!b) return -1;
        o->omit = *b;
        o->seeds = rd_bytes(r, 128);
        o->n_recons = (size_t)rd_u64(r);
        o->recons = rd_bytes(r, o->n_recons);
        o->n_corrs = (size_t)rd_u64(r);
        o->corrs = rd_bytes(r, o->n_corrs);
        o->n_inputs = (size_t)rd_u64(r);
        o->inputs = rd_bytes(r, o->n_inputs);
        if (r->bad) return -1;
    }
    n = rd_u64(r);
    if (r->bad || n > (r->len - r->pos) / 48 + 1) return -1;
    ps->n_pre = (size_t)n;
    ps->pre = calloc(n ? n : 1, sizeof(open_pre));
    if (!ps->pre) return -1;
    for (size_t i = 0; i < ps->n_pre; i++) {
        ps->pre[i].seed = rd_bytes(r, 16);
        ps->pre[i].comm_online = rd_bytes(r, 32);
        if (r->bad) return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ verifier */
typedef struct {
    job j;
    const rvo_op *ops;
    size_t n_ops;
    size_t z64_wires, gf2_wires;
    proof_single gf2, z64;
    int rc[RVO_GROUPS];
    int okay[RVO_GROUPS]; /* VerifierTranscriptOnline.okay of the group's two domains (online.rs:21,175-177) */
    u8 h[RVO_GROUPS][8][32]; /* first 5 groups online, then 27 preprocessing */
} verify_job;

static u64 le64(const u8 *p) {
    u64 x = 0;
    for (int i = 0; i < 8; i++) x |= (u64)p[i] << (8 * i);
    return x;
}

static void verify_group(job *jb, int q) {
    verify_job *vj = (verify_job *)jb;
    group *g = calloc(1, sizeof *g);
    if (!g) {
        vj->rc[q] = RVO_E_NOMEM;
        return;
    }
    int rc = 0;
    if (q < RVO_ONLINE_REPS / 8) { /* VerifierTranscriptOnline::new, online.rs:25-119 */
        const open_online *o2 = &vj->gf2.online[8 * q], *o64 = &vj->z64.online[8 * q];
        u8 keys[8][8][16];
        int omit[8];
        /* gf2 */
        for (int i = 0; i < 8; i++) {
            omit[i] = o2[i].omit;
            if (omit[i] >= 8) rc = RVO_E_PROOF_MALFORMED; /* reference: UB (gf2/share.rs:167-199), tightened */
            memcpy(keys[i], o2[i].seeds, 128);
        }
        if (!rc) {
            g->t2.mode = MODE_VONLINE;
            g->t2.okay = 1;
            sgen2_init(&g->t2.gen, keys, omit);
            const u8 *src[8];
            size_t len;
            /* Recon::unpack corrs, inputs: length of rep 0, others must be at least as long (index panic) */
            len = o2[0].n_corrs;
            for (int i = 0; i < 8; i++) {
                src[i] = o2[i].corrs;
                if (o2[i].n_corrs < len) rc = RVO_E_PROOF_MALFORMED;
            }
            if (!rc && gf2_recon_unpack(&g->t2.s_corrs, src, len)) rc = RVO_E_NOMEM;
            len = o2[0].n_inputs;
            for (int i = 0; i < 8; i++) {
                src[i] = o2[i].inputs;
                if (o2[i].n_inputs < len) rc = RVO_E_PROOF_MALFORMED;
            }
            if (!rc && gf2_recon_unpack(&g->t2.s_inputs, src, len)) rc = RVO_E_NOMEM;
            /* Share::unpack_selected: assert_eq on all lengths (gf2/share.rs:157-164) */
            len = o2[0].n_recons;
            for (int i = 0; i < 8; i++) {
                src[i] = o2[i].recons;
                if (o2[i].n_recons != len) rc = RVO_E_PROOF_MALFORMED;
            }
            if (!rc && gf2_share_unpack_selected(&g->t2.s_recs, src, len, omit)) rc = RVO_E_NOMEM;
        }
        /* z64: z64/recon.rs:68-108, z64/share.rs:51-91 (missing chunks read as zero) */
        if (!rc) {
            for (int i = 0; i < 8; i++) {
                omit[i] = o64[i].omit;
                if (omit[i] >= 8) rc = RVO_E_PROOF_MALFORMED;
                memcpy(keys[i], o64[i].seeds, 128);
            }
        }
        if (!rc) {
            g->t64.mode = MODE_VONLINE;
            g->t64.okay = 1;
            if (sgen64_init(&g->t64.gen, keys, omit)) rc = RVO_E_NOMEM;
            size_t n = o64[0].n_corrs / 8;
            for (size_t e = 0; e < n && !rc; e++) {
                rc64 v;
                for (int i = 0; i < 8; i++) v.v[i] = (e + 1) * 8 <= o64[i].n_corrs ? le64(o64[i].corrs + 8 * e) : 0;
                if (rcvec_push(&g->t64.s_corrs, &v)) rc = RVO_E_NOMEM;
            }
            n = o64[0].n_inputs / 8;
            for (size_t e = 0; e < n && !rc; e++) {
                rc64 v;
                for (int i = 0; i < 8; i++) v.v[i] = (e + 1) * 8 <= o64[i].n_inputs ? le64(o64[i].inputs + 8 * e) : 0;
                if (rcvec_push(&g->t64.s_inputs, &v)) rc = RVO_E_NOMEM;
            }
            n = o64[0].n_recons / 8;
            for (size_t e = 0; e < n && !rc; e++) {
                sh64 v;
                memset(&v, 0, sizeof v);
                for (int i = 0; i < 8; i++)
                    v.v[i][omit[i]] = (e + 1) * 8 <= o64[i].n_recons ? le64(o64[i].recons + 8 * e) : 0;
                if (shvec_push(&g->t64.s_recs, &v)) rc = RVO_E_NOMEM;
            }
        }
    } else { /* VerifierTranscriptPreprocess::new, preprocess.rs:17-43 */
        int k = q - RVO_ONLINE_REPS / 8;
        const open_pre *p2 = &vj->gf2.pre[8 * k], *p64 = &vj->z64.pre[8 * k];
        u8 keys[8][8][16];
        int omit[8];
        for (int i = 0; i < 8; i++) {
            rvo_expand_seed(p2[i].seed, keys[i]);
            omit[i] = 8;
            memcpy(g->t2.comm_online[i], p2[i].comm_online, 32);
        }
        g->t2.mode = MODE_VPRE;
        sgen2_init(&g->t2.gen, keys, omit);
        for (int i = 0; i < 8; i++) {
            rvo_expand_seed(p64[i].seed, keys[i]);
            memcpy(g->t64.comm_online[i], p64[i].comm_online, 32);
        }
        g->t64.mode = MODE_VPRE;
        if (sgen64_init(&g->t64.gen, keys, omit)) rc = RVO_E_NOMEM;
    }
    if (!rc && group_init_common(g, vj->z64_wires, vj->gf2_wires)) rc = RVO_E_NOMEM;
    if (!rc) rc = group_run(g, vj->ops, vj->n_ops);
    if (!rc) group_hashes(g, vj->h[q], NULL);
    vj->okay[q] = q >= RVO_ONLINE_REPS / 8 || (g->t2.okay && g->t64.okay);
    vj->rc[q] = rc;
    group_free(g);
    free(g);
}

int rvo_verify(const rvo_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, const uint8_t *proof,
               size_t proof_len, int threads, int *ok) {
    return rvo_verify_ex(ops, n_ops, z64_wires, gf2_wires, proof, proof_len, threads, 0, ok);
}

int rvo_verify_ex(const rvo_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, const uint8_t *proof,
                  size_t proof_len, int threads, int strict, int *ok) {
    *ok = 0;
    reader r = {proof, proof_len, 0, 0};
    const u8 *comm = rd_bytes(&r, 32);
    verify_job *vj = calloc(1, sizeof *vj);
    if (!vj) return RVO_E_NOMEM;
    int rc = 0;
    /* trailing bytes are ignored, as bincode::deserialize_from does (main.rs:101-103) */
    if (!comm || parse_single(&r, &vj->gf2) || parse_single(&r, &vj->z64)) rc = RVO_E_PROOF_MALFORMED;
    if (!rc) {
        /* check_format (proof/mod.rs:110-114,225-230): wrong counts -> false, not an error */
        if (vj->gf2.n_online != RVO_ONLINE_REPS || vj->gf2.n_pre != RVO_PRE_REPS || vj->z64.n_online != RVO_ONLINE_REPS ||
            vj->z64.n_pre != RVO_PRE_REPS)
            goto done;
        vj->j.fn = verify_group;
        vj->j.n = RVO_GROUPS;
        vj->ops = ops;
        vj->n_ops = n_ops;
        vj->z64_wires = z64_wires;
        vj->gf2_wires = gf2_wires;
        run_parallel(&vj->j, threads);
        for (int q = 0; q < RVO_GROUPS; q++)
            if (vj->rc[q] && !rc) rc = vj->rc[q];
        if (!rc) { /* proof/mod.rs:283-306 */
            u8 omit[256];
            rvo_challenge(comm, omit);
            const u8(*flat)[32] = (const u8(*)[32])vj->h;
            size_t on = 0, pre = RVO_ONLINE_REPS;
            rvo_blake3 hs;
            rvo_blake3_init(&hs);
            for (int i = 0; i < RVO_TOTAL_REPS; i++) rvo_blake3_update(&hs, omit[i] < 8 ? flat[on++] : flat[pre++], 32);
            u8 c2[32];
            rvo_blake3_finalize(&hs, c2);
            *ok = memcmp(c2, comm, 32) == 0;
            if (strict) {
                /* the reference never reads `okay` and never compares the records' `omit` with the
                 * challenge; the strict form enforces both */
                for (int q = 0; q < RVO_ONLINE_REPS / 8; q++)
                    if (!vj->okay[q]) *ok = 0;
                size_t k = 0;
                for (int i = 0; i < RVO_TOTAL_REPS; i++)
                    if (omit[i] < 8) {
                        if (vj->gf2.online[k].omit != omit[i] || vj->z64.online[k].omit != omit[i]) *ok = 0;
                        k++;
                    }
            }
        }
    }
done:
    free(vj->gf2.online);
    free(vj->gf2.pre);
    free(vj->z64.online);
    free(vj->z64.pre);
    free(vj);
    return rc;
}
