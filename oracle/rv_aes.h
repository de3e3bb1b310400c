/* TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/README.md).
 * AES-128 block encryption + the CTR keystream Reverie's PRG is built on.
 *
 * Follows /root/reference/src/crypto/prg.rs:7-37: `Ctr128BE<Aes128>`, key = seed,
 * IV = 0, 128-bit big-endian block counter starting at 0, `gen` = raw keystream.
 * The `aes`/`ctr` crates are third-party (Cargo.toml:28,34 "0.8"/"0.9", not under
 * /root/reference); the algorithm restated here is FIPS-197 AES-128 + SP 800-38A CTR.
 */
#ifndef RV_ORACLE_AES_H
#define RV_ORACLE_AES_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint8_t rk[11][16]; /* expanded round keys, byte order as FIPS-197 */
} rvo_aes128;

void rvo_aes128_init(rvo_aes128 *ctx, const uint8_t key[16]);
void rvo_aes128_encrypt(const rvo_aes128 *ctx, const uint8_t in[16], uint8_t out[16]);

/* PRG (prg.rs:12-37): AES-128-CTR keystream generator. */
typedef struct {
    rvo_aes128 aes;
    uint64_t ctr_hi, ctr_lo; /* 128-bit block counter (big-endian on the wire) */
} rvo_prg;

void rvo_prg_init(rvo_prg *prg, const uint8_t key[16]);
/* writes `len` keystream bytes (len % 16 == 0), advancing the counter */
void rvo_prg_gen(rvo_prg *prg, uint8_t *dst, size_t len);
/* random access: keystream block number `blk` of key (used by tests) */
void rvo_prg_block(const uint8_t key[16], uint64_t blk, uint8_t out[16]);

#ifdef __cplusplus
}
#endif
#endif
