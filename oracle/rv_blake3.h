/* TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/README.md).
 * Plain (unkeyed) BLAKE3 hash + XOF, scalar C.
 *
 * Reverie uses the `blake3` crate (Cargo.toml:31, "1.0.0", not under /root/reference)
 * at /root/reference/src/crypto/hash.rs:14,18,31,39,48,54-56,121-125 (Hasher::new /
 * update / finalize) and src/crypto/ro.rs:9-19 (finalize_xof + fill). The algorithm
 * restated here is the published BLAKE3 specification (tree hash, 1 KiB chunks,
 * 64 B blocks, 7-round compression); it is pinned in tests against the official
 * BLAKE3 1.8.2 C build exported by libclang-cpp.so and committed known answers.
 */
#ifndef RV_ORACLE_BLAKE3_H
#define RV_ORACLE_BLAKE3_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t cv[8];
    uint64_t chunk_counter;
    uint8_t buf[64];
    uint8_t buf_len;
    uint8_t blocks_compressed;
} rvo_b3_chunk;

typedef struct {
    rvo_b3_chunk chunk;
    uint32_t stack[54][8];
    uint8_t stack_len;
} rvo_blake3;

void rvo_blake3_init(rvo_blake3 *h);
void rvo_blake3_update(rvo_blake3 *h, const void *data, size_t len);
/* non-consuming: may be called any number of times (hash.rs:53-57 clones) */
void rvo_blake3_finalize(const rvo_blake3 *h, uint8_t out[32]);
/* XOF: `len` output bytes starting at byte offset `seek` of the output stream */
void rvo_blake3_finalize_xof(const rvo_blake3 *h, uint64_t seek, uint8_t *out, size_t len);
/* one-shot */
void rvo_blake3_hash(const void *data, size_t len, uint8_t out[32]);

#ifdef __cplusplus
}
#endif
#endif
