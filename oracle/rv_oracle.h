/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the data-parallel KKW hot path of trailofbits/reverie
 * (reverie-zk 0.3.2).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may build, load or call anything in this directory; the product
 * (reverie_amd/, include/) never links it and fails loudly without its HIP library.
 *
 * What it follows (all paths relative to /root/reference/):
 *   src/lib.rs:17-38                      protocol constants
 *   src/crypto/{prg,hash,ro}.rs           PRG / buffered BLAKE3 / random oracle
 *   src/generator/{batch,share}.rs        BatchGen / ShareGen
 *   src/algebra/{gf2,z64}/ (all files)       packed rings, (un)packing, hashing
 *   src/transcript/{mod,prover}.rs, verifier/{online,preprocess}.rs
 *   src/interpreter/{single,combine}.rs   gate semantics, B2A
 *   src/proof/mod.rs:40-307               Proof::new / Proof::verify, bincode layout
 *
 * Pinning status (see oracle/README.md and DESIGN.md):
 *   - AES-128-CTR pinned against OpenSSL libcrypto + FIPS-197 C.1.
 *   - BLAKE3 (hash + XOF) pinned against the official BLAKE3 1.8.2 C build in
 *     libclang-cpp.so and committed known answers.
 *   - Every known-answer test the reference's own test-suite holds for this path
 *     (gate-value tests with all-zero seeds, pack/unpack round trips, omitted-player
 *     share consistency, prove->verify acceptance) is re-expressed in tests/ against
 *     this oracle.
 *   - Whole-proof bytes: the reference holds NO golden proof vectors, seeds itself from
 *     OsRng, and cannot be built here (Rust, no toolchain).  Proof bytes are therefore
 *     pinned against an independently written per-repetition scalar spec model
 *     (tests/golden/gen_golden.py), not against the Rust binary: "parity unpinned"
 *     at that one level.
 */
#ifndef RV_ORACLE_H
#define RV_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* protocol constants, lib.rs:17-38 */
#define RVO_PLAYERS 8
#define RVO_PACKED 8
#define RVO_BATCH 128
#define RVO_ONLINE_REPS 40
#define RVO_TOTAL_REPS 256
#define RVO_PRE_REPS (RVO_TOTAL_REPS - RVO_ONLINE_REPS)
#define RVO_GROUPS (RVO_TOTAL_REPS / RVO_PACKED)

/* gate record: same 24-byte layout as include/reverie_amd.h `rv_op` */
typedef struct {
    uint8_t domain; /* 0 GF2, 1 Z64, 2 B2A, 3 SizeHint */
    uint8_t opcode; /* Operation variant for domain 0/1 */
    uint16_t reserved;
    uint32_t dst;
    uint32_t a;
    uint32_t b;
    uint64_t imm;
} rvo_op;

enum { RVO_DOM_GF2 = 0, RVO_DOM_Z64 = 1, RVO_DOM_B2A = 2, RVO_DOM_SIZEHINT = 3 };
enum {
    RVO_OP_INPUT = 0,
    RVO_OP_RANDOM = 1,
    RVO_OP_ADD = 2,
    RVO_OP_ADDCONST = 3,
    RVO_OP_SUB = 4,
    RVO_OP_SUBCONST = 5,
    RVO_OP_MUL = 6,
    RVO_OP_MULCONST = 7,
    RVO_OP_ASSERTZERO = 8,
    RVO_OP_CONST = 9
};

enum {
    RVO_OK = 0,
    RVO_E_WITNESS_INVALID = 1,
    RVO_E_WITNESS_SHORT = 2,
    RVO_E_WIRE_OOB = 3,
    RVO_E_PROOF_MALFORMED = 4,
    RVO_E_BAD_OP = 5,
    RVO_E_NOMEM = 6
};

/* Proof::new (proof/mod.rs:119-222) with injected per-repetition seeds.
 * wit_gf2: one byte per witness bit (0/1).  threads<=0 -> 1.  Caller frees *proof
 * with rvo_free. */
int rvo_prove(const rvo_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2,
              const uint64_t *wit_z64, size_t n_z64, size_t z64_wires, size_t gf2_wires,
              const uint8_t *seeds /* [256][16] */, int threads, uint8_t **proof, size_t *proof_len);

/* Proof::verify (proof/mod.rs:224-307).  *ok = 1/0 as the reference's bool. */
int rvo_verify(const rvo_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires,
               const uint8_t *proof, size_t proof_len, int threads, int *ok);

/* strict != 0: additionally require every AssertZero of the 40 opened repetitions to reconstruct to zero
 * (VerifierTranscriptOnline.okay, which the reference computes but never reads -- SURVEY F9) and every online
 * record's `omit` to equal the challenge's (the reference only checks which repetitions are opened).  This is the
 * boundary's RV_VERIFY_STRICT, a deliberate tightening, not reference behaviour. */
int rvo_verify_ex(const rvo_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires,
                  const uint8_t *proof, size_t proof_len, int threads, int strict, int *ok);

void rvo_free(void *p);

/* ---- hooks used by parity tests (mirror SURVEY §8 rows a1..a18) ---- */

/* expand_seed (transcript/mod.rs:99-106): rep seed -> 8 player keys */
void rvo_expand_seed(const uint8_t seed[16], uint8_t keys[8][16]);

/* ShareGen<GF2>::next() x n (generator/share.rs:54-65).  keys [8 reps][8 players][16],
 * omit[8] (8 = none).  out: n packed u64 shares. */
void rvo_sharegen_gf2(const uint8_t *keys, const uint32_t omit[8], size_t n, uint64_t *out);
/* ShareGen<Z64>::next() x n.  out: n x [8 reps][8 players] u64. */
void rvo_sharegen_z64(const uint8_t *keys, const uint32_t omit[8], size_t n, uint64_t *out);

/* DomainGF2::reconstruct (gf2/domain.rs:47-63) */
uint64_t rvo_gf2_reconstruct(uint64_t share);

/* Commitment phase only: per-rep digests.  Any of the outputs may be NULL.
 *   h[256][32]          BLAKE3(h_gf2 || h_z64)             (combine.rs:104-118)
 *   streams[256][4][32] H_pre(gf2), H_on(gf2), H_pre(z64), H_on(z64)
 *   comm[32]            combine_hashes                      (proof/mod.rs:102-108) */
int rvo_commit(const rvo_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2,
               const uint64_t *wit_z64, size_t n_z64, size_t z64_wires, size_t gf2_wires,
               const uint8_t *seeds, int threads, uint8_t *h, uint8_t *streams, uint8_t *comm);

/* challenge_to_opening (proof/mod.rs:74-83): omit[r] in 0..7 for online reps, 8 otherwise */
void rvo_challenge(const uint8_t comm[32], uint8_t omit[256]);

/* Prover-side wire values after running `ops` for ONE packed group with the 8 given
 * rep seeds (mirrors the helpers at interpreter/single.rs:177-229).
 * gf2_out: recon(mask)+corr of gf2 wire `gf2_wire` (packed, one byte per rep)
 * z64_out[8]: value of z64 wire `z64_wire` per rep.  Either may be NULL. */
int rvo_group_wire_values(const rvo_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2,
                          const uint64_t *wit_z64, size_t n_z64, size_t z64_wires,
                          size_t gf2_wires, const uint8_t *seeds8 /* [8][16] */, uint32_t gf2_wire,
                          uint64_t *gf2_out, uint32_t z64_wire, uint64_t *z64_out);

/* Pack / PackSelected round-trip hooks (algebra/mod.rs:210-409).
 * pack: src n elements -> dst[8] byte vectors (selected reps only).  The caller provides
 * dst buffers of capacity cap each; lens[8] receives the byte counts. */
void rvo_gf2_recon_pack(const uint64_t *src, size_t n, const uint8_t selected[8], uint8_t *dst,
                        size_t cap, size_t lens[8]);
size_t rvo_gf2_recon_unpack(const uint8_t *src /* [8][len] */, size_t len, uint64_t *dst);
void rvo_gf2_share_pack_selected(const uint64_t *src, size_t n, const uint32_t selected[8],
                                 uint8_t *dst, size_t cap, size_t lens[8]);
size_t rvo_gf2_share_unpack_selected(const uint8_t *src /* [8][len] */, size_t len,
                                     const uint32_t selected[8], uint64_t *dst);

#ifdef __cplusplus
}
#endif
#endif
