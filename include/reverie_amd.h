/* reverie_amd — MI355X-native KKW prover/verifier hot path: C-ABI drop-in boundary.
 *
 * This header is what a Rust `-sys` shim (or any FFI) binds.  It replaces, at the only
 * seam where an FFI call is affordable, the reference's
 *
 *     Proof::new(circuit, wit_gf2, wit_z64, (z64_wires, gf2_wires)) -> Proof
 *                                            /root/reference/src/proof/mod.rs:119-222
 *     Proof::verify(&self, circuit, (z64_wires, gf2_wires)) -> bool
 *                                            /root/reference/src/proof/mod.rs:224-307
 *
 * and everything those two drive: src/generator (AES-CTR share expansion),
 * src/interpreter over src/algebra's packed GF(2)/Z64 rings, src/transcript, and the
 * BLAKE3 commitments / random oracle of src/crypto.  Proof bytes are bincode-1.3
 * compatible with the reference's `Proof`.
 *
 * Plain pointers and sizes only; no C++/torch types.  All functions return RV_OK (0) or
 * an RV_E_* code; nothing aborts the process (the reference panics).
 * A context is bound to ONE GPU and is thread-compatible: one in-flight call per context.
 */
#ifndef REVERIE_AMD_H
#define REVERIE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- protocol constants: /root/reference/src/lib.rs:17-38 ---- */
#define RV_PLAYERS 8
#define RV_PACKED 8
#define RV_BATCH_SIZE 128
#define RV_ONLINE_REPS 40
#define RV_TOTAL_REPS 256
#define RV_PREPROCESSING_REPS (RV_TOTAL_REPS - RV_ONLINE_REPS)
#define RV_KEY_SIZE 16  /* src/crypto/prg.rs:9  */
#define RV_HASH_SIZE 32 /* src/crypto/hash.rs:8 */

/* ---- gate stream ------------------------------------------------------------------
 * One record per `mcircuit::CombineOperation` (re-exported at src/lib.rs:7; consumed at
 * src/interpreter/combine.rs:120-132 and src/interpreter/single.rs:106-156).
 *   domain RV_DOM_GF2 / RV_DOM_Z64 : `Operation<bool>` / `Operation<u64>`, opcode below;
 *            fields: Input(dst) Random(dst) Add(dst,a,b) AddConst(dst,a,imm) Sub(dst,a,b)
 *            SubConst(dst,a,imm) Mul(dst,a,b) MulConst(dst,a,imm) AssertZero(a) Const(dst,imm)
 *            (GF2 constants use bit 0 of imm)
 *   domain RV_DOM_B2A      : B2A(dst = z64 wire, a = lowest of 64 consecutive gf2 wires)
 *   domain RV_DOM_SIZEHINT : SizeHint(a = z64 wire count, b = gf2 wire count)
 */
typedef struct rv_op {
    uint8_t domain;
    uint8_t opcode;
    uint16_t reserved; /* must be 0 */
    uint32_t dst;
    uint32_t a;
    uint32_t b;
    uint64_t imm;
} rv_op; /* 24 bytes */

enum { RV_DOM_GF2 = 0, RV_DOM_Z64 = 1, RV_DOM_B2A = 2, RV_DOM_SIZEHINT = 3 };
enum {
    RV_OP_INPUT = 0,
    RV_OP_RANDOM = 1,
    RV_OP_ADD = 2,
    RV_OP_ADDCONST = 3,
    RV_OP_SUB = 4,
    RV_OP_SUBCONST = 5,
    RV_OP_MUL = 6,
    RV_OP_MULCONST = 7,
    RV_OP_ASSERTZERO = 8,
    RV_OP_CONST = 9
};

/* ---- status codes (reference behaviour in parentheses) ---- */
enum {
    RV_OK = 0,
    RV_E_WITNESS_INVALID = 1, /* (panic, src/transcript/prover.rs:221-228) an AssertZero failed */
    RV_E_WITNESS_SHORT = 2,   /* (panic "witness is too short", prover.rs:190) */
    RV_E_WIRE_OOB = 3,        /* (Vec index panic, single.rs:109-155) wire index >= wire count */
    RV_E_PROOF_MALFORMED = 4, /* (bincode unwrap / assert panics; `omit >= 8` is UB upstream) */
    RV_E_BAD_OP = 5,          /* unknown domain/opcode or reserved != 0 */
    RV_E_NOMEM = 6,
    RV_E_DEVICE = 7, /* no usable gfx950 device / HIP runtime error (see rv_last_error) */
    RV_E_UNSUPPORTED = 8,
    RV_E_ARG = 9
};

typedef struct rv_ctx rv_ctx;         /* one GPU: stream, scratch arena                    */
typedef struct rv_circuit rv_circuit; /* a gate stream levelised and resident in HBM       */
typedef struct rv_shard rv_shard;     /* committed repetitions awaiting the challenge      */
typedef struct rv_stream rv_stream;   /* a bounded-memory proof in progress (two passes over a chunked gate stream) */
typedef struct rv_comm rv_comm;       /* this GPU's rank in a group of GPUs proving together (RCCL communicator)   */

const char *rv_strerror(int code);
/* last HIP/driver error text for this thread ("" if none) */
const char *rv_last_error(void);
/* library/ABI version, bumps on any signature change */
uint32_t rv_abi_version(void);

/* ---- context ---- */
int rv_ctx_create(int device_ordinal, rv_ctx **out);
void rv_ctx_destroy(rv_ctx *ctx);
/* A context owns the device memory of the circuits and shards created on it: destroy those first. */
/* block until all work queued on the context's stream has finished */
int rv_ctx_sync(rv_ctx *ctx);

/* ---- per-phase GPU timing, measured with HIP events on the context's own stream (the
 * stream every kernel of this library is launched on).  Phases: */
enum {
    RV_PH_SETUP = 0,  /* seed expansion, key schedules, round-key bitslicing          */
    RV_PH_MASKS = 1,  /* k_aes_gf2_masks: bitsliced AES-128-CTR mask generator         */
    RV_PH_INTERP = 2, /* k_interp_full / k_interp64: one launch per dependency level; k_interp_lds (or k_interp_narrow): one per narrow stretch */
    RV_PH_HASH = 3,   /* k_b3_chunks(_bits,_contig) + k_b3_reduce + k_b3_tree_tail: transcript BLAKE3 */
    RV_PH_JOIN = 4,   /* k_join                                                        */
    RV_PH_OPEN = 5,   /* k_fs_challenge + k_open_headers + k_extract_rows / k_extract_from_bits / k_extract64 */
    RV_PH_EARLY = 6,  /* launches only (their time is inside RV_PH_INTERP): k_pack_corr_all + k_publish of rv_prove's early-corrections path */
    RV_PH_CLEAR = 7,  /* unused since round 6 (the flat prover schedule's cleartext pass): the slot stays so that rv_profile keeps its size */
    RV_PH_COUNT = 8
};
typedef struct rv_profile {
    double ms[RV_PH_COUNT];         /* accumulated GPU milliseconds per phase */
    uint64_t launches[RV_PH_COUNT]; /* kernel launches per phase              */
    uint64_t calls;                 /* commit / verify_shard calls accumulated */
} rv_profile;
/* enable != 0 turns event timing on (a stream event per phase boundary: every one costs the call ~5 us of idle GPU; enable == 2: only
 * the interpreter's phase, RV_PH_INTERP, is timed -- two events per call); reset != 0 zeroes the accumulators; out (nullable) receives
 * the current totals. */
int rv_ctx_profile(rv_ctx *ctx, int enable, int reset, rv_profile *out);

/* ---- circuit: the `Arc<Vec<CombineOperation>>` + `wire_counts` arguments of
 * Proof::new / Proof::verify (proof/mod.rs:119-125,224,232).  Compiling resolves wire
 * reuse, orders gates into dependency levels, assigns every gate its PRG mask index and
 * transcript offsets, and uploads the result to HBM; it is reusable across proofs.
 * Errors that the reference raises while stepping (wire out of range, bad op) are
 * reported here. */
int rv_circuit_compile(rv_ctx *ctx, const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires,
                       rv_circuit **out);
/* The same with a hint about how the circuit will be used.  Every entry point accepts any compiled circuit and
 * produces identical bytes; the hint only chooses between gate streams that different users run at different speeds.
 * RV_COMPILE_WHOLE_PROVER: mostly proofs of all 256 repetitions on one GPU (rv_prove, rv_prove_device,
 * rv_prove_batch -- what Proof::new does, proof/mod.rs:119-175).  Linear gates of wide circuits are then kept as
 * lazy sums of up to three rows instead of being materialised: the whole-proof interpreter keeps one cleartext value
 * byte per row and reads full 256-byte rows, so the extra operand rows cost less than the Xor gates they replace (10^7-gate
 * benchmark circuit: rv_prove 6.4 -> 6.15 ms); the verifier and repetition shards (32-byte to 128-byte rows, corr-bit
 * rows per operand) run 3-15 % slower on such a stream, so rv_circuit_compile does not choose it. */
#define RV_COMPILE_WHOLE_PROVER 1u
int rv_circuit_compile_ex(rv_ctx *ctx, const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags,
                          rv_circuit **out);
void rv_circuit_destroy(rv_circuit *c);

typedef struct rv_circuit_info {
    uint64_t n_ops;
    uint64_t gf2_inputs, gf2_muls, gf2_asserts, gf2_linear; /* linear = gates with no transcript output */
    uint64_t gf2_masks;                                     /* ShareGen::next() calls per repetition       */
    uint64_t z64_inputs, z64_muls, z64_asserts, z64_linear, z64_masks;
    uint64_t b2a;
    uint64_t levels;         /* dependency levels = kernel launches of the interpreter */
    uint64_t device_bytes;   /* HBM held by the compiled circuit                       */
    uint64_t scratch_bytes;  /* HBM a full 256-repetition prove needs on top           */
    uint64_t compile_us;     /* host time of the gate-stream compiler                  */
    uint64_t upload_us;      /* allocation + host-to-device copy of the compiled gate stream (device_bytes), synchronised */
    /* ABI 4: share rows the GF(2) interpreter reads as gate operands (a Mul's two fresh mask rows not counted) and
     * computed rows it writes (materialised linear gates), per proof -- they depend on how linear gates were compiled */
    uint64_t gf2_operand_rows, gf2_rows_written;
} rv_circuit_info; /* (no size field: this struct does not grow -- later additions get getters of their own, like the one below) */
int rv_circuit_get_info(const rv_circuit *c, rv_circuit_info *info);
/* ABI 7 (ABI 6 had it as a field of rv_circuit_info): page-locked host memory rv_prove's early-corrections path stages this
 * circuit's corrections vectors in (0: the path does not apply to the circuit), as the RV_EARLY_* environment stands at the call.
 * Allocated once per context, on the first proof that takes the path (its first mapping costs 0.15 - 1.5 s), and kept;
 * RV_EARLY=0 proves without it.  The query builds a plan of its own and leaves the circuit's (made by its first proof) alone. */
int rv_circuit_early_staging_bytes(const rv_circuit *c, uint64_t *bytes);

/* ---- Proof::new -------------------------------------------------------------------
 * wit_gf2: one byte per GF(2) witness element (0/1), consumed by Input gates in order
 *          (the reference takes Vec<bool>);  wit_z64: u64 witness elements.
 * seeds:   256 x 16 bytes, one per repetition (the reference draws them from OsRng,
 *          proof/mod.rs:131-134).  NULL => drawn from the OS (getrandom).
 * *proof:  bincode(Proof), allocated by the library, released with rv_free.  Proofs of a megabyte and more come in
 *          page-locked memory from a small process-wide pool (the device-to-host copy runs at PCIe rate and
 *          rv_free recycles the buffer for the next proof); the pointer is ordinary readable/writable host memory.
 * Memory the call leaves on the context (kept for the next proof, released by rv_ctx_destroy): the device arena's cached blocks
 *          (rv_circuit_info::scratch_bytes), and -- for circuits that take the early-corrections path, pure GF(2) with >= 2^21
 *          Mul gates or pure Z64 with >= 2^17 -- page-locked staging of rv_circuit_early_staging_bytes() (160 MB for the
 *          10^7-gate GF(2) benchmark circuit, 2 GB for the 10^6-MUL Z64 one; its first mapping costs 0.15 - 1.5 s inside the first
 *          such proof) plus as much device memory for GF(2); RV_EARLY=0 in the environment proves without it. */
int rv_prove(rv_ctx *ctx, const rv_circuit *c, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64,
             size_t n_z64, const uint8_t *seeds, uint8_t **proof, size_t *proof_len);

/* ---- Proof::new / Proof::verify on the raw op list (SURVEY 8(b)'s signature) ------------------------------
 * What the reference's own entry points take (proof/mod.rs:119-125,224-232: the op list, the witness, (z64, gf2) wire
 * counts): compile (RV_COMPILE_WHOLE_PROVER for the prover) + prove / verify + release, in one call -- the form a drop-in for
 * a single Proof::new uses.  The compile runs on several host threads (csrc/compile_par.cpp): a circuit the library has not
 * seen costs ~0.1 s per 10^7 gates before its first proof.  The context keeps the circuits these two calls compile and finds them
 * again BY CONTENT: every kept circuit owns a copy of the op array it was compiled from (24 bytes per op of host memory) and a hit is
 * a full comparison of the caller's array against it on host threads (~4 ms per 10^7 ops) -- no hash is trusted, so neither the prover
 * nor the verifier can be handed another statement's gate stream (round 5 keyed the cache by an unkeyed 128-bit hash: constructible
 * collisions made rv_verify_ops accept a proof for a different circuit).  A second Proof::new on the same op list finds its gate stream
 * on the device and costs the comparison plus a proof (rv_prove's early-corrections path included).  At most RV_OPS_CACHE circuits per
 * context (environment, default 2; the least recently used one leaves BEFORE a new one is compiled; 0: nothing is kept), released by
 * rv_ctx_destroy or rv_ctx_ops_cache_clear.  Both calls therefore MUTATE the context (one call at a time per context, like every entry
 * point).  Callers that hold a circuit anyway compile it once (rv_circuit_compile_ex) and call rv_prove: no comparison.
 * flags of rv_verify_ops: as rv_verify_ex. */
int rv_prove_ops(rv_ctx *ctx, const rv_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64, size_t n_z64,
                 size_t z64_wires, size_t gf2_wires, const uint8_t *seeds, uint8_t **proof, size_t *proof_len);
int rv_verify_ops(rv_ctx *ctx, const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, const uint8_t *proof, size_t proof_len,
                  uint32_t flags, int *ok);
int rv_ctx_ops_cache_clear(rv_ctx *ctx); /* releases the circuits rv_prove_ops / rv_verify_ops keep (ABI 7) */

/* ---- Proof::new with the openings left in device memory ------------------------------------
 * The whole prover (commit, Fiat-Shamir, openings) with ONE host synchronisation, for callers that keep working on
 * the GPU (bench.py's HBM-resident metric): writes [gf2 online | gf2 preprocessing | z64 online | z64 preprocessing]
 * (lens[4]) to dst_device and returns comm and the opening map; rv_assemble_proof frames them as bincode(Proof).
 * Capacity of dst_device: 40 * record sizes (rv_circuit_record_sizes) + 2 * 216 * 48.  seeds must not be NULL. */
int rv_prove_device(rv_ctx *ctx, const rv_circuit *c, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64,
                    size_t n_z64, const uint8_t *seeds, void *dst_device, uint8_t comm[RV_HASH_SIZE],
                    uint8_t omit[RV_TOTAL_REPS], size_t lens[4]);

/* ---- many proofs of one circuit ------------------------------------------------------------
 * `batch` independent Proof::new calls (witness b at wit_gf2 + b*n_gf2 / wit_z64 + b*n_z64, seeds b at
 * seeds + b*256*16, NULL => OS randomness) executed together; the reference reaches the same goal with one rayon
 * task per proof.  Small and medium circuits: every dependency level of the circuit AND every per-proof phase (keys,
 * masks, digests, Fiat-Shamir, openings) is launched once for the whole batch (gridDim.y = proof) -- this is what
 * makes deep, narrow circuits (AES, SHA-256: thousands of levels of a few gates, latency-bound for one proof) use the
 * GPU.  Circuits of 2^20 gates and more fill the GPU on their own: there a few host threads keep several proofs in
 * flight so that one proof's VALU-bound phases, another's memory-bound interpreter and a third one's PCIe copy overlap.
 * proofs[b] / proof_lens[b] as rv_prove (rv_free each, exactly once).  The proofs of a call are slices of one
 * page-locked buffer that goes back to the library's pool when the last of them has been freed (RV_BATCH_COPY_OUT=1:
 * separately malloc'ed buffers, small-circuit path only).  Each proof is byte-identical to what rv_prove returns for
 * the same witness and seeds.  Mixed / Z64 circuits fall back to one rv_prove per proof. */
int rv_prove_batch(rv_ctx *ctx, const rv_circuit *c, size_t batch, const uint8_t *wit_gf2, size_t n_gf2,
                   const uint64_t *wit_z64, size_t n_z64, const uint8_t *seeds, uint8_t **proofs, size_t *proof_lens);

/* ---- Proof::verify ----------------------------------------------------------------
 * Verification is STRICT by default (flags 0, rv_verify): on top of the reference's check (*ok = 0 when a ProofSingle
 * has the wrong number of repetitions or the recomputed commitment differs) it closes the two soundness gaps of the
 * reference verifier:
 *   - every AssertZero of the 40 opened repetitions must reconstruct to zero -- VerifierTranscriptOnline.okay
 *     (src/transcript/verifier/online.rs:21,117,175-177), which the reference computes and never reads, so
 *     its Proof::verify accepts a proof of an unsatisfied circuit;
 *   - every online record's `omit` must equal the player the challenge omits -- the reference only checks which
 *     repetitions are opened (src/proof/mod.rs:292-302: contains_key), never by whom.
 * RV_VERIFY_REFERENCE_COMPAT switches both extra checks off: *ok is then exactly what the reference's Proof::verify
 * returns (byte-compatibility tests; never for untrusted proofs).  RV_VERIFY_STRICT is accepted and means flags 0;
 * both bits together are RV_E_ARG.
 * Bytes that cannot be parsed as a Proof, unequal GF(2) opening lengths inside a verifier group, or an
 * `omit` value >= 8 return RV_E_PROOF_MALFORMED (the reference panics / is UB). */
#define RV_VERIFY_STRICT 1u
#define RV_VERIFY_REFERENCE_COMPAT 2u
int rv_verify(rv_ctx *ctx, const rv_circuit *c, const uint8_t *proof, size_t proof_len, int *ok);
int rv_verify_ex(rv_ctx *ctx, const rv_circuit *c, const uint8_t *proof, size_t proof_len, uint32_t flags, int *ok);

void rv_free(void *p);

/* ---- streaming prover --------------------------------------------------------
 * The reference's README promises "a streaming interface" over its single-pass just-in-time preprocessing
 * (/root/reference/README.md:14,38; src/generator/share.rs:54-65), while the surveyed source keeps every
 * reconstruction and correction of all repetitions until the challenge (src/transcript/prover.rs:29-31,211,217).
 * Here the gate stream is fed in pieces and device memory is bounded by
 *     the wire store (one share row per GF(2) wire INDEX, one slot per Z64 wire index -- the reference's own
 *     `wires` vectors, sized by wire_counts) + one chunk's working set + the proof itself,
 * independent of the number of gates: transcripts are hashed chunk by chunk into incremental BLAKE3 trees and
 * dropped.  Because the omitted players are only known after the commitment, the SAME ops are fed twice:
 *
 *     rv_stream_begin(ctx, z64_wires, gf2_wires, seeds, max_chunk_ops, &s)
 *     rv_stream_feed(s, ops_0, ...) ... rv_stream_feed(s, ops_k, ...)      pass 1: commitments
 *     rv_stream_commit(s, comm)                                            Fiat-Shamir challenge
 *     rv_stream_feed(s, ops_0, ...) ... rv_stream_feed(s, ops_k, ...)      pass 2: openings of the 40 challenged reps
 *     rv_stream_finish(s, &proof, &len)        bincode(Proof), byte-identical to rv_prove's for the same seeds
 *     rv_stream_abort(s)                       releases the stream (also after a finish or an error)
 *
 * The pieces of pass 2 need not be cut where those of pass 1 were, but their concatenation must be the same op
 * list and the same witness (both digested as they are fed, also when pass 1's compiled chunk is reused; checked at
 * finish: RV_E_ARG).  wit_gf2 / wit_z64 of a feed are the witness elements its Input gates
 * consume, in order (more may be passed; RV_E_WITNESS_SHORT if fewer).  A feed longer than max_chunk_ops
 * (0 = 2^18) is cut into device chunks of that size (a long feed of chunks >= 2^16 ops starts with pieces of 1/8, 1/4, 1/2).  SizeHint ops may not grow the wire counts given at begin
 * (RV_E_UNSUPPORTED).  After an error the stream only accepts rv_stream_abort. */
int rv_stream_begin(rv_ctx *ctx, size_t z64_wires, size_t gf2_wires, const uint8_t *seeds /* 256 x 16 or NULL */,
                    size_t max_chunk_ops, rv_stream **out);
int rv_stream_feed(rv_stream *s, const rv_op *ops, size_t n_ops, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64,
                   size_t n_z64);
int rv_stream_commit(rv_stream *s, uint8_t comm[RV_HASH_SIZE] /* nullable */);
int rv_stream_finish(rv_stream *s, uint8_t **proof, size_t *proof_len);
void rv_stream_abort(rv_stream *s);
/* A promise, made during pass 1: pass 2 will be fed in exactly pass 1's pieces (the same rv_stream_feed calls).  Pass 1 then keeps
 * the transcripts of the stream's LAST chunks on the device, as many as fit a budget (RV_STREAM_KEEP_MB; default an eighth of the
 * device's memory, at most 32 GiB; 0 = none; rv_stream_info.kept_mib), and pass 2 takes the openings of those chunks from them
 * instead of running them a second time: the earlier chunks run twice, as every chunk does without the promise.  Device memory
 * stays bounded by wire store + one chunk + the proof + that budget.  A pass 2 that breaks the promise gets RV_E_ARG from the feed
 * that would have to run a chunk after one that did not.  rv_prove_streaming makes the promise itself. */
int rv_stream_same_cuts(rv_stream *s);
typedef struct rv_stream_info {
    uint64_t n_ops, chunks, levels; /* of pass 1 (running totals while it is in progress) */
    uint64_t gf2_masks, z64_masks, gf2_muls, z64_muls;
    uint64_t wire_store_bytes;  /* HBM held by the carried wires + the largest chunk's rows */
    uint64_t peak_chunk_bytes;  /* largest single chunk's working set (rows, transcripts, gate records) */
    uint64_t hash_state_bytes;  /* incremental BLAKE3 trees + unhashed stream tails */
    uint64_t proof_bytes;       /* 0 before rv_stream_commit */
    uint32_t pass;
    uint32_t kept_mib;          /* most MiB of pass-1 transcripts held for pass 2 (RV_STREAM_KEEP_MB; the field was `reserved`, always 0) */
} rv_stream_info;
int rv_stream_get_info(const rv_stream *s, rv_stream_info *info);
/* Both passes over an op array that already sits in host memory: Proof::new with bounded DEVICE memory.
 * info (nullable) receives the stream's final figures. */
int rv_prove_streaming(rv_ctx *ctx, const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, const uint8_t *wit_gf2,
                       size_t n_gf2, const uint64_t *wit_z64, size_t n_z64, const uint8_t *seeds, size_t max_chunk_ops,
                       uint8_t **proof, size_t *proof_len, rv_stream_info *info);

/* ---- Proof::verify with bounded device memory (the streaming verifier) -----------------------------------------------
 * The reference's verify walks the op list like its prover (proof/mod.rs:259-261,276-278); rv_verify keeps the whole compiled
 * circuit and its rows resident (~6 GB for 10^7 GF(2) gates, ~90 GB for 10^6 Z64 multiplications).  Here the ops are fed in
 * pieces, ONCE (the omitted players are in the proof), and device memory is the streaming prover's: wire store + one chunk +
 * the proof.
 *     rv_stream_verify_begin(ctx, z64_wires, gf2_wires, proof, proof_len, max_chunk_ops, &s)   (proof must outlive the stream)
 *     rv_stream_feed(s, ops_0, n, NULL, 0, NULL, 0) ... rv_stream_feed(s, ops_k, ...)           (no witness)
 *     rv_stream_verify_finish(s, flags, &ok)      flags as rv_verify_ex; ok = what rv_verify_ex answers for the same ops
 *     rv_stream_abort(s)
 * A proof with the wrong repetition counts gives ok = 0 (as rv_verify); a malformed one RV_E_PROOF_MALFORMED at begin. */
int rv_stream_verify_begin(rv_ctx *ctx, size_t z64_wires, size_t gf2_wires, const uint8_t *proof, size_t proof_len, size_t max_chunk_ops,
                           rv_stream **out);
int rv_stream_verify_finish(rv_stream *s, uint32_t flags, int *ok);
/* begin + one feed of an op array that already sits in host memory + finish; info (nullable): the stream's final figures */
int rv_verify_streaming(rv_ctx *ctx, const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, const uint8_t *proof, size_t proof_len,
                        uint32_t flags, size_t max_chunk_ops, int *ok, rv_stream_info *info);

/* ---- sharded form (one process per GPU; repetitions [rep_begin, rep_begin+rep_count),
 * both multiples of 8).  rv_prove == commit(0,256) -> combine -> challenge -> open ->
 * assemble.  Between commit and open the caller exchanges the 32-byte per-repetition
 * digests (one all-gather; proof/mod.rs:160-172 is the reference's gather point). */
int rv_shard_commit(rv_ctx *ctx, const rv_circuit *c, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64,
                    size_t n_z64, const uint8_t *seeds /* rep_count x 16 */, uint32_t rep_begin, uint32_t rep_count,
                    rv_shard **out);
/* device pointer to rep_count x 32 digest bytes (valid until rv_shard_destroy), for RCCL */
int rv_shard_digests_device(rv_shard *s, void **dptr);
/* host copy of the same bytes */
int rv_shard_digests(rv_shard *s, uint8_t *out /* rep_count x 32 */);
/* device-to-device copy of the same bytes into caller-owned HBM (e.g. a torch tensor that
 * RCCL will all-gather); complete when the call returns */
int rv_shard_digests_to_device(rv_shard *s, void *dst_device);
/* Opens the shard's repetitions for the full challenge (omit[256], 8 = preprocessing).
 * Returns four blobs: the shard's OpenOnline / OpenPreprocessing records, in ascending
 * repetition order, already in bincode form, for the gf2 and z64 ProofSingle.
 * Each blob is library-allocated (rv_free). */
typedef struct rv_shard_parts {
    uint8_t *gf2_online, *gf2_pre, *z64_online, *z64_pre;
    size_t gf2_online_len, gf2_pre_len, z64_online_len, z64_pre_len;
    uint32_t n_online, n_pre;
} rv_shard_parts;
int rv_shard_open(rv_shard *s, const uint8_t omit[RV_TOTAL_REPS], rv_shard_parts *parts);
void rv_shard_destroy(rv_shard *s);

/* Device-resident variant used by bench.py: leaves the shard's four blobs concatenated
 * in HBM (gf2_online | gf2_pre | z64_online | z64_pre) and returns the device pointer,
 * valid until the shard is destroyed. */
int rv_shard_open_device(rv_shard *s, const uint8_t omit[RV_TOTAL_REPS], void **dptr, size_t lens[4]);
/* Bytes of one OpenOnline record of the gf2 / z64 ProofSingle for this circuit (every record of a
 * section has the same size; an OpenPreprocessing record is always 48 bytes).  Lets every rank
 * compute every other rank's blob sizes from the challenge alone. */
int rv_circuit_record_sizes(const rv_circuit *c, size_t *gf2_online_record, size_t *z64_online_record);
/* Sizes rv_shard_open* will produce for this challenge, without opening */
int rv_shard_open_size(const rv_shard *s, const uint8_t omit[RV_TOTAL_REPS], size_t lens[4]);
/* As rv_shard_open_device, but writes the concatenated blobs into caller-owned HBM
 * (sum of rv_shard_open_size bytes), e.g. a torch tensor handed to RCCL afterwards */
int rv_shard_open_into(rv_shard *s, const uint8_t omit[RV_TOTAL_REPS], void *dst_device, size_t lens[4]);
/* Fiat-Shamir without leaving the device, for a shard that holds ALL 256 repetitions (rep_begin 0, rep_count 256):
 * comm = BLAKE3(digests) (combine_hashes, proof/mod.rs:102-108), the challenge (RandomOracle + challenge_to_opening,
 * crypto/ro.rs:8-20, proof/mod.rs:68-83) and the openings, with no host round trip in between.  Equivalent to
 * rv_shard_digests -> rv_combine_digests -> rv_challenge -> rv_shard_open_into; returns comm and the opening map.
 * dst_device capacity: the sum rv_shard_open_size reports for ANY 40/216 map (sizes do not depend on which
 * repetitions open).  RV_E_ARG for a partial shard. */
int rv_shard_open_self(rv_shard *s, void *dst_device, uint8_t comm[RV_HASH_SIZE], uint8_t omit[RV_TOTAL_REPS], size_t lens[4]);
/* The sharded counterpart: all_digests_device = the 256 x 32 bytes every rank holds on its GPU after the all-gather.
 * Commitment, challenge and this shard's openings without a host round trip; comm / omit as above.  How many of the
 * shard's repetitions open depends on the challenge, so dst_device must hold the worst case:
 *   min(40, rep_count) * (gf2 record + z64 record) + 2 * rep_count * 48 bytes (rv_circuit_record_sizes);
 * lens[4] reports what was written: [gf2 online | gf2 preprocessing | z64 online | z64 preprocessing], contiguous. */
int rv_shard_open_gathered(rv_shard *s, const void *all_digests_device, void *dst_device, uint8_t comm[RV_HASH_SIZE],
                           uint8_t omit[RV_TOTAL_REPS], size_t lens[4]);

/* ---- multi-GPU inside the library ---------------------------------------------------------------
 * The reference fans its 32 packed groups out over a rayon pool INSIDE Proof::new (proof/mod.rs:127-157) and meets
 * again at one point, combine_hashes over the 256 digests (:160-172).  A communicator does the same over GPUs: rank r
 * of `world` (1, 2, 4, 8, 16 or 32) proves repetitions [r*256/world, (r+1)*256/world); the one data-path collective
 * is an ncclAllGather of the 32-byte digests over RCCL/xGMI on the library's own stream; every rank derives the
 * challenge on its GPU and opens its own repetitions; the openings go to rank 0 with ncclSend/ncclRecv straight into
 * their place in the proof.  RCCL is bound at run time (dlopen; RV_RCCL_PATH overrides the search): the library
 * loads without it and reuses a copy the process already has (PyTorch's).
 *   one process per GPU : rank 0 calls rv_comm_unique_id and hands the 128 bytes to the others out of band (MPI,
 *                         torch.distributed, a file); every rank calls rv_comm_create with its own context
 *   one process, n GPUs : rv_comm_create_all(ctxs, n, comms), then rv_prove_multi (a host thread per GPU; its ranks send nothing to
 *                         rank 0: every rank copies its own sections straight into ONE page-locked proof buffer over its own PCIe
 *                         link -- the all-gather of digests stays the only collective)
 * rv_prove_sharded is a COLLECTIVE call: every rank calls it with the same statement and the same 256 seeds (all of
 * them, not only its share; NULL is not allowed -- the ranks could not agree on OS randomness); *proof is set on rank
 * 0 only (NULL / 0 elsewhere) and is byte-identical to rv_prove's.  If one rank fails before the collective the
 * others wait for it: destroy the communicator.  circuits[i] / c must have been compiled on the rank's own context. */
#define RV_COMM_ID_BYTES 128
int rv_comm_unique_id(uint8_t id[RV_COMM_ID_BYTES]);
int rv_comm_create(rv_ctx *ctx, int world, int rank, const uint8_t id[RV_COMM_ID_BYTES], rv_comm **out);
int rv_comm_create_all(rv_ctx *const *ctxs, int n, rv_comm **comms /* [n] */);
void rv_comm_destroy(rv_comm *comm);
int rv_prove_sharded(rv_comm *comm, const rv_circuit *c, const uint8_t *wit_gf2, size_t n_gf2, const uint64_t *wit_z64, size_t n_z64,
                     const uint8_t *seeds /* 256 x 16 */, uint8_t **proof, size_t *proof_len);
/* seeds NULL => drawn once from the OS and shared by the ranks */
int rv_prove_multi(rv_comm *const *comms, const rv_circuit *const *circuits, int n, const uint8_t *wit_gf2, size_t n_gf2,
                   const uint64_t *wit_z64, size_t n_z64, const uint8_t *seeds, uint8_t **proof, size_t *proof_len);

/* combine_hashes (proof/mod.rs:102-108): comm = BLAKE3(h[0] || ... || h[255]) */
int rv_combine_digests(const uint8_t *h /* 256 x 32 */, uint8_t comm[RV_HASH_SIZE]);
/* challenge_to_opening (proof/mod.rs:74-83): omit[r] in 0..7 for the 40 online reps, else 8 */
int rv_challenge(const uint8_t comm[RV_HASH_SIZE], uint8_t omit[RV_TOTAL_REPS]);
/* Concatenates shard parts (ordered by rep_begin) into bincode(Proof) */
int rv_assemble_proof(const uint8_t comm[RV_HASH_SIZE], const rv_shard_parts *parts, size_t n_parts, uint8_t **proof,
                      size_t *proof_len);
/* Verifier side of the sharded form: recomputes the digests of the verifier's
 * repetition slots [slot_begin, slot_begin+slot_count) (slots 0..39 = online openings in
 * proof order, 40..255 = preprocessing openings; proof/mod.rs:234-281), multiples of 8. */
int rv_verify_shard(rv_ctx *ctx, const rv_circuit *c, const uint8_t *proof, size_t proof_len, uint32_t slot_begin,
                    uint32_t slot_count, uint8_t *digests /* slot_count x 32 */);
/* Final check of Proof::verify (proof/mod.rs:283-306) from all 256 slot digests: the reference's check and nothing
 * else (this form has no zero-check input, so it cannot be strict; use the _ex pair below for untrusted proofs). */
int rv_verify_finish(const uint8_t *proof, size_t proof_len, const uint8_t *slot_digests /* 256 x 32 */, int *ok);
/* The sharded form of rv_verify_ex: *zero_checks_ok (nullable) = 0 when an AssertZero of one of this shard's opened
 * repetitions did not reconstruct to zero; AND the shards' values together and hand the result to
 * rv_verify_finish_ex, which (unless RV_VERIFY_REFERENCE_COMPAT) requires it and also compares the records' `omit`
 * with the challenge. */
int rv_verify_shard_ex(rv_ctx *ctx, const rv_circuit *c, const uint8_t *proof, size_t proof_len, uint32_t slot_begin,
                       uint32_t slot_count, uint8_t *digests /* slot_count x 32 */, int *zero_checks_ok);
int rv_verify_finish_ex(const uint8_t *proof, size_t proof_len, const uint8_t *slot_digests /* 256 x 32 */, uint32_t flags,
                        int zero_checks_ok, int *ok);

/* Many proofs of one circuit in one pass (the verifier's counterpart of rv_prove_batch; pure GF(2) circuits below the
 * large-circuit threshold -- everything else verifies proof after proof): ok[b] as rv_verify_ex would set it for
 * proofs[b] with the same flags.  A proof whose bytes cannot be parsed (or whose records rv_verify_ex would answer with
 * RV_E_PROOF_MALFORMED) is a REJECTED proof, ok[b] = 0, and the others are verified all the same; a non-zero return
 * code means an argument or device error. */
int rv_verify_batch(rv_ctx *ctx, const rv_circuit *c, size_t batch, const uint8_t *const *proofs, const size_t *proof_lens,
                    uint32_t flags, int *ok /* [batch] */);

/* ---- Bristol front end (host only, no GPU) ---------------------------------------------
 * The reference's README promises Bristol-format circuits; the parser itself lives in the
 * un-vendored `mcircuit` crate.  This turns Bristol text into the rv_op stream:
 *   wires 0..n_in-1            -> GF2 Input(w), in wire order (= witness order)
 *   XOR -> Add, AND -> Mul, INV / NOT -> AddConst(.., 1), EQW -> AddConst(.., 0), EQ -> Const,
 *   MAND -> one Mul per lane
 *   if expected_outputs != NULL: for each output wire (the last n_out wires, in order)
 *   AddConst(tmp, w, expected) + AssertZero(tmp)  — the statement "the circuit maps the witness
 *   to these outputs" (SURVEY §8d configs 1-3); n_expected must equal the circuit's output count (RV_E_ARG otherwise,
 *   before anything is read from expected_outputs).
 * format: 0 = auto, 1 = Bristol Fashion ("ngates nwires / niv n.. / nov n.."), 2 = old Bristol
 * ("ngates nwires / n1 n2 n3").  ops is library-allocated (rv_free). */
typedef struct rv_bristol_info {
    uint64_t n_gates, n_wires, n_inputs, n_outputs;
    uint64_t n_and, n_xor, n_inv, n_other;
    uint64_t gf2_wires; /* wire count to pass to rv_circuit_compile (includes assertion temporaries) */
} rv_bristol_info;
int rv_bristol_parse(const char *text, size_t len, int format, const uint8_t *expected_outputs, size_t n_expected, rv_op **ops,
                     size_t *n_ops, rv_bristol_info *info);

/* ---- program files (host only, no GPU) ---------------------------------------------------
 * The reference CLI reads its gate stream as bincode 1.3 of Vec<mcircuit::CombineOperation>
 * (src/main.rs:66,98,122).  rv_program_from_bincode turns such a file into rv_op records, rv_program_to_bincode
 * writes one.  The enum's variant order comes from the un-vendored `mcircuit` crate and is the one SURVEY A.7
 * recalls (GF2, Z64, B2A, SizeHint; Input, Random, Add, AddConst, Sub, SubConst, Mul, MulConst, AssertZero, Const):
 * it cannot be verified here, so nothing selects this format automatically.  Wire indices above 2^32-1 ->
 * RV_E_UNSUPPORTED; a truncated file, an unknown variant or a bool that is not 0/1 -> RV_E_BAD_OP.  Outputs are
 * library-allocated (rv_free). */
int rv_program_from_bincode(const uint8_t *data, size_t len, rv_op **ops, size_t *n_ops);
int rv_program_to_bincode(const rv_op *ops, size_t n_ops, uint8_t **data, size_t *len);

/* ---- parity-test hooks: each mirrors one reference function so tests can compare the
 * HIP path with the oracle piecewise ---- */
/* PRG::new + gen (crypto/prg.rs:16-37) on the GPU: n_keys keys, blocks [first, first+n_blocks) each */
int rv_hook_prg_blocks(rv_ctx *ctx, const uint8_t *keys, size_t n_keys, uint64_t first_block, size_t n_blocks,
                       uint8_t *out /* n_keys x n_blocks x 16 */);
/* expand_seed (transcript/mod.rs:99-106) for n seeds -> n x 8 x 16 key bytes */
int rv_hook_expand_seed(rv_ctx *ctx, const uint8_t *seeds, size_t n, uint8_t *keys);
/* ShareGen<GF2>::next() x n for one packed group (generator/share.rs:54-65): keys 8x8x16,
 * omit[8] (8 = none) -> n packed u64 shares in the reference's bit order */
int rv_hook_sharegen_gf2(rv_ctx *ctx, const uint8_t *keys, const uint32_t omit[8], size_t n, uint64_t *out);
/* ShareGen<Z64>::next() x n -> n x 8 x 8 u64 */
int rv_hook_sharegen_z64(rv_ctx *ctx, const uint8_t *keys, const uint32_t omit[8], size_t n, uint64_t *out);
/* The gate-stream compiler alone (host only, no device: ctx-free): what rv_circuit_compile_ex would report through
 * rv_circuit_get_info -- the counters that are pure functions of the op list (ShareGen::next() calls per repetition,
 * generator/share.rs:54-65; transcript events, prover.rs:194,210,216), dependency levels, operand rows -- and the errors the
 * reference raises while stepping (wire out of range, bad op).  device_bytes / scratch_bytes / upload_us stay zero.
 * chunk_first_ops (0 = off): also checks, like a streaming feed cut every that many ops would, that every piece compiled
 * on its own with the ShareGen phases predicted from the ops before it adds up to the whole (RV_E_DEVICE if not). */
int rv_hook_compile_info(const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, size_t chunk_ops,
                         rv_circuit_info *info);
/* The two gate-stream compilers against each other (host only): the program is compiled by the sequential compiler and by the
 * parallel one with `threads` host threads (>= 2), whatever its size, and the results are compared field by field.
 * *diff = 0: identical; > 0: a number naming the first differing table (csrc/compile_par.cpp, compiled_diff); -1: the parallel
 * compiler declined the program (B2A gates, an error in the op list) -- the sequential result is what rv_circuit_compile uses.
 * Returns the sequential compiler's status (RV_OK or the error the reference raises while stepping, single.rs:106-156). */
int rv_hook_compile_compare(const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, int threads, int *diff);
/* DomainGF2::reconstruct (gf2/domain.rs:47-63) on n packed u64 shares (bit 63 - (8*rep + player)) -> n ReconGF2 words
 * (one 0x00/0xFF byte per repetition), through the interpreter's own device function */
int rv_hook_gf2_reconstruct(rv_ctx *ctx, const uint64_t *shares, size_t n, uint64_t *out);
/* DomainZ64::reconstruct (z64/domain.rs:53-61) on n ShareZ64 values ([8 reps][8 players] u64) -> n x 8 wrapping sums */
int rv_hook_z64_reconstruct(rv_ctx *ctx, const uint64_t *shares /* n x 64 */, size_t n, uint64_t *out /* n x 8 */);
/* BLAKE3 of n_streams independent byte strings of equal length len (row-major), on the GPU
 * tree-hash kernels used for the transcripts (crypto/hash.rs:17-57) */
int rv_hook_blake3(rv_ctx *ctx, const uint8_t *data, size_t n_streams, size_t len, uint8_t *out /* n x 32 */);
/* per-stream digests of a committed shard: rep_count x 4 x 32 = H_pre(gf2), H_on(gf2), H_pre(z64), H_on(z64) */
int rv_hook_shard_stream_digests(rv_shard *s, uint8_t *out);
/* Proofs this process has produced through rv_prove's early-corrections path (csrc/api.hip, rv_prove_impl: for large GF(2)
 * circuits the corrections vectors of all repetitions -- Pack of ReconGF2, gf2/recon.rs:189-239, half of what
 * ProverTranscript::extract returns, prover.rs:57-175 -- cross PCIe before the challenge exists).  The bytes are the same
 * either way; the tests use the counter to know which path they compared.  RV_EARLY=0 turns the path off. */
uint64_t rv_hook_early_proofs(void);
/* ... and those of them whose opened repetitions' broadcast vectors (the omitted player's shares, prover.rs:57-175) were written into
 * the page-locked proof buffer by the extraction kernel itself (csrc/internal.h: OpenDirect; RV_OPEN_DIRECT=0/1 turns it off).  Same
 * bytes either way. */
uint64_t rv_hook_open_direct_proofs(void);
/* rv_prove_ops / rv_verify_ops calls of this process that found their op list's compiled circuit in the context's cache (ABI 7). */
uint64_t rv_hook_ops_cache_hits(void);
/* Host only, no device: the comparison a cache lookup of rv_prove_ops / rv_verify_ops decides on -- 1 iff the two ranges hold the same
 * bytes (parallel memcmp, first difference ends it), 0 if not, -1 on a NULL range. */
int rv_hook_ops_same(const void *a, const void *b, size_t bytes);
/* Shard commitments of this process whose GF(2) mask generator ran BESIDE the interpreter's level launches (round 5: the
 * lane-distributed cipher of csrc/aes_col4.hip on a stream of its own, chunk by chunk; RV_OVERLAP=0 runs it before the first level;
 * circuits below RV_OVERLAP_MIN = 8192 cipher blocks and rows narrower than 64 repetitions keep that order anyway).  Same bytes. */
uint64_t rv_hook_overlap_commits(void);
/* Verifications this process has run with one u64 of public corrections per share row instead of corr rows (csrc/kernels.hip:
 * MODE_VERIFY_C -- the verify-mode interpreter of whole proofs of pure GF(2) one-base gate streams; replaces nothing of the
 * reference's: verifier/online.rs:122-183 computes the same values).  The answer is the same either way; the tests use the
 * counter to know which path they compared.  RV_VERIFY_VC=0 turns the path off. */
uint64_t rv_hook_verify_vc_count(void);
/* The early-corrections plan of a program (host only, no device): the ops are compiled as rv_circuit_compile_ex(flags) would and
 * the plan rv_prove would use is built and checked against the compiled gate records.  out[0] = a plan exists (0 / 1: the circuit
 * is pure GF(2) with >= 2^21 Mul gates -- RV_EARLY_MIN -- or pure Z64, its preprocessing rows complete in step with the levels
 * and fit the PCIe window), [1] = Z64 form, [2] = repetitions staged, [3] = chunks, [4] = staging bytes, [5] = 1 when no level
 * after a chunk's ready level writes one of its rows and the ready level itself does, [6 + k] = chunk k's ready level (k < 16).
 * Returns the compiler's status. */
int rv_hook_early_plan(const rv_op *ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, uint32_t flags, uint64_t out[22]);

#ifdef __cplusplus
}
#endif
#endif /* REVERIE_AMD_H */
