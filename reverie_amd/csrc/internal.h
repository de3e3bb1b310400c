// Internal (non-ABI) declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/reverie_amd.h"

namespace rv {

// ---- HBM data layout -------------------------------------------------------------
// A shard holds R repetitions (multiple of 8).  Four repetitions x eight players are
// packed into one 32-bit word ("quad word"): repetition i4 = r % 4 sits in byte i4
// counted from the most significant byte, player p at bit 31 - (8*i4 + p).  Two
// consecutive quad words (hi, lo) are exactly the reference's packed u64 share
// (src/algebra/gf2/share.rs:13-24: bit 63 - (8*rep + player)).
// A "row" is NQ = R/4 consecutive quad words = one value for every repetition of the
// shard; with R = 256 a row is 256 bytes = one wavefront-wide coalesced access.
//   rows   [n_rows][NQ]         share rows.  Rows [0, n_masks_pad) are the fresh PRG masks
//                               (index = ShareGen::next() call number, written by the AES
//                               kernel); the rest are computed wire masks (XOR outputs) with
//                               row n_masks_pad = the all-zero row.  A wire whose mask IS a
//                               PRG mask (Input/Random/Mul outputs) or equals another wire's
//                               (AddConst, MulConst) just points at that row: nothing is copied.
//   corr   [n_rows][NQ/2] bytes corrections of the BASE wires (indexed by row id), ONE BIT per repetition (the reference keeps a
//                               0x00/0xFF byte, src/algebra/gf2/recon.rs:13-25): quad q owns
//                               nibble q&1 of byte q/2, nibble bit k <-> repetition 4q + 3 - k
//   on     [n_on_events][NQ]    online transcript: one byte per rep per event
//   pre    [n_pre_events][NQ/2] preprocessing transcript (corrections), one bit per rep, same
//                               nibble layout; expanded to 0x00/0xFF bytes when hashed

// compiled gate (device + host).  XOR / AddConst / MulConst / Const never reach the device: the
// compiler keeps every wire as "XOR of at most RV_LIN_K base rows, plus a constant" and only
// materialises a row (G_XORK) when that set outgrows RV_LIN_K or is re-read often enough to pay
// for itself.  A base row is either a PRG mask row (Input / Random / Mul outputs: their mask IS a
// fresh mask) or a computed row; corr bits are stored per ROW id.
constexpr int RV_LIN_K = 3;
struct Gate {
    uint32_t op;   // GateOp | na << 8 | nb << 12 | ca << 16 | cb << 17   (operand base counts, operand constants)
    uint32_t dst;  // row written: G_XORK / G_RECON a computed row, G_MUL m + 1, G_INPUT / G_RANDOM m
    uint32_t m;    // first PRG mask index consumed (Input/Random: 1 mask, Mul: 2)
    uint32_t eo;   // row in the online transcript (Input, Mul, AssertZero, Recon)
    uint32_t ep;   // row in the preprocessing transcript (Mul)
    uint32_t x;    // Input: witness index; Mul/AssertZero/Recon: reconstruction ordinal
    uint32_t a[RV_LIN_K], b[RV_LIN_K];  // base rows of operand a / b (G_XORK: up to 2K bases in a then b);
                                        // unused slots hold the zero row
};

enum GateOp : uint32_t {
    G_INPUT = 0,
    G_XORK,    // dst = XOR of the listed base rows (mask and corr), corr ^= ca
    G_MUL,
    G_ASSERT,
    G_RANDOM,  // fresh mask m, corr 0
    G_RECON    // B2A: transcript.reconstruct(mask) + corr, result kept as {mask 0, corr value}
};
__host__ __device__ inline uint32_t g_op(const Gate& g) { return g.op & 0xFFu; }
__host__ __device__ inline uint32_t g_na(const Gate& g) { return (g.op >> 8) & 0xFu; }
__host__ __device__ inline uint32_t g_nb(const Gate& g) { return (g.op >> 12) & 0xFu; }
__host__ __device__ inline uint32_t g_ca(const Gate& g) { return (g.op >> 16) & 1u; }
__host__ __device__ inline uint32_t g_cb(const Gate& g) { return (g.op >> 17) & 1u; }

// Z64 ring (src/algebra/z64): one u64 per (repetition, player).
//   masks64 [n_masks64][R*8] u64   (row m = m-th ShareGen<Z64>::next(), [rep][player])
//   wmask64 [n_ssa64][R*8]  u64,  wcorr64 [n_ssa64][R] u64
//   on64 / pre64: per-repetition CONTIGUOUS transcripts [R][n_words] u64 (events have two
//   sizes, 8 B and 64 B per rep, so a row layout would not have a fixed stride)
struct Gate64 {
    uint32_t op;   // Gate64Op
    uint32_t dst;  // z64 SSA id
    uint32_t a, b; // z64 SSA ids (B2A: a = first of 64 consecutive gf2 SSA ids holding the revealed sum bits)
    uint32_t m;    // z64 mask index (Input/Random/B2A: 1, Mul: 2)
    uint32_t m2;   // B2A: first of the 64 fresh gf2 mask indices
    uint64_t eo;   // word offset in the per-rep online transcript
    uint64_t ep;   // word offset in the per-rep preprocessing transcript
    uint32_t x;    // Input: witness index / input ordinal; Mul, AssertZero: reconstruction ordinal
    uint32_t xc;   // Mul, B2A: correction ordinal
    uint64_t imm;  // constants
    uint32_t am, bm;  // where the operands' mask rows live: a wmask row (= the SSA id) or G64_MASK_ROW | PRG mask row
};
constexpr uint32_t G64_MASK_ROW = 0x80000000u;

enum Gate64Op : uint32_t {
    G64_INPUT = 0, G64_ADD, G64_SUB, G64_ADDC, G64_SUBC, G64_MULC, G64_MUL, G64_ASSERT, G64_RANDOM, G64_CONST, G64_B2A
};

struct Interp64Params {
    uint32_t R;
    uint64_t* wmask;
    uint64_t* wcorr;
    const uint64_t* masks;
    uint64_t* on;        // [R][on_words]
    uint64_t* pre;       // [R][pre_words]
    uint64_t on_words, pre_words;
    const uint64_t* wit;
    const uint8_t* omit;      // verify: [R] omitted player of online-verified reps, 8 otherwise
    const uint64_t* sup_in;   // verify: [n_in64][R]
    const uint64_t* sup_corr; // verify: [n_corr64][R]
    const uint64_t* sup_rec;  // verify: [n_rec64][R] share of the omitted player
    uint32_t sup_r;           // verify: repetitions per row of the three sup_* arrays (R, or 64 when the opened ones are the first 64 at most)
    // gf2 side, for B2A
    const uint8_t* corr2;     // compact gf2 corr rows
    const uint32_t* masks2;   // gf2 share rows (PRG part)
    uint32_t NQ;
    int* err;
};

// MODE_PROVE_V: the prover without corr rows.  A wire's public correction is  value XOR reconstruct(mask), and the prover
// knows every wire's cleartext value -- one bit per share row, the same in all repetitions (InterpParams::vclr, maintained
// by the interpreter itself level by level).  The three 32-byte corr-row accesses of a gate become one-byte accesses at
// wave-uniform addresses.  Not for circuits with Random gates / B2A (values that differ between repetitions) and only
// in the one-launch-per-level kernels (a level reads what the previous LAUNCH wrote).
// MODE_VERIFY_C: the verifier of a whole proof with one u64 of public corrections per share row (the opened repetitions' quad
// words, InterpParams::vc) instead of corr rows; full-width rows, every level launched on its own, no Random / B2A gates.
enum Mode : int { MODE_PROVE = 0, MODE_VERIFY = 1, MODE_PROVE_V = 2, /* 3: the flat schedule of rounds 4 - 5, gone */ MODE_VERIFY_C = 4 };
// bit set in the device error word when an AssertZero of an online-verified repetition does not reconstruct to zero
// (VerifierTranscriptOnline.okay, online.rs:175-177; only the strict verifier looks at it)
constexpr int RV_DEV_ZERO_CHECK = 0x100;

struct InterpParams {
    uint32_t NQ;
    uint32_t* rows;           // share rows (PRG masks + computed)
    uint8_t* corr;            // [n_rows][NQ/2]
    uint32_t* on;
    uint8_t* pre;             // [n_pre][NQ/2]
    uint8_t* vclr;            // MODE_PROVE_V: [n_rows] cleartext value of every share row's wire (0 / 1)
    uint64_t* vc;             // MODE_VERIFY_C: [n_rows] nibble q = the corr bits of quad word q (q < 16: the opened repetitions' quad words)
    const uint8_t* wit;       // prover: witness bits, one byte each
    const uint32_t* on_mask;  // verify: [NQ] 0xFF byte per online-verified rep
    const uint32_t* sup_in;   // verify: [n_inputs][NQ] supplied masked inputs (smeared)
    const uint32_t* sup_corr; // verify: [n_mul][NQ] supplied corrections (smeared)
    const uint32_t* sup_rec;  // verify: [n_rec][NQ] supplied broadcast bit of the omitted player
    uint32_t sup_nq;          // verify: quad words per row of the three sup_* arrays (NQ, or 16 when every opened repetition sits in the first sixteen)
    int* err;                 // device flag: RV_E_WITNESS_INVALID
};

// LDS per workgroup of the device in use (set by rv_ctx_create): the dynamic-LDS kernels raise their limit to it, not beyond
void set_device_lds_limit(size_t bytes);
size_t device_lds_limit();

// ---- launchers (implemented in the .hip files) ----
void launch_expand_seeds(hipStream_t st, const uint8_t* d_seeds, uint32_t n_reps, uint8_t* d_keys /*[n][8][16]*/);
// Per AES key: the 11 round keys (176 bytes) followed by 32 bytes of first-round constants (k_key_schedule):
// CTR blocks with index j < 2^24 differ from the all-zero block only in bytes 13..15, so SubBytes of round 1 on
// bytes 0..12, most of its MixColumns and the four round-2 S-boxes fed by state column 3 depend on the key alone.
//   [176..179] SK   = S(round-1 output column 3)            (round-2 S-box outputs of state bytes 12..15)
//   [189..191] rk0[13..15]                                  (the three bytes that meet the counter)
//   [192..207] K1   = round-1 output with S(byte 13..15) taken as zero (columns 0..2), the constant column 3
// Bitsliced they form two more 128-word areas after the 11 round keys.
constexpr uint32_t RK_BYTES = 208;
constexpr uint32_t RK_AREAS = 13;
constexpr uint64_t RV_MAX_CTR_BLOCKS = 1ull << 24;
void launch_key_schedule(hipStream_t st, const uint8_t* d_keys, uint32_t n_slots, uint8_t* d_rkbytes /*[n][RK_BYTES]*/);
// expand_seeds + key_schedule + bitslice_rk (+ the lane-distributed generator's key image, d_img != null) of an NQ-quad-word shard in one launch
void launch_setup_keys(hipStream_t st, const uint8_t* d_seeds, uint32_t NQ, uint8_t* d_keys, uint8_t* d_rkbytes, uint32_t* d_rk, uint32_t* d_img);
void launch_bitslice_rk(hipStream_t st, const uint8_t* d_rkbytes, uint32_t NQ, uint32_t* d_rk /*[RK_AREAS][128][NQ]*/);
void launch_aes_gf2_masks(hipStream_t st, const uint32_t* d_rk, const uint32_t* d_keep, uint32_t NQ, uint64_t first_block,
                          uint64_t n_blocks, uint32_t* d_masks);
// the lane-distributed generator (aes_col4.hip: a quad of lanes per bitsliced state, 80 registers, same rows): its key image
// [NQ / 16][88 KiB] comes from the plane-major round keys of launch_bitslice_rk
bool aes_col4_supports(uint32_t NQ);
size_t aes_col4_image_bytes(uint32_t NQ);
void launch_rk_col4(hipStream_t st, const uint32_t* d_rk, uint32_t NQ, uint32_t* d_img);
void launch_aes_gf2_masks_col4(hipStream_t st, const uint32_t* d_img, const uint32_t* d_keep, uint32_t NQ, uint64_t first_block, uint64_t n_blocks,
                               uint32_t* d_masks);
void launch_aes_blocks(hipStream_t st, const uint8_t* d_rkbytes, uint32_t n_keys, uint64_t first_block, uint64_t n_blocks,
                       uint8_t* d_out);
// per-level class boundaries: [lo, mul11) G_MUL with one base per operand, [mul11, mul) other G_MUL,
// [mul, xor2) G_XORK of two bases, [xor2, xork) other G_XORK, [xork, hi) anything else
struct LevelRange {
    uint32_t lo, mul11, mul, xor2, xork, hi;
};
// some level has enough multi-base Mul / Xor gates for the kernel variant with their loops (kernels.hip: level_is_general)
bool persist_general(const LevelRange* lr, size_t n_levels);
// next: the level launched after this one by launch_interp too (nullable) -- the tail of this launch prefetches
// its first gate records
void launch_interp(hipStream_t st, int mode, const Gate* d_gates, const LevelRange& r, const InterpParams& p,
                   const LevelRange* next = nullptr);
// levels [l0, l1) (all narrow, GF(2) only) in one launch by a single workgroup
// tiny: plain per-gate loop (one gate per wavefront) instead of the 4-way unrolled class loops
void launch_interp_narrow(hipStream_t st, int mode, const Gate* d_gates, const LevelRange* d_level_range, uint32_t l0, uint32_t l1,
                          int tiny, const InterpParams& p);
// rv_prove_batch: the same level / narrow-run launches for `batch` proofs at once (prover, NQ = 64); d_pp = device
// array of one InterpParams per proof
void launch_interp_batched(hipStream_t st, const Gate* d_gates, const LevelRange& r, const InterpParams* d_pp, uint32_t batch,
                           int mode = MODE_PROVE);
// large proofs: the two transcripts' trees in shared launches (kernels.hip); four ping-pong buffers of b3_stream_scratch_words each
// (d_quads / n_quads as in launch_b3_stream: the verifier's online stream of the quad words with an opened repetition)
bool b3_pair_big_ok(uint64_t n_pre, uint64_t n_on, uint32_t NQ, const uint32_t* d_quads = nullptr, uint32_t n_quads = 0);
uint32_t launch_b3_pair_big(hipStream_t st, const uint8_t* d_pre, uint64_t n_pre, const uint32_t* d_on, uint64_t n_on, uint32_t NQ, uint32_t* cv_a0,
                            uint32_t* cv_a1, uint32_t* cv_b0, uint32_t* cv_b1, uint32_t* d_dig_pre, uint32_t* d_dig_on, const uint32_t* d_quads = nullptr,
                            uint32_t n_quads = 0);
bool launch_b3_pair_small(hipStream_t st, const uint8_t* d_pre, uint64_t n_pre, const uint32_t* d_on, uint64_t n_on, uint32_t NQ, uint32_t* d_cv_a,
                          uint32_t* d_cv_b, uint32_t* d_dig_pre, uint32_t* d_dig_on, const uint32_t* d_quads = nullptr, uint32_t n_quads = 0);
void launch_store_word(hipStream_t st, const int* d_src, int* dst_mapped);
// early corrections (kernels.hip, api.hip: rv_prove on large GF(2) circuits)
void launch_pack_corr_all(hipStream_t st, const uint8_t* d_bits, uint64_t n_items, uint64_t byte0, uint64_t n_bytes, uint64_t pitch, uint8_t* d_out);
// OpenDirect: the first n_direct tiles of the opened repetitions' broadcast vectors (tile = `tile` bytes, a power of two, of every
// opened repetition; vector at record + rvec_at, rvec_len bytes) are written to the page-locked proof buffer by the extraction kernel
// itself -- its first workgroups; whole 16-byte aligned words only -- and k_copy_gaps leaves those words out (it still copies the
// word across every tile boundary).  n_direct = 0: off.
struct OpenDirect {
    uint32_t n_direct = 0, tile = 0;
    uint64_t rvec_at = 0, rvec_len = 0;
};
void launch_copy_gaps(hipStream_t st, const uint8_t* d_img, uint8_t* dst_mapped, uint64_t total, uint64_t first, uint64_t rec, uint64_t corr_at,
                      uint64_t corr_len, uint32_t n_rec, const uint8_t* d_omit /*[256]*/, uint32_t rep_limit, OpenDirect od = OpenDirect(),
                      const int* d_err = nullptr, int* err_dst_mapped = nullptr /* also: the error word into a host-mapped word */);
uint32_t extract_tile_bytes(uint64_t n_items);  // the tile of launch_extract_bits for vectors of n_items bits
void launch_publish(hipStream_t st, const uint32_t* d_src, uint32_t n_words, uint32_t* dst_mapped, uint32_t* flag_mapped, uint32_t seq);
void launch_store_words(hipStream_t st, const uint32_t* d_src, uint32_t n_words, uint32_t* dst_mapped, const int* d_err, int* dst_err_mapped);
// a narrow stretch with its live wires in LDS (ldsrun.h); d_pp != null: `batch` proofs, parameters from the device array
struct LdsRec;
void launch_interp_lds(hipStream_t st, int mode, uint32_t QS, uint32_t NQ, const LdsRec* d_recs, uint32_t n_steps, uint32_t n_slots,
                       uint32_t eo0, uint32_t ep0, const InterpParams& p, const InterpParams* d_pp, uint32_t batch);
void launch_interp_narrow_batched(hipStream_t st, const Gate* d_gates, const LevelRange* d_level_range, uint32_t l0, uint32_t l1,
                                  int tiny, const InterpParams* d_pp, uint32_t batch, int mode = MODE_PROVE);
void launch_interp64(hipStream_t st, int mode, const Gate64* d_gates, uint32_t lo, uint32_t hi, const Interp64Params& p);
// Z64 masks: masks64[m][slot] = LE64(keystream[slot][8m..8m+8)), blocks [first, first+n_blocks) -> masks 2*first..
void launch_aes_z64_masks(hipStream_t st, const uint32_t* d_rk, const uint32_t* d_keep, uint32_t NQ, uint64_t n_blocks,
                          uint64_t* d_masks64, uint64_t first_block = 0);
// The Z64 prover with the mask generator INSIDE the interpreter (round 4; aes.hip: k_z64_fused).  A Mul's two fresh masks
// (lambda_ab = row m, lambda_new = row m + 1, m even) are exactly one cipher block per (repetition, player) stream, so the
// lane that runs the bitsliced cipher for counter m / 2 and the 32 slots of a quad word holds both rows of its 4 repetitions
// x 8 players in registers: lambda_ab never reaches HBM, lambda_new is stored once (later gates read it as an operand), and
// the gate's row traffic is issued by a wavefront whose neighbour on the SIMD is busy with its own cipher rounds.  Public
// corrections are not kept at all: the prover knows the wires' cleartext values (v: one u64 per Z64 SSA id, the same in every
// repetition; corr = value - reconstruct(mask), the sum over a repetition's 8 players being lane-local in this mapping).
// Eligible: Input / Add / Sub / AddConst / SubConst / MulConst / Mul / AssertZero / Const gates only (Random and B2A values
// differ between repetitions), every Mul's m even, NQ a multiple of 16.
struct Z64FLevel {
    uint32_t mul0, mul1, lin1, oth1;  // gates [mul0, mul1) Mul, [mul1, lin1) Add .. MulConst, [lin1, oth1) Input / AssertZero / Const
};
struct Z64FParams {
    const uint32_t* rk;  // bitsliced round keys (k_bitslice_rk)
    uint32_t NQ;
    uint64_t* wmask;     // [ssa][R * 8]
    uint64_t* masks;     // [mask row][R * 8]
    uint64_t* on;        // [R][on_words]
    uint64_t* pre;       // [R][pre_words]
    uint64_t on_words, pre_words;
    const uint64_t* wit;
    uint64_t* v;         // [ssa] cleartext values
    int* err;
    uint64_t first_block;  // counter of mask rows 0, 1
    uint32_t qg0, qgn;     // the launch covers quad groups [qg0, qg0 + qgn) of the NQ / 16 (rows of different groups never meet)
    // the verifier (omit != null): which player each repetition hides (8: none, a preprocessing-only repetition), the kept streams
    // per quad word (k_aes_z64_masks' keep), per-repetition public corrections [ssa][R] instead of `v`, and the proof's values
    const uint8_t* omit;
    const uint32_t* keep;
    uint64_t* wcorr;
    const uint64_t* sup_in;
    const uint64_t* sup_corr;
    const uint64_t* sup_rec;
    uint32_t sup_r;
};
bool z64_fused_supports(uint32_t NQ);
uint32_t z64_fused_qw(uint32_t NQ);  // quad words per workgroup block of that path for rows of NQ quad words (16 or 8; 0: not taken)
void launch_z64_fused(hipStream_t st, const Gate64* d_gates, const Z64FLevel& lv, const Z64FParams& p);
// BLAKE3 of R contiguous streams of n_words u64 each -> digests[R][8]
uint32_t launch_b3_contig(hipStream_t st, const uint64_t* d_streams, uint64_t n_words, uint32_t R, uint32_t* d_cv_a, uint32_t* d_cv_b,
                      uint32_t* d_digest);
void launch_extract64(hipStream_t st, const uint64_t* d_stream, uint64_t stride_words, const uint64_t* d_offs /*[n_items]*/,
                      uint64_t n_items, int add_omit, uint32_t R, const uint8_t* d_omit, const uint64_t* d_dst_off, uint8_t* d_out,
                      const struct OnlineList* d_ol = nullptr /* device: the shard's opened repetitions; given, only those get threads */,
                      uint32_t rep_min = 0 /* with d_ol: repetitions below this one are left out */);
void launch_unpack64(hipStream_t st, const uint8_t* d_blob, const uint64_t* d_src_off, const uint64_t* d_src_len,
                     const uint8_t* d_omit, uint64_t n_items, uint32_t R, uint64_t* d_out, uint32_t out_r /* repetitions per output row (<= R) */);
// BLAKE3 over a row-format transcript: digests[R][8] words
// d_quads / n_quads (nullable): hash only the listed quad words (the verifier's opened repetitions)
uint32_t launch_b3_stream(hipStream_t st, const uint32_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv_a,
                      uint32_t* d_cv_b, uint32_t* d_digest /*[R][8]*/, const uint32_t* d_quads = nullptr, uint32_t n_quads = 0);
// same for a bit-per-rep transcript [n_events][NQ/2] (each bit hashed as a 0x00/0xFF byte)
uint32_t launch_b3_stream_bits(hipStream_t st, const uint8_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv_a,
                           uint32_t* d_cv_b, uint32_t* d_digest);
size_t b3_stream_scratch_words(uint64_t n_events, uint32_t R);
// streaming prover: chunk chaining values of a PIECE of a stream (first chunk counter chunk_base; a lone chunk is
// the root only if root_ok), one level of the incremental tree, and the final fold
void launch_b3_stream_chunks(hipStream_t st, const uint32_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv, const uint32_t* d_quads,
                             uint32_t n_quads, uint64_t chunk_base, uint32_t root_ok);
void launch_b3_stream_bits_chunks(hipStream_t st, const uint8_t* d_stream, uint64_t n_events, uint32_t NQ, uint32_t* d_cv, uint64_t chunk_base,
                                  uint32_t root_ok);
// streams [R][stride_words] u64, the first n_words of each hashed
void launch_b3_contig_chunks(hipStream_t st, const uint64_t* d_streams, uint64_t stride_words, uint64_t n_words, uint32_t R, uint32_t* d_cv,
                             uint64_t chunk_base, uint32_t root_ok);
void launch_b3_pairs(hipStream_t st, const uint32_t* d_pending, const uint32_t* d_in, uint64_t n_pairs, uint32_t R, uint32_t* d_out);
struct B3FoldList {
    uint32_t n;
    const uint32_t* p[48];  // pending subtree roots, smallest subtree first ([R][8] each)
};
void launch_b3_fold(hipStream_t st, const B3FoldList& L, const uint32_t* d_last, uint32_t R, uint32_t* d_digest);
uint32_t b3_reduce_tree(hipStream_t st, uint32_t* cur, uint32_t* nxt, uint64_t n, uint32_t R, uint32_t* d_digest);
void launch_join(hipStream_t st, const uint32_t* d_pre2, const uint32_t* d_on2, const uint32_t* d_pre64, const uint32_t* d_on64,
                 uint32_t R, uint8_t* d_h /*[R][32]*/);
// the (at most 40) opened repetitions of a shard with the output offset of one of their vectors;
// passed by value so the kernel reads it from the kernel-argument segment (scalar loads)
struct OnlineList {
    uint32_t n;
    uint32_t rep[RV_ONLINE_REPS];
    uint64_t dst[RV_ONLINE_REPS];
};
void launch_extract_from_bits(hipStream_t st, const uint8_t* d_bits, uint64_t n_items, uint32_t NQ,
                              const OnlineList* d_ol /* device */, uint8_t* d_out, uint32_t rep_min = 0 /* repetitions below it are left out */);
// Fiat-Shamir on the device for a shard that holds all 256 repetitions (proof/mod.rs:68-108,158-175):
// comm = BLAKE3(h[0..256)), the challenge map, and from it everything the opening kernels consume
// (omit[256], the 8 x 256 output offsets, the OnlineList) without a host round trip.
struct FsLayout {
    uint64_t base[4];  // framed: the four section starts (gf2 online, gf2 preprocessing, z64 online, z64 preprocessing);
                       // otherwise only base[0] = start of the output, the rest follow from the number of opened reps
    uint64_t sz2, sz64, l2r, l2c, l64r, l64c;
    uint32_t framed;
    uint8_t* comm2;    // nullable: a second destination for comm (the head of a framed proof buffer)
};
// d_h: all 256 digests.  Produces for the shard (rep_begin, R): d_omit[R], d_offs[8*R], *d_ol, and d_res = {opened,
// not opened} repetition counts; d_comm[32]; d_omit_all[256] (nullable) = the whole opening map
void launch_fs_challenge(hipStream_t st, const uint8_t* d_h, const FsLayout& L, uint32_t rep_begin, uint32_t R, uint8_t* d_comm,
                         uint8_t* d_omit, uint8_t* d_omit_all, uint64_t* d_offs, OnlineList* d_ol, uint32_t* d_res,
                         uint32_t* mbox = nullptr /* host-mapped: comm, the opening map, the counts for the host; then *mbox_flag = mbox_seq */,
                         uint32_t* mbox_flag = nullptr, uint32_t mbox_seq = 0);
// one small GF(2) proof's openings (heads + the three kinds of vectors [+ the error word]) in ONE launch (kernels.hip: k_open_small);
// false: not taken (a recorded batch, long vectors) -- the caller launches the pieces
bool launch_open_small(hipStream_t st, uint32_t R, const uint8_t* d_omit, const uint8_t* d_seeds, const uint8_t* d_keys, const uint32_t* d_on2,
                       const uint32_t* d_on64, const uint64_t* d_offs, uint64_t l2r, uint64_t l2c, uint64_t l2i, uint64_t l64r, uint64_t l64c, uint64_t l64i,
                       const uint32_t* d_on, const uint32_t* d_rec_rows, uint64_t n_rec, const uint8_t* d_pre, uint64_t n_pre, const uint32_t* d_in_rows,
                       uint64_t n_in, uint32_t NQ, const OnlineList* d_ol, uint32_t corr_rep_min, uint8_t* d_out, const int* d_err, int* err_dst_mapped,
                       bool heads_inputs_only = false);
// kind 0: omitted player's bit of a share row; 1: smeared byte of a row
// d_out2 / n_direct (kind 0, rv_prove's early path): the first n_direct tiles ALSO go to d_out2 (the proof buffer's device address), same offsets
void launch_extract_bits(hipStream_t st, const void* d_stream, const uint32_t* d_rows /*nullable*/, uint64_t n_items,
                         uint32_t NQ, int kind, const uint8_t* d_omit, const uint64_t* d_dst_off, uint8_t* d_out, uint8_t* d_out2 = nullptr,
                         uint32_t n_direct = 0);
void launch_unpack_bits(hipStream_t st, const uint8_t* d_blob, const uint64_t* d_src_off, const uint64_t* d_src_len,
                        const uint8_t* d_omit, uint64_t n_items, uint32_t NQ, int kind, uint32_t* d_rows_out, uint32_t out_nq /* row stride in quad words */,
                        uint64_t first_item = 0 /* the vectors' item row 0 of the output is */);
void launch_shard_init(hipStream_t st, int* d_err, uint32_t* d_zero_mask, uint32_t n_mask_words /* <= 64 */, uint8_t* d_zero_corr,
                       uint32_t n_corr_bytes /* <= 64 */, uint32_t* d_fill = nullptr /* also: n_fill_rows copies of digest[8] */, uint32_t n_fill_rows = 0,
                       const uint32_t* digest = nullptr, uint8_t* d_zero_byte = nullptr /* also: one byte cleared */);
void launch_fill_digests(hipStream_t st, uint32_t* d_dst, uint32_t n_rows, const uint32_t digest[8]);
void launch_overlay_rows(hipStream_t st, uint32_t* d_dst, const uint32_t* d_src, const uint8_t* d_omit, uint32_t R,
                         uint32_t row_words, int want_online);
void launch_hook_recon_gf2(hipStream_t st, const uint64_t* d_shares, uint64_t n, uint64_t* d_out);
void launch_hook_recon_z64(hipStream_t st, const uint64_t* d_shares /*[n][8][8]*/, uint64_t n, uint64_t* d_out /*[n][8]*/);
void launch_open_headers(hipStream_t st, uint32_t R, const uint8_t* d_omit, const uint8_t* d_seeds, const uint8_t* d_keys,
                         const uint32_t* d_on2, const uint32_t* d_on64, const uint64_t* d_off2, const uint64_t* d_off64,
                         uint64_t lens2_rec, uint64_t lens2_corr, uint64_t lens2_in, uint64_t lens64_rec, uint64_t lens64_corr,
                         uint64_t lens64_in, uint8_t* d_out);


}  // namespace rv
