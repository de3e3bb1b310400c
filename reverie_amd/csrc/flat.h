// "Flat" prover schedule of a compiled GF(2) circuit (round 4): the gate stream without its dependency levels.
//
// The level-synchronous interpreter (kernels.hip) runs the gates of dependency level l after every gate of level
// l - 1: 163 launches for the 10^7-gate benchmark circuit, each 1.5 generations of short-lived wavefronts.  For the
// PROVER of a pure GF(2) circuit almost none of these dependencies is real (interpreter/single.rs:25-69 read again):
//   * the mask of a Mul / Input output wire IS a fresh PRG mask (single.rs:27,58): it exists before the interpreter
//     starts, whatever level its gate sits on;
//   * what a Mul needs of its operands besides their masks is their public correction c = value - reconstruct(mask)
//     (prover.rs:181-199): the prover knows every wire's cleartext value, one bit that is the same in all 256
//     repetitions -- a plain evaluation of the circuit, done once per proof by a small kernel of its own (k_clear)
//     beside the mask generator;
//   * only a materialised XOR row (G_XORK) is computed from other rows -- and only from PRG rows and earlier XOR rows.
// So the schedule is: (1) the XOR rows, level-synchronous over XOR -> XOR dependencies only ("x-levels": a handful,
// the first ones holding nearly all gates); (2) every Mul of the circuit in PROGRAM order, in as few launches as the
// early-corrections chunks ask for -- mask rows, transcript rows and preprocessing bits become sequential streams;
// (3) the few Input / AssertZero transcript rows.  Proof bytes are unchanged: every gate computes what it computed
// before (tests/test_gpu_parity.py runs both schedules against the oracle).
#pragma once
#include <stdint.h>

#include <vector>

#include "compile.h"
#include "internal.h"

namespace rv {

// a Mul gate of the flat schedule, index = its preprocessing row (= correction ordinal, prover.rs:209-219)
struct MulRec {
    uint32_t a[RV_LIN_K], b[RV_LIN_K];  // operand base rows (unused slots: the zero row)
    uint32_t m;                         // lambda_ab = row m, lambda_new = row m + 1
    uint32_t eo_flags;                  // online transcript row | (na - 1) << 26 | (nb - 1) << 28 | ca << 30 | cb << 31
};
constexpr uint32_t MULREC_EO_BITS = 26;
constexpr uint32_t MULREC_EO_MASK = (1u << MULREC_EO_BITS) - 1;

// The cleartext pass's view of a gate (k_clear): 16 bytes when each operand is one base row (or none), 32 otherwise.
//   meta: bits 0-2 GateOp, 3 ca, 4 cb, 8-9 bases of operand a, 10-11 of operand b.  G_INPUT: a[0] = witness index.
struct ClearRec {
    uint32_t dst, a0, b0, meta;
};
struct ClearRecK {
    uint32_t dst, meta, a[RV_LIN_K], b[RV_LIN_K];
};
// per dependency level: simple records [s0, s1) of clear_s, general ones [g0, g1) of clear_k
struct ClearLevel {
    uint32_t s0, s1, g0, g1;
};

struct FlatPlan {
    bool ok = false;
    // A band = a range of the program's Mul gates plus the XOR rows no earlier band needed.  Bands run in order: the band's
    // XOR rows x-level by x-level (XOR -> XOR depth INSIDE the band: rows of earlier bands are complete), then its Mul range.
    // One band = the fewest launches; several = preprocessing rows that complete in step with the schedule (early corrections)
    // and XOR chains that can run ahead of the previous band's Mul gates on a second stream.
    struct Band {
        uint32_t x0, x1;      // x-levels [x0, x1) of `xlevels`
        uint32_t mul0, mul1;  // Mul records [mul0, mul1); mul0 a multiple of 1024
        uint32_t level_end;   // split schedule: the band's Mul gates may run once the level chain has passed levels [0, level_end)
        uint32_t on_end;      // leading online-transcript rows that are final once the Input rows and this band's Mul gates have run
    };
    std::vector<Band> bands;
    std::vector<Gate, BigAlloc<Gate>> xgates;   // G_XORK gates sorted by (band, x-level), inside one by class (two bases, others)
    std::vector<LevelRange> xlevels;            // per x-level: lo = mul11 = mul, [mul, xor2) two bases, [xor2, xork) others, hi = xork
    std::vector<MulRec, BigAlloc<MulRec>> muls; // program order: muls[ep]
    std::vector<Gate> others;                   // G_INPUT / G_ASSERT (their transcript rows): the n_other_inputs Input gates first
    uint32_t n_other_inputs = 0;
    // the cleartext pass: every gate of the circuit, level by level
    std::vector<ClearRec, BigAlloc<ClearRec>> clear_s;
    std::vector<ClearRecK, BigAlloc<ClearRecK>> clear_k;
    std::vector<ClearLevel> clear_levels;
    // the split schedule's level chain: per dependency level the value records of everything that is NOT an XOR gate (Mul, Input,
    // AssertZero: a lane each, bytes only) -- the XOR gates of the level run from the level-sorted gate stream itself
    std::vector<ClearRec, BigAlloc<ClearRec>> lite_s;
    std::vector<ClearRecK, BigAlloc<ClearRecK>> lite_k;
    std::vector<ClearLevel> lite_levels;
    uint64_t n_clear_levels = 0;                // dependency levels the cleartext pass walks (= the circuit's)
};

// false (plan.ok = false) when the circuit is not eligible: Z64 / B2A / Random gates, a streaming chunk, transcript rows
// beyond the record's 26 bits.  want_bands: equal ranges of the Mul gates
bool build_flat_plan(const Compiled& cc, FlatPlan& plan, uint32_t want_bands = 1);

// ---- device side (flatk.hip) ----
// the cleartext pass: d_v [n_rows] (its zero-row byte zeroed by the caller; bit 0 = the row's wire value, a Mul's output row
// also carries its operands' values in bits 1 and 2), d_sync two zeroed words; n_wgs workgroups of 1024 threads that must all
// be resident (the caller leaves them compute units)
void launch_clear(hipStream_t st, uint32_t n_wgs, const ClearRec* d_recs, const ClearRecK* d_recs_k, const ClearLevel* d_levels, uint32_t n_levels,
                  const uint8_t* d_wit, uint8_t* d_v, int* d_err, uint32_t* d_sync);
constexpr int RV_DEV_CLEAR_ABORT = 0x40000000;  // device error word: k_clear gave up waiting for its other workgroups
// *d_dst |= *d_src (the cleartext pass's error word joins the proof's behind the event that ends the pass)
void launch_or_word(hipStream_t st, int* d_dst, const int* d_src);
bool mul_flat_supports(uint32_t NQ);
// the chain on one XCD (kernels.hip: k_chain)
struct PLevel;
void build_chain_levels(const LevelRange* lr, size_t n_levels, uint32_t NQ, bool general, PLevel* out);
bool chain_general(const LevelRange* lr, size_t n_levels);
bool chain_supports(uint32_t NQ);
void launch_chain(hipStream_t st, uint32_t n_wgs, bool general, const Gate* d_gates, const PLevel* d_xlevels, const ClearLevel* d_lite, const ClearRec* d_lite_s,
                  const ClearRecK* d_lite_k, uint32_t l0, uint32_t l1, uint32_t* d_chosen, uint32_t* d_ctr, uint32_t* d_abort, const InterpParams& p);
// split schedule, one dependency level of the chain: the level's XOR gates (rows and value bytes, MODE_PROVE_V's arithmetic) and
// the value bytes of its other gates (kernels.hip)
void launch_level_split(hipStream_t st, const Gate* d_gates, const LevelRange& r, const ClearLevel& lite, const ClearRec* d_lite_s, const ClearRecK* d_lite_k,
                        const InterpParams& p);
// Mul records [i0, i1) (i0 a multiple of 8)
void launch_mul_flat(hipStream_t st, uint32_t NQ, const MulRec* d_recs, uint32_t i0, uint32_t i1, const uint32_t* d_rows, uint32_t* d_on, uint8_t* d_pre,
                     const uint8_t* d_v);

}  // namespace rv
