// Rep-sliced program: the compiled gate stream re-expressed for the LDS-resident interpreter (rep.hip), where a
// workgroup is ONE repetition (8 players = one byte per wire) and the live wires sit in LDS.
//
// north_star: "one wavefront lane = one (repetition, player) slot, wire values staged in LDS, gate stream read
// coalesced from HBM".  The reference walks the gate list once per packed group with every wire of the group in a Vec
// (/root/reference/src/interpreter/single.rs:14-23,106-157); here every share ROW of the compiled circuit
// (compile.h) gets an LDS byte slot for exactly the levels between its definition and its last reader, and the
// gates of a level are cut into SEGMENTS: runs of up to 256 gates of one kind whose transcript / mask / witness
// ordinals are consecutive, so that a wavefront streams a segment's masks and transcript bytes coalesced and writes
// its outputs as one contiguous run of slots.
//
// Only what the rep-sliced kernels support is eligible (build_rep_program says why not otherwise): prover side, pure
// GF(2), one base row per wire (compile with lazy_k = 1), no Random gates (the kernels reconstruct a wire's public
// correction as  value XOR parity(mask)  from the CLEARTEXT value, which a prover knows and which is the same in all
// repetitions -- a Random gate's value is not), live set within the LDS.
#pragma once
#include <stdint.h>

#include <vector>

#include "compile.h"

namespace rv {

enum RepSegKind : uint32_t { RS_MUL = 0, RS_XOR = 1, RS_INPUT = 2, RS_ASSERT = 3, RS_NONE = 4 };
constexpr uint32_t REP_SEG_RECS = 256;  // records reserved per segment (its recs start at segment index * REP_SEG_RECS): addressable without the header
// Gates per segment.  A lane handles the four gates whose ONLINE-TRANSCRIPT bytes share an aligned dword, so a segment
// that starts `off` = eo0 % 4 bytes into a dword occupies ceil((off + count) / 4) <= 64 lanes; 252 is a multiple of four
// (back-to-back segments of a level keep the same `off`) and leaves room for off <= 3.
constexpr uint32_t REP_SEG_MAX = 252;

struct RepRec {
    uint32_t a, b;  // operand LDS slot | constant << 31   (Xor: a carries the gate's constant; AssertZero: a only)
};

struct RepSeg {
    uint32_t kind, first, count, dst0;  // lane L, k = 0..3 <-> gate i = 4L + k - off (valid when 0 <= i < count):
                                        //   record first + 4L + k (first = segment index * REP_SEG_RECS; the `off` leading records are dummies),
                                        //   output slot dst0 + 4L + k (dst0: multiple of 4)
    uint32_t m0, eo0, ep0, x0;          // Mul: masks m0 + 2i (+1), online byte eo0 + i, preprocessing byte ep0 + i
                                        // Input: mask m0 + i, online byte eo0 + i, witness x0 + i;  AssertZero: online byte eo0 + i
    uint32_t vb0, off;                  // operand-value words of the segment: vb0 = segment index * 64 (one u32 per lane, see
                                        // k_rep_clear); off = eo0 % 4 for Mul / Input segments, 0 otherwise
    uint32_t pad0, pad1;                // 48 bytes: three 16-byte loads
};

struct RepLevel {
    uint32_t seg0, seg1;
};

// kernel arguments of k_rep_interp (rep.hip): the program, the per-proof value bits and witness, and the rep-major
// mask / transcript arrays with their per-repetition strides
struct RepParams {
    const RepLevel* levels;
    const RepSeg* segs;
    const RepRec* recs;
    const uint32_t* vbits;
    const uint8_t* wit;
    const uint8_t* masks;
    uint8_t* on;
    uint8_t* pre;
    uint64_t mask_stride, on_stride, pre_stride;
    uint32_t n_levels;
};

struct RepProgram {
    std::vector<RepLevel> levels;
    std::vector<RepSeg> segs;
    std::vector<RepRec> recs;
    uint32_t n_levels = 0, lds_slots = 0;
    uint32_t n_vb_words = 0;  // operand-value words of all Mul segments (per proof)
};

// false (and *why) when the circuit cannot take the rep-sliced path
bool build_rep_program(const Compiled& cc, uint32_t lds_slots, RepProgram& out, const char** why);

}  // namespace rv
