// Narrow / deep stretches of a GF(2) circuit with the live wires in LDS ("LDS runs").
//
// A stretch of consecutive narrow dependency levels (ripple-carry adders, AES / SHA rounds from Bristol files: thousands
// of levels of a few dozen gates) is latency-bound in the row interpreter: every level is one L2 round trip for the
// operand rows plus a workgroup barrier, ~1.2 us, on ONE compute unit (k_interp_narrow, kernels.hip).  Here the
// repetitions of the shard are cut into NQ / QS independent slices of QS quad words (4 QS repetitions); a slice is one
// workgroup whose CONSUMER wavefront walks the run step by step -- a step = up to 64 / QS gates of one level, one lane
// per (gate, quad word) -- with every wire that is live inside the run in an LDS slot (QS share words + QS words of corr bits).
// LDS operations of one wavefront execute in order, so consecutive steps need no barrier and no wait: a step costs its
// LDS gather latency plus its arithmetic.  Nothing on the consumer's path touches global memory except fire-and-forget
// stores (transcript rows, live-out wires): a PRODUCER wavefront of the same workgroup stages the step records and
// every global operand a step needs (fresh mask rows, witness bits / supplied openings) into an LDS ring one chunk of
// steps ahead, and the two meet at an LDS-only barrier once per chunk.
//
// The program (one record per gate slot of a step, slots allocated by liveness) is built on the host at circuit-compile
// time; it does not depend on NQ, only on QS.
//
// Reference semantics: interpreter/single.rs:25-157 (Instance::step) exactly as interp_one_impl (kernels.hip) restates
// them; this is a second schedule of the same gates, not a second implementation of the protocol.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <new>

#include <vector>

#include "compile.h"

namespace rv {

constexpr uint32_t LR_CHUNK = 8;          // steps per producer/consumer hand-over
constexpr uint32_t LR_NONE = 0xFFFFu;     // (builder) no LDS slot yet
// record kinds: GateOp values 0..5, plus
constexpr uint32_t LK_LOAD = 6;           // live-in wire: slot <- global row `m` and its corr bits
constexpr uint32_t LK_NOP = 7;
constexpr uint32_t LF_CA = 1u << 4, LF_CB = 1u << 5;  // operand constants
constexpr uint32_t LF_OUT = 1u << 6;      // the result is read after the run: also store row / corr bits to global memory
constexpr uint32_t LF_ON = 1u << 7;       // the gate puts a row on the online transcript (Input, Mul, AssertZero, Recon)
// one-hot copy of the kind (the step code selects its results with masks made from these bits, not with branches)
constexpr uint32_t LB_MUL = 8, LB_XOR = 9, LB_RECON = 10, LB_IN = 11, LB_OTHER = 12, LB_ASSERT = 13;

struct LdsRec {  // 32 bytes
    uint16_t a[RV_LIN_K], b[RV_LIN_K];  // operand slots (unused: slot 0 = the zero wire)
    uint16_t dst;                       // result slot (a result nothing inside the run reads goes to the run's scratch slot)
    uint16_t op;                        // kind | LF_*
    uint32_t eo, ep;                    // transcript rows (as in Gate)
    uint32_t m;                         // Input / Random / Mul: first PRG mask row; Xor / Recon / Load: the global row of the result
    uint32_t x;                         // as in Gate
};
static_assert(sizeof(LdsRec) == 32, "LdsRec layout");

struct LdsRun {
    uint32_t l0 = 0, l1 = 0;   // levels [l0, l1)
    uint32_t n_steps = 0;      // multiple of LR_CHUNK
    uint32_t n_slots = 0;      // LDS slots needed (slot 0 = the zero wire, the last one = scratch)
    uint32_t eo0 = 0, ep0 = 0; // lowest online / preprocessing transcript row written by the run
    uint64_t rec0 = 0;         // first record in the circuit's record array; step s, gate k: rec0 + s * (64 / QS) + k
};

// LDS bytes a run needs at slice width QS (ring of two chunks + the wire slots)
inline size_t lds_run_bytes(uint32_t QS, uint32_t n_slots) {
    const size_t ring = 2 * (size_t)LR_CHUNK * 4 * 64 * 16;  // per step and lane: four 16-byte fields (ldsrun.hip)
    return ring + (size_t)n_slots * QS * 8 + 64;
}

// uint32 array whose elements all start as 0xFFFFFFFF (= -1) without being written: the values are kept plus one in
// zero pages from the kernel (three arrays of n_rows entries are 144 MB for the 10^7-gate circuit, and filling them took
// longer than building the run they serve)
struct MinusOneArray {
    uint32_t* p = nullptr;
    size_t bytes = 0;
    MinusOneArray() = default;
    MinusOneArray(const MinusOneArray&) = delete;
    MinusOneArray& operator=(const MinusOneArray&) = delete;
    ~MinusOneArray() { big_free(p, bytes); }
    void init(size_t n) {
        big_free(p, bytes);
        bytes = std::max<size_t>(n, 1) * 4;
        p = (uint32_t*)big_alloc(bytes);
        if (!p) throw std::bad_alloc();
    }
    uint32_t get(size_t i) const { return p[i] - 1u; }
    int32_t geti(size_t i) const { return (int32_t)(p[i] - 1u); }
    void set(size_t i, uint32_t v) { p[i] = v + 1u; }
};
struct LdsRunScratch {  // per circuit, sized n_rows, shared by all runs
    MinusOneArray last_use_level;  // last level that reads the row (-1: never; only levels from the first narrow stretch on are looked at)
    MinusOneArray slot_of;         // row -> slot during a build (0xFFFFFFFF otherwise)
    MinusOneArray last_step;
    void init(const Compiled& cc, uint32_t first_level);
};

// Appends the run's records to `recs`.  false: the live wires do not fit `max_slots` (nothing appended).
bool build_lds_run(const Compiled& cc, uint32_t l0, uint32_t l1, uint32_t QS, uint32_t max_slots, LdsRunScratch& scratch,
                   std::vector<LdsRec>& recs, LdsRun& run);

}  // namespace rv
