// Gate-stream compiler (host side): turns the reference's sequential instruction list
// (`for op in circuit { ins.step(op) }`, /root/reference/src/proof/mod.rs:150-152) into a
// dependency-levelled, SSA-renamed gate array the GPU can execute level by level.
//
// Everything that is a pure function of the op list is resolved here, once per circuit:
//   * wire reuse        -> every write gets a fresh SSA id (id 0 = the all-zero default
//                          wire a never-written index reads as, interpreter/single.rs:16)
//   * linear gates      -> Add/Sub/AddConst/SubConst/MulConst/Const (single.rs:71-104) are folded
//                          away: a wire is tracked as XOR of <= RV_LIN_K base rows + a constant;
//                          only Mul / Input / Random / AssertZero and the occasional materialising
//                          G_XORK reach the device, and dependency depth counts those only
//   * ShareGen::next()  -> call number `m` per gate (generator/share.rs:54-65 is a pure
//                          counter over the op list: Input 1, Random 1, Mul 2, in order)
//   * transcript rows   -> position of each hashed event in the online / preprocessing
//                          stream (transcript/prover.rs:194,210,216)
//   * reconstruction / input ordinals for the opening vectors (prover.rs:29-31)
//   * bounds errors the reference raises while stepping (Vec index panics)
#pragma once
#include <stdlib.h>
#include <stdint.h>

#include <new>
#include <utility>
#include <vector>

#include "internal.h"

namespace rv {

// Large host arrays of the compilers (hundreds of MB for a 10^7-gate circuit): anonymous mappings with transparent huge
// pages requested (a 4 KiB first-touch fault per page was a third of the parallel compiler's time), zero-filled by the
// kernel, never value-initialised a second time by the container.
void* big_alloc(size_t bytes);  // nullptr when out of memory; contents are zero
void big_free(void* p, size_t bytes);
void big_free_later(void* p, size_t bytes);
// +1 / -1 around a library call that drives the GPU (LibBusy below): the background thread of big_free_later unmaps nothing meanwhile --
// its munmap calls stall other threads' HIP calls for 10 - 100 ms (round 6: rv_prove_batch's workers right after a compile)
void lib_busy(int d);
struct LibBusy {
    LibBusy() { lib_busy(+1); }
    ~LibBusy() { lib_busy(-1); }
    LibBusy(const LibBusy&) = delete;
    LibBusy& operator=(const LibBusy&) = delete;
};  // the same from a background thread (nobody waits for an unmap)
template <class T>
struct BigAlloc {
    using value_type = T;
    BigAlloc() = default;
    template <class U>
    BigAlloc(const BigAlloc<U>&) {}
    T* allocate(size_t n) {
        void* p = big_alloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return (T*)p;
    }
    void deallocate(T* p, size_t n) { big_free_later(p, n * sizeof(T)); }
    // default-initialisation instead of value-initialisation: resize() does not write the (already zero) pages
    template <class U>
    void construct(U* p) {
        ::new ((void*)p) U;
    }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        ::new ((void*)p) U(std::forward<A>(a)...);
    }
    template <class U>
    bool operator==(const BigAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const BigAlloc<U>&) const { return false; }
};

struct Compiled {
    std::vector<Gate, BigAlloc<Gate>> gates;  // sorted by level, program order inside a level
    std::vector<uint32_t> level_start;  // gates of level l = [level_start[l], level_start[l+1])
    // inside a level gates are grouped by class (see LevelRange)
    std::vector<LevelRange> level_range;
    // pipelining aids (both monotone in l):
    std::vector<uint32_t> level_need_blocks;  // AES blocks (128 masks) that levels 0..l read
    std::vector<uint32_t> level_done_on;      // leading online-transcript rows complete once level l has run
    std::vector<uint32_t> rec_rows;     // reconstruction ordinal -> online transcript row
    std::vector<uint32_t> in_rows;      // input ordinal -> online transcript row
    uint64_t n_ssa = 1;                 // SSA wires incl. the zero wire
    uint64_t n_masks_pad = 0;           // PRG mask rows, padded to whole AES blocks (128)
    uint64_t n_rows = 1;                // share rows: n_masks_pad + computed rows (first computed = zero row)
    uint64_t n_masks = 0, n_on = 0, n_pre = 0, n_in = 0, n_rec = 0;
    uint64_t n_random_or_recon = 0;     // G_RANDOM + G_RECON gates (wire values that differ between repetitions: no MODE_PROVE_V)
    // Z64 domain (gates64 share the level numbering: level l = [level_start64[l], level_start64[l+1]))
    std::vector<Gate64> gates64;
    std::vector<uint32_t> level_start64;
    std::vector<uint64_t> rec_offs64;   // reconstruction ordinal -> word offset in the online transcript
    std::vector<uint64_t> in_offs64;    // input ordinal -> word offset in the online transcript
    uint64_t n_ssa64 = 1;
    uint64_t n_masks64 = 0, on_words64 = 0, pre_words64 = 0, n_in64 = 0, n_rec64 = 0, n_corr64 = 0;
    // row numbering: [0, row_prg_base) carried wire rows (streaming chunks only), then the PRG mask rows, then the
    // computed rows; zero_row = the all-zero row (first computed row)
    uint64_t row_prg_base = 0, zero_row = 0;
    rv_circuit_info info{};
};

// Streaming (rv_stream_*): a chunk of a longer gate stream.  The chunk starts with every wire holding the value the
// previous chunks left in it -- GF(2) wire w in share row w ("carried rows", ahead of the PRG rows), Z64 wire w in SSA
// slot 1 + w -- and ends with one extra level that writes the final value of every wire the chunk wrote back there.
// The transcript / mask counters continue where the previous chunk stopped:
struct ChunkStart {
    uint32_t mask_phase = 0;    // ShareGen<GF2>::next() calls so far, modulo 128 (the chunk's first AES block is shared)
    uint32_t mask64_phase = 0;  // ShareGen<Z64>::next() calls so far, modulo 2
    uint64_t on0 = 0, pre0 = 0;              // transcript rows reserved in front of the chunk's own (carried events)
    uint64_t on_words64_0 = 0, pre_words64_0 = 0;
};

// ShareGen::next() calls the ops make (generator/share.rs:54-65 as a pure count over the op list: Input / Random 1, Mul 2,
// B2A 64 + 63 x 2 GF(2) and one Z64) -- all a chunk compiled AHEAD of its predecessors needs to know about them
void count_masks(const rv_op* ops, size_t n_ops, uint64_t* gf2_masks, uint64_t* z64_masks);
// The transcript events the ops make -- what a chunk compiled AHEAD needs to know about the ops before it to predict the carried
// events in front of its own (stream.inc): GF(2) inputs / reconstructions (both online rows) and preprocessing rows, Z64 online
// and preprocessing words
struct StreamEvents {
    uint64_t in2 = 0, rec2 = 0, pre2 = 0, on64 = 0, pre64 = 0;
};
void count_events(const rv_op* ops, size_t n_ops, StreamEvents* ev);
// A chunk compiled with zero transcript offsets, moved behind `on0` / `pre0` carried transcript rows and `on_words64_0` /
// `pre_words64_0` carried Z64 words (they enter the compiled stream only as additive offsets)
void relocate_chunk(Compiled& cc, uint64_t on0, uint64_t pre0, uint64_t on_words64_0, uint64_t pre_words64_0);

// returns RV_OK or RV_E_*
// force_lazy_k: 0 = choose (RV_LAZY_K / circuit shape), 1..RV_LIN_K = that many base rows per wire at most
// Whole programs of RV_COMPILE_PAR_MIN ops and more (default 200 000) without B2A gates are compiled by several host threads
// (compile_par.cpp; RV_COMPILE_THREADS, default min(16, hardware threads); RV_COMPILE_SEQ=1 turns it off); everything else,
// and every program with an error in it, by the sequential compiler.  The result is the same bit for bit.
int compile_ops(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out, const ChunkStart* chunk = nullptr,
                int force_lazy_k = 0);
// the sequential compiler (one thread; the reference implementation of the gate stream, the error path, streaming chunks)
// Does keeping XORs of up to RV_LIN_K rows symbolic pay for this circuit?  Decided from its K = 1 compile: deep circuits whose
// levels fit the narrow-run kernels (at most 256 gates on average) are bound by the number of dependency levels and of
// 32-gate steps, not by row traffic -- SHA-256: 5 386 -> 4 291 levels; AES-128 (95 gates per level): 1 096 -> 624 LDS-run steps,
// 0.46 -> 0.36 ms per proof, verify 0.49 -> 0.38 ms.  Wide circuits run fastest with every XOR materialised.
// ... and then traffic is no argument against a symbolic wire either: with the fan-out limit of the wide circuits lifted SHA-256
// has 3 717 levels instead of 4 291 (3 944 steps instead of 4 368, 1.39 -> 1.28 ms per proof).  RV_LAZY_SLACK overrides.
inline uint32_t lazy_slack_for(int lazy_k, bool forced) {
    static const int env = getenv("RV_LAZY_SLACK") ? atoi(getenv("RV_LAZY_SLACK")) : -1;
    if (env >= 0) return (uint32_t)env;
    return (lazy_k > 1 && !forced) ? (1u << 30) : 1u;
}
inline bool lazy_forms_pay(uint64_t n_levels, uint64_t n_gates) { return n_gates && n_levels > 64 && n_gates / n_levels < 256 && n_gates < 5000000; }

int compile_ops_seq(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out, const ChunkStart* chunk = nullptr,
                    int force_lazy_k = 0);
// the parallel compiler: RV_OK, or RV_COMPILE_FALLBACK when the program is one it leaves to compile_ops_seq
constexpr int RV_COMPILE_FALLBACK = -1;
int compile_ops_par(const rv_op* ops, size_t n_ops, size_t z64_wires, size_t gf2_wires, Compiled& out, int force_lazy_k, int n_threads);
int compile_threads();
unsigned cpu_budget();  // logical CPUs capped by the cgroup's CPU quota
// 0 when the two compiled circuits are identical field by field, else a number naming the first difference (test hook)
int compiled_diff(const Compiled& a, const Compiled& b);

}  // namespace rv
